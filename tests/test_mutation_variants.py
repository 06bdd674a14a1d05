"""Host-logic tests of the structural and point mutations on the CPU, through the TEST-ONLY
oracle-backed ops of tests/cpu_ops.py: every result must be a well-formed population, and each operator must do what
its reference counterpart documents (hoist/delete shrink, insert grows, point mutations keep the structure)."""
import numpy as np
import pytest
import torch

import cpu_ops


@pytest.fixture(scope="module", autouse=True)
def _cpu_ops():
    cpu_ops.register()
    from evogp_amd.tree import set_default_device, default_device

    old = default_device()
    set_default_device("cpu")
    yield
    set_default_device(old)


def _forest(pop=600, funcs=("+", "-", "*", "/", "sin", "neg", "if"), out_len=1, seed=3):
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(seed)
    desc = GenerateDescriptor(max_tree_len=128, input_len=4, output_len=out_len, using_funcs=list(funcs), max_layer_cnt=4,
                              const_samples=[-1.0, 0.0, 1.0, 0.5], out_prob=0.5)
    return Forest.random_generate(pop, desc, keys=torch.tensor([5, 6], dtype=torch.uint32)), desc


def _check_well_formed(f):
    """prefix encoding: size[i] = 1 + sizes of its arity children; size[0] = live length <= L"""
    t = f.batch_node_type.numpy().astype(np.int64) & 0x7F
    s = f.batch_subtree_size.numpy().astype(np.int64)
    L = t.shape[1]
    for r in range(t.shape[0]):
        n = s[r, 0]
        assert 1 <= n <= L, (r, n)
        def walk(i):
            ar = max(int(t[r, i]) - 1, 0)
            j = i + 1
            for _ in range(ar):
                j = walk(j)
            assert s[r, i] == j - i, (r, i, s[r, i], j - i)
            return j
        assert walk(0) == n


def test_hoist_delete_shrink_and_stay_well_formed():
    from evogp_amd.algorithm import DeleteMutation, HoistMutation

    f, _ = _forest()
    before = f.batch_subtree_size[:, 0].clone()
    for op in (HoistMutation(1.0), HoistMutation(1.0, inner_is_offset=True), DeleteMutation(1.0), DeleteMutation(1.0, max_mutatable_size=5)):
        g = op(f)
        _check_well_formed(g)
        after = g.batch_subtree_size[:, 0]
        if isinstance(op, HoistMutation) and not op.inner_is_offset:
            continue  # the reference's ABSOLUTE inner index can name the root: its hoist may also grow a tree
        assert bool((after <= before).all())
        if isinstance(op, DeleteMutation):
            assert bool((after[before > 1] < before[before > 1]).all())   # a function node always loses at least itself
            assert bool((after[before == 1] == 1).all())
    same = HoistMutation(0.0)(f)
    assert torch.equal(same.batch_node_value, f.batch_node_value) and torch.equal(same.batch_subtree_size, f.batch_subtree_size)


def test_insert_grows_and_stays_well_formed():
    from evogp_amd.algorithm import InsertMutation

    f, desc = _forest(funcs=("+", "*", "neg"))
    small = desc.update(max_layer_cnt=2)
    g = InsertMutation(1.0, small)(f)
    _check_well_formed(g)
    before, after = f.batch_subtree_size[:, 0], g.batch_subtree_size[:, 0]
    grew = after > before
    # a generated tree that is a single leaf has no position below its root: as in the reference it then simply replaces the
    # chosen subtree (insert.py:74-85) -- those trees shrink
    assert float(grew.float().mean()) > 0.5
    same = InsertMutation(0.0, small)(f)
    assert torch.equal(same.batch_node_type, f.batch_node_type)


@pytest.mark.parametrize("out_len", [1, 3])
def test_point_mutations_keep_structure_and_kind(out_len):
    from evogp_amd.algorithm import (CombinedMutation, MultiConstMutation, MultiPointMutation, SingleConstMutation,
                                     SinglePointMutation)

    f, desc = _forest(out_len=out_len)
    L = f.max_tree_len
    live = torch.arange(L)[None, :] < f.batch_subtree_size[:, :1]
    kind = f.batch_node_type.to(torch.int64) & 0x7F
    # fix_roulette: the draw is scaled by the arity class's total probability.  (The reference -- and the default here -- draw
    # in [0, 1) on a roulette whose class total is below 1 and so land on the invalid id 29 now and then: parity with that is
    # tests/test_gpu_mutation_parity.py's business, here the SENSIBLE behaviour is checked.)
    for op, max_changes in ((SinglePointMutation(1.0, desc, fix_roulette=True), 1), (MultiPointMutation(1.0, desc, 0.5, fix_roulette=True, per_node=True), L),
                            (SingleConstMutation(1.0, desc), 1), (MultiConstMutation(1.0, desc, 0.5), L),
                            (SinglePointMutation(1.0, desc, modify_output=True, fix_roulette=True), 1),
                            (CombinedMutation([SinglePointMutation(0.5, desc, fix_roulette=True), MultiConstMutation(0.5, desc)]), L)):
        g = op(f)
        assert torch.equal(g.batch_node_type, f.batch_node_type) and torch.equal(g.batch_subtree_size, f.batch_subtree_size)
        changed = (g.batch_node_value.view(torch.int32) != f.batch_node_value.view(torch.int32))
        assert not bool((changed & ~live).any())
        assert int(changed.sum(1).max()) <= max_changes
        if isinstance(op, (SingleConstMutation, MultiConstMutation)):
            assert bool((kind[changed] == 1).all())
        # functions keep their arity; variables stay in range; output indices stay in range
        v = g.batch_node_value
        is_out = (f.batch_node_type.to(torch.int64) & 0x80) != 0
        func = torch.where(is_out, (v.contiguous().view(torch.int32) & 0xFFFF).to(torch.int64), v.to(torch.int64))
        for k, lo, hi in ((2, 14, 29), (3, 1, 14), (4, 0, 1)):
            sel = live & (kind == k)
            assert bool(((func[sel] >= lo) & (func[sel] < hi)).all()), (type(op).__name__, k)
        var = v[live & (kind == 0)]
        assert bool(((var >= 0) & (var < f.input_len)).all())
        if out_len > 1:
            oi = (v.contiguous().view(torch.int32) >> 16)[live & is_out]
            assert bool(((oi >= 0) & (oi < out_len)).all())
    none = MultiPointMutation(0.0, desc)(f)
    assert torch.equal(none.batch_node_value.view(torch.int32), f.batch_node_value.view(torch.int32))
