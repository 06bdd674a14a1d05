"""GPU: the reference's structural and point mutations as one native launch each (csrc/mutate_ops.hip, round 5).

The kernels draw with the counter-based words of evogp_amd/parallel.py random_words, so every decision can be recomputed here: the
tests rebuild the draws in numpy (the same float32 arithmetic), derive what the operator must do with them from the REFERENCE's rules
(delete.py:44-105, hoist.py:43-75, single_point.py / multi_point.py / single_const.py / multi_const.py) and demand equality -- of the
decisions, and of the resulting forests with tree_crossover (already bit-exact against the reference) fed those decisions."""
import numpy as np
import pytest

from helpers import ARITH, depth2leaf, roulette_uniform

pytestmark = pytest.mark.gpu
IF, ADD, SUB, MUL, DIV, LDIV, POW, LPOW, MAX, MIN, LT, GT, LE, GE = range(14)
SIN, COS, TAN, SINH, COSH, TANH, LOG, LLOG, EXP, INV, LINV, NEG, ABS, SQRT, LSQRT = range(14, 29)
CS = [-1.0, 0.0, 1.0, 0.5, 2.0]


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


def words(seed, call, rows, lo, hi):
    import torch

    from evogp_amd.parallel import random_words

    return random_words(seed, call, rows, lo, hi, torch.device("cpu")).numpy().astype(np.int64)


def uniform(w):
    return ((w >> 7).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


@pytest.mark.parametrize("mode,L,funcs,max_size,offset,skip", [(0, 64, ARITH + [NEG, IF], 0, False, 0), (0, 256, ARITH + [SIN], 5, False, 100), (1, 64, ARITH + [NEG, IF], 0, False, 37),
                                                              (1, 256, ARITH, 0, True, 0)], ids=["delete", "delete-L256-max5", "hoist", "hoist-offset-L256"])
def test_structural_mutation_decisions_and_rows(g, oracle, mode, L, funcs, max_size, offset, skip):
    import torch

    import evogp_amd  # noqa: F401

    pop, rate, seed, call = 5003, 0.7, 123456789, 17
    v, t, s = oracle.generate(pop, L, 5, 1, 0.0, 0.5, [L, mode], depth2leaf(7 if L > 64 else 5, 0.15), roulette_uniform(funcs), CS)
    v, t, s = v.copy(), t.copy(), s.copy()
    s[11, 0] = 0                                                  # a row without a tree: copied
    keep = s[:, 0] <= L                                           # (IF can outgrow the row: such rows are whatever they are -- skipped below)
    dv, dt, ds = (torch.from_numpy(a).to(g.DEV) for a in (v, t, s))
    rv, rt, rs, dec = torch.ops.evogp_hip.structural_mutate(mode, rate, max_size, offset, skip, seed, call, dv, dt, ds, True)
    dec = dec.cpu().numpy()
    w = words(seed, call, 3, 0, pop)
    u0, u2 = uniform(w[0]), uniform(w[2])
    S = np.clip(s[:, 0].astype(np.int64), 0, L)
    want_p, want_q = np.full(pop, -1, np.int64), np.zeros(pop, np.int64)
    for n in range(pop):
        if n < skip or not (u0[n] < np.float32(rate)) or S[n] < 1:
            continue
        if mode == 0:
            if S[n] <= 1:
                continue
            sz = s[n, :S[n]].astype(np.int64)
            elig = np.nonzero((sz > 1) & ((max_size <= 0) | (sz <= max_size)))[0]
            p = int(elig[int(w[1][n]) % len(elig)]) if len(elig) else 0
            arity = (int(t[n, p]) & 0x7F) - 2 + 1
            nth = int(np.float32(1.0) + u2[n] * np.float32(arity - 1))
            c1 = min(p + 1, L - 1); c2 = min(c1 + int(s[n, c1]), L - 1); c3 = min(c2 + int(s[n, c2]), L - 1)
            want_p[n], want_q[n] = p, (c3 if nth == 3 else c2 if nth == 2 else c1)
        else:
            p = min(int(uniform(w[1][n:n + 1])[0] * np.float32(S[n])), int(S[n]) - 1)
            want_p[n], want_q[n] = p, int(u2[n] * np.float32(int(s[n, p]))) + (p if offset else 0)
    assert np.array_equal(dec[keep, 0], want_p[keep]), np.nonzero(dec[:, 0] != want_p)[0][:5]
    mut = (want_p >= 0) & keep
    assert np.array_equal(dec[mut, 1], want_q[mut])
    share = mut[skip:].mean()
    assert abs(share - rate * (np.mean(S[skip:] > 1) if mode == 0 else 1.0)) < 0.03, share
    if mode == 0:   # the replaced node is a function, the donor one of its children
        assert ((t[np.nonzero(mut)[0], want_p[mut]] & 0x7F) >= 2).all()
    # the rows: tree_crossover of every tree with itself at those nodes (replace.hip, bit-exact against the reference)
    ar = np.arange(pop, dtype=np.int32)
    cv, ct, cs = g.crossover(v, t, s, ar, ar, want_p.astype(np.int32), want_q.astype(np.int32))
    got = (rv.cpu().numpy(), rt.cpu().numpy(), rs.cpu().numpy())
    for a, b in zip(got, (cv, ct, cs)):
        a, b = a[keep], b[keep]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    livek = np.arange(L)[None, :] < S[:skip, None]                # (the rows of the elites: copied -- the live prefix; the tail is zeroed)
    assert np.array_equal(got[2][:skip][livek], s[:skip][livek]) and np.array_equal(got[0][:skip][livek].view(np.uint32), v[:skip][livek].view(np.uint32))


@pytest.mark.parametrize("L,funcs,skip", [(64, ARITH + [NEG], 0), (128, ARITH + [SIN, IF], 41)], ids=["L64", "L128-if"])
def test_insert_mutation_decisions_and_rows(g, oracle, L, funcs, skip):
    """InsertMutation in two launches (insert.py:45-85): the fresh trees are those of the donor kernel under the operator's words -- checked
    against the oracle's generate with the keys of words (7, 0 / 1) --, node p and position r are recomputed from words 1 and 2, and the
    rows must be tree_mutate(tree, p, tree_crossover(fresh, tree, r, p)) with the bit-exact kernels of replace.hip."""
    import torch

    import evogp_amd  # noqa: F401

    pop, rate, seed, call, V = 4001, 0.6, 987654321, 5, 4
    d2l, rou = depth2leaf(6 if L > 64 else 5, 0.15), roulette_uniform(funcs)
    v, t, s = (a.copy() for a in oracle.generate(pop, L, V, 1, 0.0, 0.5, [L, 9], d2l, rou, CS))
    s[7, 0] = 0                                                   # a row without a tree: copied
    keep = s[:, 0] <= L
    dv, dt, ds = (torch.from_numpy(a).to(g.DEV) for a in (v, t, s))
    d2l_f, rou_f = depth2leaf(3, 0.15), roulette_uniform(funcs)   # the operator's own descriptor: small fresh trees
    below = int(rate * (2**31 - 1))
    tabs = [torch.from_numpy(np.asarray(a, np.float32)).to(g.DEV) for a in (d2l_f, rou_f, CS)]
    fresh = torch.ops.evogp_hip.tree_generate_masked_hashed(pop, L, V, 1, len(CS), 0.0, 0.5, *tabs, 0, seed, call, below)
    rv, rt, rs, dec = torch.ops.evogp_hip.insert_mutate(below, skip, seed, call, dv, dt, ds, *fresh, True)
    dec = dec.cpu().numpy()
    w = words(seed, call, 8, 0, pop)
    mask = w[4] < below
    assert 0.55 < mask.mean() < 0.65
    keys = [int(w[7][0]) % 1000000, int(w[7][1]) % 1000000]
    fv, ft, fs = oracle.generate(pop, L, V, 1, 0.0, 0.5, keys, d2l_f, rou_f, CS)
    got_fresh = [a.cpu().numpy() for a in fresh]
    for a, b in zip(got_fresh, (fv, ft, fs)):
        assert np.array_equal(a[mask].view(np.uint8), b[mask].view(np.uint8)), "the fresh trees are not the oracle's"
    S = np.clip(s[:, 0].astype(np.int64), 0, L)
    SF = np.clip(fs[:, 0].astype(np.int64), 0, L)
    u1, u2 = uniform(w[1]), uniform(w[2])
    mut = mask & (np.arange(pop) >= skip) & (S >= 1)
    p = np.minimum((u1 * S.astype(np.float32)).astype(np.int64), S - 1)
    r = (np.float32(1.0) + u2 * (SF - 1).astype(np.float32)).astype(np.int64)
    # tree_crossover: recipients = the fresh trees, donors = the trees; then tree_mutate of the trees with the grafted rows
    ar = np.arange(pop, dtype=np.int32)
    both = tuple(np.concatenate([a, b]) for a, b in zip((fv, ft, fs), (v, t, s)))
    gr = g.crossover(*both, ar, ar + pop, np.where(mut, r, -1).astype(np.int32), np.where(mut, p, 0).astype(np.int32))
    want = g.mutate(v, t, s, np.where(mut, p, -1).astype(np.int32), *gr)
    got = (rv.cpu().numpy(), rt.cpu().numpy(), rs.cpu().numpy())
    sel = keep & (fs[:, 0] <= L)
    live = np.arange(L)[None, :] < np.clip(want[2][:, :1].astype(np.int64), 0, L)
    for a, b in zip(got, want):
        a, b = a[sel], b[sel]
        lv = live[sel]
        assert np.array_equal(a[lv].view(np.uint8), b[lv].view(np.uint8)), "rows differ from tree_mutate(tree, p, tree_crossover(fresh, tree, r, p))"
    changed = (got[2][:, 0] != s[:, 0]) & sel
    assert changed[skip:].mean() > 0.3 and not changed[:skip].any()
    assert np.array_equal(dec[mut & sel & changed, 0], p[mut & sel & changed]) and np.array_equal(dec[mut & sel, 1], r[mut & sel])


def _roulettes(funcs):
    w = np.zeros(29, np.float64); w[list(funcs)] = 1.0 / len(funcs)
    p = w.astype(np.float32)

    def of(lo, hi):
        only = np.zeros(29, np.float32); only[lo:hi] = p[lo:hi]
        return np.cumsum(only, dtype=np.float32)
    return of(14, 29), of(1, 14), of(0, 1)


@pytest.mark.parametrize("mode,per_node,modify_output,fix,out_len", [(0, False, False, False, 1), (0, True, True, True, 3), (1, False, False, False, 1), (1, False, True, False, 3),
                                                                     (2, False, False, False, 1), (2, True, False, False, 1), (3, False, False, False, 1)],
                         ids=["multi-point", "multi-point-per-node-out-fix", "single-point", "single-point-out", "multi-const", "multi-const-per-node", "single-const"])
def test_point_mutations_node_by_node(g, oracle, mode, per_node, modify_output, fix, out_len):
    import torch

    import evogp_amd  # noqa: F401

    pop, L, V, rate, intensity, seed, call, skip = 3001, 64, 6, 0.6, 0.4, 987654321, 5, 29
    funcs = ARITH + [NEG, SIN, IF, MAX]
    consts = np.linspace(-3, 3, 101).astype(np.float32)
    v, t, s = oracle.generate(pop, L, V, out_len, 0.5 if out_len > 1 else 0.0, 0.5, [mode, out_len], depth2leaf(5, 0.15), roulette_uniform(funcs), CS)
    ru, rb, rt_ = _roulettes(funcs)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(g.DEV)
    got = torch.ops.evogp_hip.point_mutate(mode, rate, intensity, per_node, modify_output, fix, skip, V, out_len, seed, call, dev(v), dev(t), dev(s), dev(ru),
                                           dev(rb), dev(rt_), dev(consts)).cpu().numpy()
    wt = words(seed, call, 2, 0, pop)
    wn = words(seed, call, 13, 0, pop * L).reshape(13, pop, L)
    u0, u1 = uniform(wt[0]), uniform(wt[1])
    S = np.clip(s[:, 0].astype(np.int64), 0, L)
    mutate = (np.arange(pop) >= skip) & (u0 < np.float32(rate))
    live = np.arange(L)[None, :] < S[:, None]
    if mode == 1:
        pos = np.minimum((u1 * S.astype(np.float32)).astype(np.int64), S - 1)
        target = mutate[:, None] & (np.arange(L)[None, :] == pos[:, None]) & live
    elif mode == 3:
        target = np.zeros((pop, L), bool)
        for n in np.nonzero(mutate)[0]:
            c = np.nonzero((t[n, :S[n]] == 1))[0]
            if len(c):
                target[n, c[int(wt[1][n]) % len(c)]] = True
    else:
        on = (uniform(wn[12]) < np.float32(intensity)) if per_node else (u1 < np.float32(intensity))[:, None]
        target = mutate[:, None] & live & on
    if mode >= 2:
        target &= t == 1
    want = v.copy()
    kind = t.astype(np.int64) & 0x7F
    is_out = (t.astype(np.int64) & 0x80) != 0
    bits = v.view(np.uint32).astype(np.int64)
    const_new = consts[np.minimum((uniform(wn[10]) * np.float32(len(consts))).astype(np.int64), len(consts) - 1)]
    var_new = np.minimum((uniform(wn[9]) * np.float32(V)).astype(np.int64), V - 1).astype(np.float32)
    u8 = uniform(wn[8])
    func = np.zeros((pop, L), np.int64)
    for k, rou in ((2, ru), (3, rb), (4, rt_)):
        sel = kind == k if k < 4 else kind >= 4
        if fix:
            total = rou[-1]
            idx = np.minimum(np.searchsorted(rou, (u8 * total).astype(np.float32), side="right"), 28)
            old_func = np.where(is_out, bits & 0xFFFF, v.astype(np.int64))
            idx = idx if total > 0 else old_func
        else:
            idx = np.searchsorted(rou, u8, side="left")
        func = np.where(sel, idx, func)
    oi = bits >> 16
    if modify_output:
        oi = np.minimum((uniform(wn[11]) * np.float32(out_len)).astype(np.int64), out_len - 1)
    packed = ((func + (oi << 16)) & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    func_new = np.where(is_out, packed, func.astype(np.float32))
    fresh = np.where((mode >= 2) | (kind == 1), const_new, np.where(kind == 0, var_new, func_new))
    want = np.where(target, fresh, v).astype(np.float32)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
    assert len(bad[0]) == 0, (len(bad[0]), bad[0][:5], bad[1][:5], got[bad][:5], want[bad][:5], t[bad][:5])
    assert target.any() and abs(mutate[skip:].mean() - rate) < 0.04
    assert np.array_equal(got[:skip].view(np.uint32), v[:skip].view(np.uint32))


def test_generation_step_with_the_reference_paper_operator_set(g, oracle):
    """example/brax_task.py:38-45: DefaultCrossover + CombinedMutation[DefaultMutation(0.2), DeleteMutation(0.8)] under DefaultSelection --
    the fused breeding pass and ONE more launch; elites come first and unchanged, every row is a well-formed tree (its fitness is what the
    oracle computes for it), trees get shorter on average under Delete, and the composed torch programs (EVOGP_NATIVE_MUTATION=0) still run."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import (CombinedMutation, DefaultCrossover, DefaultMutation, DefaultSelection, DeleteMutation, GeneticProgramming, HoistMutation,
                                     InsertMutation, MultiConstMutation, SinglePointMutation)
    from evogp_amd.tree import Forest, GenerateDescriptor
    from helpers import assert_close_classes, c2_dataset

    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    desc = GenerateDescriptor(max_tree_len=128, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
    pop = 6000
    X, y = c2_dataset()
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    for ops in ([DefaultMutation(0.2, desc.update(max_layer_cnt=3)), DeleteMutation(0.8)], [HoistMutation(0.3), InsertMutation(0.3, desc.update(max_layer_cnt=3))],
                [SinglePointMutation(0.5, desc, fix_roulette=True), MultiConstMutation(0.5, desc)]):
        algo = GeneticProgramming(Forest.random_generate(pop, desc, keys=torch.tensor([5, 6], dtype=torch.uint32, device=dev)), DefaultCrossover(),
                                  CombinedMutation(ops), DefaultSelection(survival_rate=0.3, elite_rate=0.01))
        assert algo._native_plan() is not None
        for gen in range(3):
            fit = -algo.forest.SR_fitness(Xd, yd)
            fit = torch.where(torch.isnan(fit), torch.full_like(fit, float("-inf")), fit)
            best = torch.topk(fit, 60).values.sort().values       # DefaultSelection(elite_rate 0.01): the 60 best come first, untouched by any mutation
            algo.step(fit)
            f = algo.forest
            again = -f.SR_fitness(Xd, yd)[:60]
            assert torch.equal(again.sort().values, best), "the first rows of the next generation are not the elites"
        trees = (f.batch_node_value.cpu().numpy(), f.batch_node_type.cpu().numpy(), f.batch_subtree_size.cpu().numpy())
        assert_close_classes(f.SR_fitness(Xd, yd).cpu().numpy(), oracle.sr_fitness(*trees, X, y), 1e-5, what=f"generation 3 under {[type(o).__name__ for o in ops]}")


# ---- the native kernels on the REFERENCE's own draws ------------------------------------------------------------
# tests/golden/mutation_*.npz hold, for thirteen runs of the reference's Python operators (tests/golden/make_mutation_golden.py), the
# input forest, every random number the operator drew and its result.  evogp_hip_debug_*_given (include/evogp_hip_debug.h) run the
# kernels of csrc/mutate_ops.hip with those draws handed in instead of hashed: the forests must equal the reference's, bit for bit on
# the live prefix of every tree.  (The draws arrive in the reference's meaning: a node index, a child number, a uniform number for the
# roulette search, an index into the constants -- what the kernels compute from their counter words in the tests above.)
def _native_on_the_reference_draws(g, case):
    from evogp_amd.tree import utils as tree_utils

    before = tree_utils._DEVICE            # (the descriptors and forests below are built on the CPU; the suite's default goes back in place)
    try:
        return _native_on_the_reference_draws_cpu_tables(g, case)
    finally:
        tree_utils._DEVICE = before


def _native_on_the_reference_draws_cpu_tables(g, case):
    import torch

    import mutation_replay as mr
    from evogp_amd.algorithm import MultiConstMutation, MultiPointMutation, SingleConstMutation, SinglePointMutation
    from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

    set_default_device("cpu")
    z, meta, log = mr.load(case)
    v, t, s = z["in_value"], z["in_type"], z["in_size"]
    pop, L = v.shape
    dk, par, kind = meta["descriptor"], meta["params"], meta["operator"]
    rate = par["mutation_rate"]
    mask = log[0] < np.float32(rate)
    if kind == "DeleteMutation":
        mask = mask & (s[:, 0] > 1)
    idx = np.nonzero(mask)[0]

    def per_tree(a, fill=0):
        full = np.full((pop,) + a.shape[1:], fill, dtype=a.dtype)
        full[idx] = a
        return full

    if kind == "HoistMutation":
        given = np.stack([mask.astype(np.int64), per_tree(log[1]).astype(np.int64), per_tree(log[2]).astype(np.int64)], 1)
        return g.structural_mutate_given(v, t, s, 1, given)
    if kind == "DeleteMutation":
        # delete.py:66-85: the node is the arg-max of the uniform scores over the function nodes that are small enough (0: the root)
        sizes = s.astype(np.int64)
        score = per_tree(log[1]) * (np.arange(L)[None, :] < sizes[:, :1])
        score = np.where(sizes == 1, 0, score)
        if par.get("max_mutatable_size"):
            score = np.where(sizes > par["max_mutatable_size"], 0, score)
        given = np.stack([mask.astype(np.int64), np.argmax(score, 1), per_tree(log[2]).astype(np.int64)], 1)
        return g.structural_mutate_given(v, t, s, 0, given)
    if kind == "InsertMutation":
        od = meta["op_descriptor"]
        set_default_device("cuda:0")
        fresh = Forest.random_generate(len(idx), GenerateDescriptor(**od), keys=torch.from_numpy(log[2].astype(np.int64)).to(torch.uint32).to("cuda:0"))
        fr = [np.zeros((pop, L), a.dtype) for a in (v, t, s)]
        for dst, src in zip(fr, (fresh.batch_node_value, fresh.batch_node_type, fresh.batch_subtree_size)):
            dst[idx] = src.cpu().numpy()      # the reference generates the fresh trees for the mutating trees only: row = rank among them
        given = np.stack([mask.astype(np.int64), per_tree(log[1]).astype(np.int64), per_tree(log[3]).astype(np.int64)], 1)
        return g.insert_mutate_given(v, t, s, given, fr)
    desc = GenerateDescriptor(**dk)
    forest = Forest(dk["input_len"], dk["output_len"], torch.from_numpy(v), torch.from_numpy(t), torch.from_numpy(s))
    tm, consts = torch.from_numpy(mask), desc.const_samples.numpy()
    rous = [r.numpy() for r in (desc.roulette_ufuncs, desc.roulette_bfuncs, desc.roulette_tfuncs)]
    if kind in ("SinglePointMutation", "MultiPointMutation"):
        modify = par.get("modify_output", False)
        if kind == "SinglePointMutation":
            targets = SinglePointMutation(rate, desc, modify_output=modify).targets(forest, tm, torch.from_numpy(per_tree(log[1]))).numpy()
        else:
            targets = MultiPointMutation(rate, desc, par["mutation_intensity"], modify_output=modify).targets(forest, tm, torch.from_numpy(per_tree(log[1], fill=2.0))).numpy()
        names = ["u_uf", "u_bf", "u_tf"] + (["out_idx"] if modify else []) + ["var_idx", "const_idx"]
        draws = {}
        for name, a in zip(names, log[2:]):
            full = np.zeros((pop, L), a.dtype)
            full[targets] = a                  # the reference draws one number per target, in row-major order of the mutating trees
            draws[name] = full
        k = t.astype(np.int64) & 0x7F
        u = np.where(k >= 4, draws["u_tf"], np.where(k == 3, draws["u_bf"], draws["u_uf"]))   # the draw of the node's own arity class (single_point.py:86-89)
        out = g.point_mutate_given(v, t, s, 1 if kind == "SinglePointMutation" else 0, targets, u, draws["var_idx"], draws["const_idx"], draws.get("out_idx"),
                                   rous, consts, dk["input_len"], dk["output_len"], modify_output=modify)
        return out, t, s
    if kind == "SingleConstMutation":
        targets = SingleConstMutation(rate, desc).targets(forest, tm, torch.from_numpy(per_tree(log[1]))).numpy()
        ci = np.broadcast_to(per_tree(log[2])[:, None], (pop, L))
        return g.point_mutate_given(v, t, s, 3, targets, None, None, ci, None, None, consts, dk["input_len"], dk["output_len"]), t, s
    assert kind == "MultiConstMutation", kind
    targets = MultiConstMutation(rate, desc, par["mutation_intensity"]).targets(forest, tm, torch.from_numpy(per_tree(log[1], fill=2.0))).numpy()
    ci = np.zeros((pop, L), np.int64)
    ci[targets] = log[2]
    return g.point_mutate_given(v, t, s, 2, targets, None, None, ci, None, None, consts, dk["input_len"], dk["output_len"]), t, s


def _golden_cases():
    import mutation_replay as mr

    return mr.cases()


@pytest.mark.parametrize("case", _golden_cases())
def test_native_kernels_reproduce_the_reference_operators_from_its_own_draws(g, case):
    import mutation_replay as mr

    got = _native_on_the_reference_draws(g, case)
    z, _, _ = mr.load(case)
    want = (z["out_value"], z["out_type"], z["out_size"])
    assert np.array_equal(got[2][:, 0], want[2][:, 0]), f"{case}: tree lengths differ"
    live = np.arange(got[2].shape[1])[None, :] < got[2][:, :1].astype(np.int64)      # the reference leaves the tails undefined
    assert (want[0].view(np.uint32) != z["in_value"].view(np.uint32)).any(), "the recorded run changed nothing"
    for name, a, b in zip(("value", "type", "size"), got, want):
        a = a.view(np.uint32) if a.dtype == np.float32 else a
        b = b.view(np.uint32) if b.dtype == np.float32 else b
        bad = np.argwhere((a != b) & live)
        assert bad.size == 0, f"{case}: {name} differs at tree {bad[0][0]} node {bad[0][1]} ({len(bad)} entries)"
