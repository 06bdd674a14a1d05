"""CPU, world_size = 2 and 4 on the gloo backend: the sharded generation step (evogp_amd/parallel.py)
produces the same population as the single-process run — the N > 1 path of bench.py / multi-GPU
pipelines, exercised without a GPU (tree ops are served by the test-only oracle-backed CPU ops)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POP, L, GENS = 600, 32, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _selection(name):
    from evogp_amd.algorithm.selection import DefaultSelection, TournamentSelection

    return {
        "default": lambda: DefaultSelection(survival_rate=0.3, elite_rate=0.01),
        "default_more_elites_than_parents": lambda: DefaultSelection(survival_rate=0.05, elite_cnt=70),   # legal, order[:n_surv] was wrong
        "tournament_replace": lambda: TournamentSelection(3, best_probability=0.9, replace=True, survivor_rate=0.5, elite_rate=0.01),
        "tournament_noreplace": lambda: TournamentSelection(4, best_probability=1, replace=False, survivor_rate=0.7, elite_cnt=3),
        # the reference's default arguments (example/uci_sr.py:73-75): contenders from the counter-based words, no generator state
        "tournament_default_args": lambda: TournamentSelection(tournament_size=20, survivor_rate=0.5, elite_rate=0.1),
    }[name]()


SELECTIONS = ["default", "default_more_elites_than_parents", "tournament_replace", "tournament_noreplace", "tournament_default_args"]
MODES = [("rows", "exact"), ("rows", "bound"), ("packed", "exact")]


def _run(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    cpu_ops.register()
    from evogp_amd.parallel import ShardedGeneticProgramming
    from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

    set_default_device("cpu")
    desc = GenerateDescriptor(max_tree_len=L, input_len=3, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=4,
                              const_samples=[-1, 0, 1])
    n_local = POP // world
    keys = torch.tensor([42, 0], dtype=torch.int64)
    X = torch.tensor([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=torch.float32)
    y = (X.sum(1) % 2)[:, None]
    for sel in SELECTIONS:
        for exchange, cap in (MODES if world > 1 else MODES[:1]):
            local = Forest.random_generate(n_local, desc, keys=keys, tree_index_offset=rank * n_local)
            gp = ShardedGeneticProgramming(local, 0.2, desc.update(max_layer_cnt=3), selection=_selection(sel), seed=123,
                                           exchange=exchange, cap=cap)
            sent = []
            for _ in range(GENS):
                fit = -gp.forest.SR_fitness(X, y)
                fit[torch.isnan(fit)] = -torch.inf
                gp.step(fit)
                sent.append(gp.last_exchange.get("bytes_sent", 0))
            f = gp.forest
            np.savez(os.path.join(outdir, f"w{world}_r{rank}_{sel}_{exchange}_{cap}.npz"), v=f.batch_node_value.numpy(),
                     t=f.batch_node_type.numpy(), s=f.batch_subtree_size.numpy(), sent=np.array(sent),
                     collectives=np.array(gp.last_exchange.get("collectives", 0)))
    # BASELINE configs[4] shape in small: policy trees with several outputs, fitness = a rollout of every rank's
    # own trees (no collective inside the episode), per-tree episodes keyed by the GLOBAL tree index
    from evogp_amd.problem import RolloutProblem
    from evogp_amd.problem.rollout import LinearTrackingEnv

    pdesc = GenerateDescriptor(max_tree_len=L, input_len=4, output_len=2, using_funcs=["+", "-", "*", "/"], max_layer_cnt=4,
                               const_samples=[-1, 0, 1, 0.5])
    local = Forest.random_generate(n_local, pdesc, keys=keys, tree_index_offset=rank * n_local)
    prob = RolloutProblem(LinearTrackingEnv(obs_dim=4, act_dim=2, seed=3, randomize=0.5), 12, use_graph=False)
    gp = ShardedGeneticProgramming(local, 0.2, pdesc.update(max_layer_cnt=3), selection=_selection("default"), seed=77, exchange="packed")
    fits = []
    for _ in range(2):
        fit = prob.evaluate(gp.forest, tree_index_offset=rank * n_local)
        fits.append(fit.numpy().copy())
        gp.step(fit)
    f = gp.forest
    np.savez(os.path.join(outdir, f"w{world}_r{rank}_rollout.npz"), v=f.batch_node_value.numpy(), t=f.batch_node_type.numpy(),
             s=f.batch_subtree_size.numpy(), fit=np.stack(fits))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sharded"))
    mp.spawn(_run, args=(1, _free_port(), out), nprocs=1, join=True)  # separate process: leaves this one's state alone
    mp.spawn(_run, args=(2, _free_port(), out), nprocs=2, join=True)
    mp.spawn(_run, args=(4, _free_port(), out), nprocs=4, join=True)   # more than two blocks per collective
    return out


@pytest.mark.parametrize("sel", SELECTIONS)
@pytest.mark.parametrize("exchange,cap", MODES)
def test_four_ranks_equal_one_rank(runs, sel, exchange, cap):
    """world 4 == world 1 (the shape bench.py --gpus 4 / 8 has: several blocks per all-gather, shards of a quarter of the population)"""
    one = np.load(os.path.join(runs, f"w1_r0_{sel}_rows_exact.npz"))
    parts = [np.load(os.path.join(runs, f"w4_r{r}_{sel}_{exchange}_{cap}.npz")) for r in range(4)]
    for k in ("v", "t", "s"):
        both = np.concatenate([p[k] for p in parts])
        a = one[k].view(np.uint32) if k == "v" else one[k]
        b = both.view(np.uint32) if k == "v" else both
        assert np.array_equal(a, b), f"sharded population differs in {k}"


@pytest.mark.parametrize("sel", SELECTIONS)
@pytest.mark.parametrize("exchange,cap", MODES)
def test_two_ranks_equal_one_rank(runs, sel, exchange, cap):
    """world 2 == world 1 for every selection operator (BASELINE configs[2]: tournament selection over the gathered fitness)
    and every form of the exchange (two collectives with an exact or a sync-free row count; one packed collective)"""
    out = runs
    one = np.load(os.path.join(out, f"w1_r0_{sel}_rows_exact.npz"))
    parts = [np.load(os.path.join(out, f"w2_r{r}_{sel}_{exchange}_{cap}.npz")) for r in range(2)]
    for k in ("v", "t", "s"):
        both = np.concatenate([p[k] for p in parts])
        a = one[k].view(np.uint32) if k == "v" else one[k]
        b = both.view(np.uint32) if k == "v" else both
        assert np.array_equal(a, b), f"sharded population differs in {k}"
    # the population actually evolved and stayed valid
    from oracle.pyoracle import Oracle
    o = Oracle("port")
    assert all(o.validate_tree(one["t"][i], one["s"][i]) == 0 for i in range(POP))
    # what the step sent: one collective of the whole shard + its fitness, or two with at most the bound's rows
    n_local = POP // 2
    sent = parts[0]["sent"]
    if exchange == "packed":
        assert int(parts[0]["collectives"]) == 1 and (sent == n_local * (4 + 8 * L)).all()
    else:
        assert int(parts[0]["collectives"]) == 2 and (sent <= n_local * 4 + n_local * 8 * L).all()
        if cap == "exact":
            bound = np.load(os.path.join(out, f"w2_r0_{sel}_rows_bound.npz"))["sent"]
            assert (sent <= bound).all()


def test_sharded_rollout_two_ranks_equal_one_rank(runs):
    """C5 across ranks: every rank rolls out its own policy trees (per-tree episodes keyed by the global tree index), the fitness
    values meet in the sharded step; world 2 == world 1 in the fitness of every generation and in the final population"""
    one = np.load(os.path.join(runs, "w1_r0_rollout.npz"))
    parts = [np.load(os.path.join(runs, f"w2_r{r}_rollout.npz")) for r in range(2)]
    assert np.array_equal(one["fit"].view(np.uint32), np.concatenate([p["fit"] for p in parts], axis=1).view(np.uint32))
    assert len(np.unique(one["fit"][0])) > POP // 4, "the episodes do not depend on the tree"
    for k in ("v", "t", "s"):
        both = np.concatenate([p[k] for p in parts])
        assert np.array_equal(one[k].view(np.uint32) if k == "v" else one[k], both.view(np.uint32) if k == "v" else both), k


def test_selection_that_reads_the_trees_is_refused():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops
    cpu_ops.register()
    from evogp_amd.algorithm.selection import BaseSelection
    from evogp_amd.parallel import ShardedGeneticProgramming
    from evogp_amd.tree import Forest

    class BySize(BaseSelection):
        def __call__(self, forest, fitness):
            return torch.empty(0, dtype=torch.int32), torch.argsort(forest.batch_subtree_size[:, 0])[:10]

    f = Forest.zero_generate(20, 8, 2, 1)
    if f.batch_node_value.is_cuda:
        pytest.skip("host-logic test")
    with pytest.raises(TypeError, match="BaseSelection"):
        ShardedGeneticProgramming(f, 0.1, None, selection=lambda forest, fit: None)
    gp = ShardedGeneticProgramming(f, 0.1, None, selection=BySize())
    with pytest.raises(TypeError, match="cannot run in a sharded step"):
        gp.select(torch.zeros(20))


def test_pack_unpack_roundtrip():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops
    cpu_ops.register()
    from evogp_amd.parallel import _pack, _unpack
    from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device
    from evogp_amd.tree import utils as tree_utils

    saved = tree_utils._DEVICE
    try:
        _roundtrip(Forest, GenerateDescriptor, set_default_device, _pack, _unpack)
    finally:
        tree_utils._DEVICE = saved


def _roundtrip(Forest, GenerateDescriptor, set_default_device, _pack, _unpack):
    from evogp_amd.parallel import default_lists, kept_rows, plan_exchange, select_order

    set_default_device("cpu")
    desc = GenerateDescriptor(max_tree_len=L, input_len=3, output_len=2, using_funcs=["+", "*", "sin"], max_layer_cnt=4,
                              const_samples=[-1, 0, 1])
    f = Forest.random_generate(60, desc, keys=torch.tensor([1, 2]))
    g = _unpack(_pack(f), L, 3, 2)
    assert torch.equal(f.batch_node_type, g.batch_node_type)
    assert torch.equal(f.batch_node_value.view(torch.int32), g.batch_node_value.view(torch.int32))
    assert torch.equal(f.batch_subtree_size, g.batch_subtree_size)
    # the exchange plan: the best n_keep trees, wherever they live, land in the gathered table where `order` says
    torch.manual_seed(4)
    fit = torch.randn(60)
    fit[7] = fit[31]  # a tie: the stable sort prefers the lower index on every rank
    world, n_elite, n_keep = 3, 4, 17
    elites, parents = default_lists(fit, n_elite, n_keep)
    per_rank, cap, elite_rows, order = plan_exchange(elites, parents, 60, world)   # the sets by selection, table rows by index arithmetic
    assert per_rank.shape == (3, 20) and int(per_rank.sum()) == n_keep and cap == int(per_rank.sum(1).max())
    assert torch.equal(elite_rows, order[:n_elite])
    rows = [kept_rows(per_rank[r], cap) for r in range(world)]
    sends = [_pack(f[r * 20:(r + 1) * 20], rows[r]) for r in range(world)]
    table = _unpack(torch.cat(sends), L, 3, 2)
    assert table.pop_size == world * cap
    # order names the elites first, then the other survivors, each group by tree index: the sets of a stable descending sort
    ranked = torch.sort(fit, descending=True, stable=True).indices
    best = torch.cat([torch.sort(ranked[:n_elite]).values, torch.sort(ranked[n_elite:n_keep]).values])
    assert torch.equal(select_order(fit, n_elite, n_keep).long(), best)
    assert 7 in best.tolist() or 31 not in best.tolist()       # of two equal values the lower index is taken first
    assert torch.equal(table.batch_node_value[order.long()].view(torch.int32), f.batch_node_value[best].view(torch.int32))
    assert torch.equal(table.batch_subtree_size[order.long()], f.batch_subtree_size[best])
