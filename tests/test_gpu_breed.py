"""GPU parity of the fused default generation step: evogp_hip_generate_masked and
evogp_hip_breed_default through the C ABI against the oracle's generate / crossover / mutate composed the way
GeneticProgramming.step composes them (genetic_programming.py:105-124), on explicit random words."""
import numpy as np
import pytest

from helpers import ARITH, ALLF, assert_forest_equal, depth2leaf, roulette_uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpu_capi

    return gpu_capi


@pytest.fixture(scope="module")
def oracle():
    from oracle.pyoracle import Oracle

    return Oracle("port")


def _expected(oracle, forest, order, rnd, below, n_elite, n_surv, donors, parents=None, pop=None):
    """What the reference's operators produce for these draws (genetic_programming.py:110-122: elites = forest[elite indices],
    parents drawn from forest[survivor indices] — a list that may repeat trees)."""
    v, t, s = forest
    pop = v.shape[0] if pop is None else pop
    L = v.shape[1]
    n_new = pop - n_elite
    sizes = s[:, 0].astype(np.int64)
    r = rnd.astype(np.int64)
    parents = order if parents is None else parents
    li = parents[r[0] % n_surv]
    ri = parents[r[1] % n_surv]
    p = (r[2] % sizes[li]).astype(np.int32)
    q = (r[3] % sizes[ri]).astype(np.int32)
    child = [a.copy() for a in oracle.crossover(v, t, s, li.astype(np.int32), ri.astype(np.int32), p, q)]
    mut = r[4] < below
    pm = np.full(n_new, -1, np.int32)
    if mut.any():
        cs = child[2][mut, 0].astype(np.int64)
        pm_m = ((r[5][mut] % 1024) % cs).astype(np.int32)
        pm[mut] = pm_m
        res = oracle.mutate(child[0][mut], child[1][mut], child[2][mut], pm_m, donors[0][mut], donors[1][mut], donors[2][mut])
        for a, b in zip(child, res):
            a[mut] = b
    elite = order[:n_elite]
    out = tuple(np.concatenate([src[elite], ch]) for src, ch in zip((v, t, s), child))
    dec = np.stack([li, ri, p, q, mut.astype(np.int64), pm], axis=1).astype(np.int32)
    return out, dec


@pytest.mark.parametrize("pop,L,mlc,dmlc,funcs,rate,elite_rate,surv_rate", [
    (4000, 64, 6, 3, ARITH, 0.2, 0.01, 0.3),
    (3000, 32, 4, 3, ARITH, 1.0, 0.0, 0.5),
    (2500, 128, 5, 4, ALLF, 0.5, 0.1, 0.05),
    (700, 1024, 9, 5, ARITH, 0.3, 0.02, 1.0),
    (64, 16, 3, 2, ARITH, 0.0, 0.5, 0.3),
    (900, 50, 5, 3, ARITH, 0.4, 0.03, 0.3),    # row length no multiple of 4: the one-row-per-wave kernel
    (1100, 400, 7, 4, ARITH, 0.3, 0.02, 0.4),  # staging rows of 16 groups would not fit: the one-row-per-wave kernel
    (100_000, 64, 6, 3, ARITH, 0.2, 0.01, 0.3),    # BASELINE configs[1] at its full size: the generation step of bench.py's configs1
    (1_000_000, 64, 6, 3, ARITH, 0.2, 0.01, 0.3),  # ... and the headline population (four chunks per workgroup unit, gathered donor launch)
])
def test_breed_default_bit_exact(g, oracle, pop, L, mlc, dmlc, funcs, rate, elite_rate, surv_rate):
    rng = np.random.default_rng(pop + L)
    forest = oracle.generate(pop, L, 5, 1, 0.5, 0.5, [11, 22], depth2leaf(mlc), roulette_uniform(funcs), [-1, 0, 1, 0.5])
    n_elite, n_surv = int(pop * elite_rate), max(1, int(pop * surv_rate))
    n_new = pop - n_elite
    fitness = rng.normal(size=pop).astype(np.float32)
    order = np.argsort(-fitness, kind="stable").astype(np.int32)[:max(n_elite, n_surv)]
    rnd = rng.integers(0, 2**31 - 1, (6, n_new)).astype(np.int32)
    below = int(rate * (2**31 - 1))
    keys = [7, 9]
    dargs = (n_new, L, 5, 1, 0.5, 0.5, keys, depth2leaf(dmlc), roulette_uniform(funcs), [-1, 0, 1])
    donors = oracle.generate(*dargs)
    got_d = g.generate_masked(*dargs, rnd[4], below)
    act = rnd[4].astype(np.int64) < below
    for a, b in zip(got_d, donors):
        assert np.array_equal(a[act].view(np.uint8), b[act].view(np.uint8)), "masked generate: active rows differ from tree_generate"
    # inactive rows keep the poison the helper put there (value NaN, type/size -7)
    assert np.isnan(got_d[0][~act]).all() and (got_d[1][~act] == -7).all() and (got_d[2][~act] == -7).all()
    want, want_dec = _expected(oracle, forest, order, rnd, below, n_elite, n_surv, donors)
    # hand the kernel the masked donor rows only (the others are poison): it must not read them
    got, dec = g.breed_default(*forest, order, rnd, below, n_elite, n_surv, *got_d)
    assert np.array_equal(dec, want_dec), "breed: decisions differ"
    assert_forest_equal(got, want, "breed_default")


@pytest.mark.parametrize("pop,L,n_elite,n_surv,rate", [
    (3000, 64, 30, 1500, 0.2),      # a tournament's survivor list: half the population, many repeats
    (2000, 32, 0, 2600, 0.5),       # no elites, more parents than trees (drawn with replacement)
    (1500, 128, 400, 7, 1.0),       # more elites than parents
    (600, 50, 5, 300, 0.3),         # the one-row-per-wave kernel
])
def test_breed_lists_bit_exact(g, oracle, pop, L, n_elite, n_surv, rate):
    """evogp_hip_breed_lists — the breeding pass under ANY selection operator (BASELINE configs[2]: tournament selection):
    separate elite and parent lists, the parent list with repeats as selection/tournament.py:59-133 returns it; against the
    oracle's crossover / mutate composed as genetic_programming.py:110-122 composes them, whole population and in slices."""
    rng = np.random.default_rng(pop * 7 + L)
    forest = oracle.generate(pop, L, 5, 1, 0.5, 0.5, [3, 4], depth2leaf(5), roulette_uniform(ARITH), [-1, 0, 1, 0.5])
    n_new = pop - n_elite
    elites = rng.permutation(pop)[:n_elite].astype(np.int32)          # any order, no repeats
    parents = rng.integers(0, pop // 3, n_surv).astype(np.int32)      # repeats
    rnd = rng.integers(0, 2**31 - 1, (6, n_new)).astype(np.int32)
    below = int(rate * (2**31 - 1))
    dargs = (n_new, L, 5, 1, 0.5, 0.5, [7, 9], depth2leaf(3), roulette_uniform(ARITH), [-1, 0, 1])
    donors = oracle.generate(*dargs)
    want, want_dec = _expected(oracle, forest, elites, rnd, below, n_elite, n_surv, donors, parents=parents)
    pad = [np.concatenate([np.zeros((n_elite, L), a.dtype), a]) for a in donors]    # donor row k belongs to next-generation row k
    got, dec = g.breed_lists(*forest, elites, parents, rnd, below, *pad)
    assert np.array_equal(dec[n_elite:], want_dec), "breed_lists: decisions differ"
    assert_forest_equal(got, want, "breed_lists")
    # three ragged slices concatenate to the same rows
    cuts = [0, pop // 3 + 1, pop // 3 + 2, pop]
    parts = [g.breed_lists(*forest, elites, parents, rnd, below, *[a[lo:hi] for a in pad], pop=pop, row_begin=lo, row_count=hi - lo)[0]
             for lo, hi in zip(cuts[:-1], cuts[1:])]
    assert_forest_equal(tuple(np.concatenate([p[k] for p in parts]) for k in range(3)), want, "breed_lists in slices")


def test_genetic_programming_default_step_uses_the_fused_path_and_stays_valid(g):
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5,
                              const_samples=[-1, 0, 1])
    mdesc = desc.update(max_layer_cnt=3)
    pop = 5000
    forest = Forest.random_generate(pop, desc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
    algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, mdesc), DefaultSelection(0.3, elite_rate=0.01))
    assert algo._native_plan() is not None
    X = torch.rand(256, 4, device=dev) * 4 - 2
    y = (X[:, 0] * X[:, 1] - X[:, 2]).unsqueeze(1)
    best = []
    for _ in range(6):
        fit = -algo.forest.SR_fitness(X, y)
        fit[torch.isnan(fit)] = -torch.inf
        best.append(float(fit.max()))
        top = algo.forest[int(torch.argmax(fit))]
        new = algo.step(fit)
        # the elites survive verbatim in the first rows (by tree index, not by fitness: nothing uses an order among them); every
        # row is a structurally valid tree
        n_elite = DefaultSelection(0.3, elite_rate=0.01).counts(pop)[0]
        assert str(top) in {str(new[i]) for i in range(n_elite)}
        sizes = new.batch_subtree_size[:, 0].to(torch.int64)
        assert int(sizes.min()) >= 1 and int(sizes.max()) <= 64
        ntype = new.batch_node_type.to(torch.int64)
        live = torch.arange(64, device=dev)[None, :] < sizes[:, None]
        leaf = (ntype <= 1)
        delta = torch.where(live, torch.where(leaf, 1, -1), 0)   # binary functions only
        assert bool((delta.sum(1) == 1).all())
    assert best[-1] >= best[0]


def test_structural_and_point_mutations_on_the_device(g):
    """The N3 operators through the real ops (tree_crossover with left position -1 = copy, tree_mutate, masked
    generation): well-formed populations, expected size behaviour."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import (CombinedMutation, DeleteMutation, HoistMutation, InsertMutation, MultiConstMutation,
                                     MultiPointMutation, SingleConstMutation, SinglePointMutation)
    from evogp_amd.tree import Forest, GenerateDescriptor
    from test_mutation_variants import _check_well_formed

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=128, input_len=4, output_len=2, using_funcs=["+", "-", "*", "/", "sin", "neg", "if"],
                              max_layer_cnt=4, const_samples=[-1.0, 0.0, 1.0, 0.5])
    f = Forest.random_generate(3000, desc, keys=torch.tensor([5, 6], dtype=torch.uint32, device=dev))
    before = f.batch_subtree_size[:, 0].clone()

    def cpu(forest):
        return Forest(forest.input_len, forest.output_len, forest.batch_node_value.cpu(), forest.batch_node_type.cpu(),
                      forest.batch_subtree_size.cpu())

    # (inner_is_offset: the hoist that can only shrink; the reference's absolute inner index is covered by test_gpu_mutation_parity.py)
    for op in (HoistMutation(0.7, inner_is_offset=True), DeleteMutation(0.7), InsertMutation(0.7, desc.update(max_layer_cnt=2)),
               SinglePointMutation(0.7, desc), MultiPointMutation(0.7, desc), SingleConstMutation(0.7, desc),
               MultiConstMutation(0.7, desc),
               CombinedMutation([HoistMutation(0.3, inner_is_offset=True), InsertMutation(0.3, desc.update(max_layer_cnt=2)), DeleteMutation(0.3)])):
        out = op(f)
        _check_well_formed(cpu(out))
        after = out.batch_subtree_size[:, 0]
        if isinstance(op, (HoistMutation, DeleteMutation)):
            assert bool((after <= before).all()) and bool((after < before).any())
        if isinstance(op, InsertMutation):
            assert bool((after > before).any())
        if "Point" in type(op).__name__ or "Const" in type(op).__name__:
            assert torch.equal(out.batch_subtree_size, f.batch_subtree_size)
            assert not torch.equal(out.batch_node_value, f.batch_node_value)
    assert torch.equal(f.batch_subtree_size[:, 0], before)   # operators never modify their input


def test_rollout_problem_graph_replay_matches_eager(g):
    """N4: one captured step of the rollout loop replayed as a HIP graph gives the eager result."""
    import time

    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.problem import LinearTrackingEnv, RolloutProblem
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    env = LinearTrackingEnv(device=dev)
    desc = GenerateDescriptor(max_tree_len=64, input_len=17, output_len=6, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5,
                              const_samples=[-1.0, 0.0, 1.0, 0.5], out_prob=0.5)
    forest = Forest.random_generate(20000, desc, keys=torch.tensor([8, 9], dtype=torch.uint32, device=dev))
    eager = RolloutProblem(env, 60, use_graph=False)
    graph = RolloutProblem(env, 60, use_graph=True)
    a = eager.evaluate(forest)
    b = graph.evaluate(forest)
    assert torch.equal(a, b)
    for use_graph, name in ((False, "eager"), (True, "graph")):
        prob = RolloutProblem(env, 500, use_graph=use_graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prob.evaluate(forest)
        torch.cuda.synchronize()
        print(f"rollout {name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per step at pop 20000 (500 steps, capture included)")


@pytest.mark.parametrize("selection", ["default", "more_elites_than_parents", "tournament_replace", "tournament_noreplace", "tournament_default_args"])
def test_sharded_native_step_union_equals_single_device(g, selection):
    """SURVEY.md §8e on one GPU: the rows the ranks of a G-way sharded run would build (same gathered fitness, same seed)
    concatenate to the single-device next generation, bit for bit, for G = 1, 2, 3, 8 — under DefaultSelection and under the
    TournamentSelection BASELINE configs[2] names (selection/tournament.py:59-133; its survivor list repeats trees)."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import DefaultSelection
    from evogp_amd.algorithm.selection import TournamentSelection
    from evogp_amd.parallel import ShardedGeneticProgramming, _pack, _unpack, kept_rows, plan_exchange
    from evogp_amd.tree import Forest, GenerateDescriptor

    make = {"default": lambda: DefaultSelection(0.3, elite_rate=0.01),
            "tournament_default_args": lambda: TournamentSelection(20, survivor_rate=0.5, elite_rate=0.1),
            "more_elites_than_parents": lambda: DefaultSelection(0.02, elite_cnt=500),
            "tournament_replace": lambda: TournamentSelection(3, best_probability=0.9, replace=True, survivor_rate=0.5, elite_rate=0.01),
            "tournament_noreplace": lambda: TournamentSelection(5, best_probability=1, replace=False, survivor_rate=0.8, elite_cnt=7)}[selection]
    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=5, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5,
                              const_samples=[-1.0, 0.0, 1.0])
    pop = 4800
    full = Forest.random_generate(pop, desc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
    fitness = torch.randn(pop, device=dev)
    fitness[::13] = float("-inf")

    def same(got, ref, what):
        for a, b in zip(got, ref):
            assert torch.equal(a.view(torch.uint8) if a.dtype != torch.float32 else a.view(torch.int32),
                               b.view(torch.uint8) if b.dtype != torch.float32 else b.view(torch.int32)), what

    def run(G):
        parts = []
        for r in range(G):
            sg = ShardedGeneticProgramming(full[:pop // G], 0.2, desc.update(max_layer_cnt=3), make(), seed=5)
            parts.append(sg.next_slice_native(full, fitness, r * (pop // G), (r + 1) * (pop // G)))
        return [torch.cat([getattr(p, n) for p in parts]) for n in ("batch_node_value", "batch_node_type", "batch_subtree_size")]

    ref = run(1)
    assert int(ref[2][:, 0].min()) >= 1
    sg = ShardedGeneticProgramming(full, 0.2, desc.update(max_layer_cnt=3), make(), seed=5)
    elites, parents = sg.select(fitness)
    n_elite, n_surv = sg.selection.counts(pop)
    assert elites.numel() == n_elite and parents.numel() == n_surv
    # the elites are the best trees, copied verbatim to the first rows
    best = torch.topk(fitness, max(n_elite, 1)).indices[:n_elite]
    assert set(elites.tolist()) == set(best.tolist())
    assert torch.equal(ref[0][:n_elite].view(torch.int32), full.batch_node_value[elites.long()].view(torch.int32))
    if selection.startswith("tournament"):
        assert parents.unique().numel() < n_surv                               # repeats: winners of several tournaments
        pf = fitness[parents.long()]
        assert float(pf[torch.isfinite(pf)].mean()) > float(fitness[torch.isfinite(fitness)].mean()) + 0.3   # selection pressure
    for G in (2, 3, 8):
        same(run(G), ref, f"G = {G}")
    # the exchange of a sharded run gathers only the trees the two lists name: the slices built from that compact table
    # (lists expressed in table rows; exact row count or the sync-free bound) are the same rows again
    for G in (2, 8):
        n_local = pop // G
        for cap_mode in ("exact", "bound"):
            sg = ShardedGeneticProgramming(full[:n_local], 0.2, desc.update(max_layer_cnt=3), make(), seed=5, cap=cap_mode)
            elites, parents = sg.select(fitness)
            cap = None if cap_mode == "exact" else sg._cap_bound(elites.numel(), parents.numel())
            per_rank, cap, elite_rows, parent_rows = plan_exchange(elites, parents, pop, G, cap)
            assert cap <= n_local and int(per_rank.sum(1).max()) <= cap
            rows = [kept_rows(per_rank[r], cap) for r in range(G)]
            table = _unpack(torch.cat([_pack(full[r * n_local:(r + 1) * n_local], rows[r]) for r in range(G)]), 64, 5, 1)
            parts = [sg.slice_native(table, elite_rows, parent_rows, pop, r * n_local, (r + 1) * n_local) for r in range(G)]
            got = [torch.cat([getattr(p, n) for p in parts]) for n in ("batch_node_value", "batch_node_type", "batch_subtree_size")]
            same(got, ref, f"compact table, G = {G}, cap {cap_mode}")
    # and the torch composition of the same step is a valid population of the same shape (different random words)
    sg = ShardedGeneticProgramming(full, 0.2, desc.update(max_layer_cnt=3), make(), seed=5)
    nxt = sg.next_slice_torch(full, fitness, 0, pop)
    assert nxt.pop_size == pop and int(nxt.batch_subtree_size[:, 0].min()) >= 1


def test_genetic_programming_step_with_tournament_selection_takes_the_fused_path(g):
    """GeneticProgramming.step with TournamentSelection + DefaultCrossover + DefaultMutation runs selection -> one breeding pass
    (evogp_hip_breed_lists) and improves the population; the elites are copied verbatim"""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, GeneticProgramming
    from evogp_amd.algorithm.selection import TournamentSelection
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5,
                              const_samples=[-1, 0, 1])
    pop = 6000
    forest = Forest.random_generate(pop, desc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
    sel = TournamentSelection(4, best_probability=0.95, replace=False, survivor_rate=0.5, elite_rate=0.005)
    algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=3)), sel)
    assert algo._native_plan() is not None
    X = torch.rand(256, 4, device=dev) * 4 - 2
    y = (X[:, 0] * X[:, 1] - X[:, 2] / (X[:, 3] * X[:, 3] + 1.5) + 0.5 * X[:, 0]).unsqueeze(1)    # not in generation 0
    best, median = [], []
    for _ in range(8):
        fit = -algo.forest.SR_fitness(X, y)
        fit[torch.isnan(fit)] = -torch.inf
        best.append(float(fit.max())); median.append(float(fit.median()))
        top = str(algo.forest[int(torch.argmax(fit))])
        new = algo.step(fit)
        assert top in {str(new[i]) for i in range(sel.counts(pop)[0])}
        sizes = new.batch_subtree_size[:, 0]
        assert int(sizes.min()) >= 1 and int(sizes.max()) <= 64
    assert best[-1] >= best[0] and median[-1] > median[0], (best, median)     # elites keep the best; the tournaments move the population


def test_genetic_programming_step_with_lists_the_fused_pass_cannot_take(g):
    """A user selection may hand back index tensors on the CPU, or a float64 fitness may arrive: the fused step moves the lists to
    the forest's device, and leaves a fitness vector it would have to cast (ties of a float64 ranking) to the composed operators
    -- in both cases a valid next generation with the elites copied verbatim."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=3, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    pop = 3000

    class CpuLists:   # the shape of selection/*.py operators: (elite indices, survivor indices) -- here on the CPU
        def __call__(self, forest, fitness):
            order = torch.argsort(fitness, descending=True).cpu()
            return order[:30], order[:900]

    X = torch.rand(128, 3, device=dev) * 4 - 2
    y = (X[:, 0] * X[:, 1] - X[:, 2]).unsqueeze(1)
    for selection, dtype in ((CpuLists(), torch.float32), (DefaultSelection(0.3, elite_rate=0.01), torch.float64)):
        forest = Forest.random_generate(pop, desc, keys=torch.tensor([5, 6], dtype=torch.uint32, device=dev))
        algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=3)), selection)
        fit = -forest.SR_fitness(X, y)
        fit[torch.isnan(fit)] = -torch.inf
        top = str(forest[int(torch.argmax(fit))])
        new = algo.step(fit.to(dtype))
        assert new.pop_size == pop
        assert top in {str(new[i]) for i in range(30)}
        sizes = new.batch_subtree_size[:, 0]
        assert int(sizes.min()) >= 1 and int(sizes.max()) <= 64
        assert torch.isfinite(new.SR_fitness(X, y)).any()


def test_select_survivors_equals_the_sets_of_a_stable_sort(g):
    """csrc/select.hip (exact radix select + compaction in one cooperative launch) against its definition in torch ops
    (evogp_amd.parallel.select_order on the CPU): the n_elite best, then the other survivors, each group by ascending index; ties at
    a threshold to the lower index; NaN worst; every size from one value to a million, heavy ties, infinities."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.parallel import select_order

    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(11)
    cases = []
    for n in (1, 2, 3, 255, 256, 257, 1000, 4097, 65536, 100_000, 333_333, 1_000_003):
        x = torch.randn(n, generator=gen)
        cases.append((x, (0, 1, n // 100, n // 3)))
        q = torch.round(torch.randn(n, generator=gen) * 3)          # few distinct values: long runs of ties at every threshold
        q[torch.rand(n, generator=gen) < 0.05] = float("nan")
        q[torch.rand(n, generator=gen) < 0.05] = float("-inf")
        q[torch.rand(n, generator=gen) < 0.01] = float("inf")
        cases.append((q, (0, 1, n // 100, n // 3)))
        cases.append((torch.zeros(n), (0, n // 2)))               # all equal
        z = torch.zeros(n); z[::2] = -0.0; z[::7] = float("nan")    # -0 = +0 (ties by index), NaN behind them
        cases.append((z, (0, n // 3)))
        cases.append((-torch.arange(n, dtype=torch.float32) * 1e-3 - 1e-40, (1,)))   # descending incl. denormal steps near zero
    for x, elites in cases:
        n = x.shape[0]
        xd = x.to(dev)
        # the ranking the kernel implements: value descending, NaN behind everything (also behind -inf), equal values by index
        nan = torch.isnan(x)
        by_value = torch.sort(torch.where(nan, torch.full_like(x, float("-inf")), x), descending=True, stable=True).indices
        rank = by_value[torch.sort(nan[by_value].to(torch.int8), stable=True).indices]
        assert torch.equal(select_order(x, 3 if n > 3 else 0, n).long().sort().values, torch.arange(n))   # (the CPU definition: a permutation)
        for n_elite in elites:
            for n_keep in sorted({max(n_elite, 1), max(n_elite, n * 3 // 10, 1), n}):
                got = torch.ops.evogp_hip.select_survivors(xd, n_elite, n_keep).cpu()
                want = torch.cat([torch.sort(rank[:n_elite]).values, torch.sort(rank[n_elite:n_keep]).values]).to(torch.int32)
                assert torch.equal(got, want), (n, n_elite, n_keep, int((got != want).sum()))
                assert torch.equal(select_order(x, n_elite, n_keep), want)   # the torch definition used off the GPU agrees, NaN and -0 included


def test_tournament_kernel_equals_the_torch_formulation(g):
    """csrc/select.hip tournament_kernel (TournamentSelection with the reference's default arguments: contenders with replacement
    from the counter-based words, the best one wins) against the same selection written in torch ops on the CPU: the same winners,
    ties to the first contender drawn, NaN never preferred; and the elites are the best trees"""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm.selection import TournamentSelection

    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(3)
    for n, t, seed, generation in ((1000, 3, 0, 0), (100_000, 20, 1234, 7), (1_000_003, 5, 2**40 + 1, 99), (17, 50, 5, 5)):
        fit = torch.round(torch.randn(n, generator=gen) * 4) / 4                  # many ties
        fit[torch.rand(n, generator=gen) < 0.05] = float("nan")
        fit[torch.rand(n, generator=gen) < 0.02] = float("-inf")
        sel = TournamentSelection(t, survivor_rate=0.5, elite_rate=0.01)
        e_cpu, p_cpu = sel.counter_based(fit, seed, generation)
        e_gpu, p_gpu = sel.counter_based(fit.to(dev), seed, generation)
        assert torch.equal(p_gpu.cpu(), p_cpu), (n, t, int((p_gpu.cpu() != p_cpu).sum()))
        assert torch.equal(e_gpu.cpu(), e_cpu)
        assert p_cpu.numel() == n // 2 and int(p_cpu.min()) >= 0 and int(p_cpu.max()) < n
        won = fit[p_cpu.long()]
        assert float(torch.nan_to_num(won, nan=-1e9, neginf=-1e9).mean()) > float(torch.nan_to_num(fit, nan=-1e9, neginf=-1e9).mean())
    assert TournamentSelection(4, best_probability=0.9).counter_based(fit, 0, 0) is None          # not the default arguments:
    assert TournamentSelection(4, replace=False).counter_based(fit, 0, 0) is None                 # the operator's own torch program


def test_native_random_words_equal_the_python_definition(g):
    """csrc/breed.hip random_words_kernel vs evogp_amd/parallel.py random_words (the CPU paths and the gloo test use the latter):
    the same hash of (seed, generation, word, offspring), a slice equals the same columns of the whole"""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.parallel import random_words

    dev = torch.device("cuda", 0)
    for seed, gen, n, lo, hi in ((0, 0, 1000, 0, 1000), (1234, 7, 99_000, 12_345, 40_000), (2**40 + 3, 123456, 5000, 4999, 5000)):
        nat = torch.ops.evogp_hip.random_words(seed, gen, 6, n, lo, hi, dev)
        ref = random_words(seed, gen, 6, lo, hi, "cpu")
        assert torch.equal(nat[:, lo:hi].cpu(), ref)
        assert int(ref.min()) >= 0 and int(ref.max()) < 2**31 - 1
    assert abs(float(random_words(5, 5, 6, 0, 200_000, "cpu").float().mean()) / 2**31 - 0.5) < 0.005


def test_hashed_words_equal_the_array_forms(g, oracle):
    """evogp_hip_generate_masked_hashed / evogp_hip_breed_lists_hashed compute the counter-based words in the kernels; fed the
    words evogp_hip_random_words writes for the same (seed, generation), the array forms must build the same rows bit for bit —
    whole population and a rank's slice with its tree-index offset."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.parallel import random_words
    from evogp_amd.tree import GenerateDescriptor

    dev = torch.device("cuda", 0)
    for L, funcs in ((64, ["+", "-", "*", "/"]), (50, ["+", "*", "sin", "if"])):      # (50: the one-row-per-wave kernels)
        desc = GenerateDescriptor(max_tree_len=L, input_len=5, output_len=1, using_funcs=funcs, max_layer_cnt=3 if "if" in funcs else 5,
                                  const_samples=[-1.0, 0.5, 2.0])
        d = desc.update(max_layer_cnt=3)
        pop, n_elite, n_surv, seed, generation = 6000, 60, 1800, 1234567, 42
        forest = oracle.generate(pop, L, 5, 1, 0.5, 0.5, [3, 4], np.asarray(desc.depth2leaf_probs.cpu()), np.asarray(desc.roulette_funcs.cpu()), [-1.0, 0.5, 2.0])
        v, t, s = (torch.from_numpy(a).to(dev) for a in forest)
        gen = torch.Generator().manual_seed(L)
        elites = torch.randperm(pop, generator=gen)[:n_elite].to(torch.int32).to(dev)
        parents = torch.randint(0, pop, (n_surv,), generator=gen).to(torch.int32).to(dev)
        n_new = pop - n_elite
        below = int(0.3 * (2**31 - 1))
        rnd = torch.ops.evogp_hip.random_words(seed, generation, 6, n_new, 0, n_new, dev)
        assert torch.equal(rnd.cpu(), random_words(seed, generation, 6, 0, n_new, "cpu"))
        keys = (torch.ops.evogp_hip.random_words(seed, generation, 8, 2, 0, 2, dev)[7] % 1000000).to(torch.uint32)
        args = (L, d.input_len, d.output_len, d.const_samples.shape[0], d.out_prob, d.const_prob)
        tabs = (d.depth2leaf_probs, d.roulette_funcs, d.const_samples)
        for lo, hi in ((0, pop), (1000, 2500), (0, 40), (30, 100)):     # whole population; slices inside, of elites only, across the elite border
            o_lo, o_hi = max(lo, n_elite) - n_elite, max(hi, n_elite) - n_elite
            rows = hi - lo
            if o_hi > o_lo:
                don_a = torch.ops.evogp_hip.tree_generate_masked(o_hi - o_lo, *args, keys, *tabs, o_lo, rnd[4, o_lo:o_hi].contiguous(), below)
                don_h = torch.ops.evogp_hip.tree_generate_masked_hashed(o_hi - o_lo, *args, *tabs, o_lo, seed, generation, below)
                act = rnd[4, o_lo:o_hi].to(torch.int64) < below
                assert 0.2 < float(act.float().mean()) < 0.4
                for a, b in zip(don_a, don_h):
                    assert torch.equal(a[act].view(torch.uint8), b[act].view(torch.uint8)), "donors differ"
            else:
                don_a = don_h = tuple(torch.zeros((rows, L), dtype=dt, device=dev) for dt in (torch.float32, torch.int16, torch.int16))
            out_a = torch.ops.evogp_hip.breed_rows(pop, L, v, t, s, elites, parents, rnd, below, *don_a, lo, rows)
            out_h = torch.ops.evogp_hip.breed_rows_hashed(pop, L, v, t, s, elites, parents, seed, generation, below, *don_h, lo, rows)[:3]
            for a, b in zip(out_a, out_h):
                assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (L, lo, hi)
