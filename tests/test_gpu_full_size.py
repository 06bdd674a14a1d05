"""GPU: BASELINE.json's configurations at their FULL sizes, through properties that do not depend on the size plus a sample against
the oracle (VERDICT r03 "missing" #2 and #3):

  * configs[3] — classifier trees, pop 200 000, 10 outputs, max_tree_len 128, sklearn's digits shape (1797 rows x 64 features:
    four LDS pieces per fitness pass): the fused arg-max count of the whole population equals that of its two halves, and a
    2 000-tree sample equals torch's argmax(clip(softmax)) count on the oracle's outputs;
  * configs[4] — policy trees, pop 50 000, 17 observations, 6 actions, max_tree_len 256, 1000 steps per episode: the rollout of
    the whole population (prepared forward pass inside a replayed HIP graph) against the oracle's loop on a sample of trees;
  * the reference's own SR script shape (example/uci_sr.py:45-75): max_tree_len 512, max_layer_cnt 9, 10 000 constants in [-5, 5],
    functions + - * / sin cos tan, layer_leaf_prob 0.3, pop 100 000 x 1024 rows: a 5 000-tree sample against the oracle, the
    population against its halves;
  * the headline itself (round 6): all 1 M trees x 1024 rows of BASELINE's north_star against the oracle on the host's cores."""
import numpy as np
import pytest

from helpers import ARITH, PAPER7, assert_close_classes, c2_dataset, depth2leaf, per_tree_tolerance, roulette_uniform, torch_rule_counts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


def test_configs3_classifier_population_at_full_size(g, oracle):
    rng = np.random.default_rng(3)
    pop, L, var_len, out_len, D = 200_000, 128, 64, 10, 1797
    f = g.generate(pop, L, var_len, out_len, 0.5, 0.5, [7, 0], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
    try:
        from sklearn.datasets import load_digits

        d = load_digits()
        X, labels = d.data.astype(np.float32), d.target.astype(np.int32)
    except Exception:  # (the shape is what counts)
        X, labels = rng.integers(0, 17, (D, var_len)).astype(np.float32), rng.integers(0, out_len, D).astype(np.int32)
    assert X.shape == (D, var_len)
    full = g.batch_argmax_count(*f, X, labels, out_len)
    assert full.min() >= 0 and full.max() <= D
    h = pop // 2
    halves = np.concatenate([g.batch_argmax_count(*(a[:h] for a in f), X, labels, out_len),
                             g.batch_argmax_count(*(a[h:] for a in f), X, labels, out_len)])
    assert np.array_equal(full, halves), (np.flatnonzero(full != halves)[:5], "the work distribution shows in the counts")
    pick = np.sort(rng.choice(pop, 2000, replace=False))
    sub = tuple(a[pick] for a in f)
    want = torch_rule_counts(oracle.batch_evaluate(*sub, X, out_len), labels)
    assert np.array_equal(full[pick], want), (np.abs(full[pick] - want).max(), (full[pick] != want).mean())
    # ... and the same sample as a launch of its own (another number of trees per workgroup, no dynamic tail)
    assert np.array_equal(g.batch_argmax_count(*sub, X, labels, out_len), want)


def test_configs4_policy_rollout_at_full_size(g, oracle):
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.problem import RolloutProblem
    from evogp_amd.tree import Forest

    rng = np.random.default_rng(4)
    pop, obs_dim, act_dim, steps = 50_000, 17, 6, 1000

    class Env:   # elementwise (no matrix product): the two sides differ only in the order of two small sums
        def __init__(self, device=None):
            self.device, self.obs_dim, self.act_dim = device, obs_dim, act_dim
            self.x0 = torch.linspace(-1, 1, obs_dim).to(device)

        def reset(self, n):
            return self.x0[None, :].repeat(n, 1)

        def observe(self, state):
            return state

        def step(self, state, action):
            push = torch.cat([action, action, action[:, :obs_dim - 2 * act_dim]], dim=1)
            nxt = 0.95 * state + 0.1 * push
            reward = -(nxt * nxt).sum(1) - 0.1 * (action * action).sum(1)
            return nxt, reward, nxt.abs().amax(1) > 5.0

    cs = np.linspace(-1, 1, 100).astype(np.float32)
    f = g.generate(pop, 256, obs_dim, act_dim, 0.5, 0.5, [7, 0], depth2leaf(6), roulette_uniform(ARITH), cs)
    forest = Forest(obs_dim, act_dim, *(torch.from_numpy(a).cuda() for a in f))
    clamp = lambda a: a.clamp(-1, 1)  # noqa: E731
    got = RolloutProblem(Env("cuda"), steps, output_transform=clamp, use_graph=True).evaluate(forest).cpu().numpy()
    assert forest._prepared is not None, "the rollout must run from the operation lists"
    assert got.shape == (pop,) and np.isfinite(got).all()
    pick = np.sort(rng.choice(pop, 300, replace=False))
    sub = tuple(a[pick] for a in f)
    state = np.tile(np.linspace(-1, 1, obs_dim, dtype=np.float32)[None, :], (len(pick), 1))
    total = np.zeros(len(pick), np.float32); done = np.zeros(len(pick), bool)
    for _ in range(steps):
        with np.errstate(all="ignore"):
            action = np.clip(oracle.evaluate(*sub, state, act_dim), -1, 1).astype(np.float32)
            action = np.where(np.isnan(action), np.float32(np.nan), action)
            push = np.concatenate([action, action, action[:, :obs_dim - 2 * act_dim]], 1)
            nxt = (np.float32(0.95) * state + np.float32(0.1) * push).astype(np.float32)
            reward = (-(nxt * nxt).sum(1) - np.float32(0.1) * (action * action).sum(1)).astype(np.float32)
            now_done = np.abs(nxt).max(1) > 5.0
            reward = np.nan_to_num(reward, nan=-1e6, posinf=-1e6, neginf=-1e6).astype(np.float32)
            total = total + np.where(done, np.float32(0), reward)
            done = done | now_done | ~np.isfinite(nxt).all(1)
            state = np.where(done[:, None], state, np.nan_to_num(nxt)).astype(np.float32)
    assert np.allclose(got[pick], total, rtol=5e-4, atol=1e-2), np.abs(got[pick] - total).max()
    # a sample evaluated as a forest of its own gives the same episodes (per-tree environments: nothing depends on the neighbours)
    small = Forest(obs_dim, act_dim, *(torch.from_numpy(a).cuda() for a in sub))
    again = RolloutProblem(Env("cuda"), steps, output_transform=clamp, use_graph=True).evaluate(small).cpu().numpy()
    assert np.array_equal(again.view(np.uint32), got[pick].view(np.uint32))


def uci_sr_forest(g, pop, keys=(42, 0)):
    """the forest of example/uci_sr.py:45-54 on a 10-variable problem: max_tree_len 512, max_layer_cnt 9, layer_leaf_prob 0.3, seven
    functions, 10 000 constants drawn from U(-5, 5) (descriptor.py: const_range / sample_cnt)"""
    d2l = np.array([0.3] * 8 + [1.0] * 2, np.float32)
    cs = np.random.default_rng(5).uniform(-5, 5, 10_000).astype(np.float32)
    return g.generate(pop, 512, 10, 1, 0.5, 0.5, list(keys), d2l, roulette_uniform(PAPER7), cs)


def test_uci_sr_script_shape_against_the_oracle(g, oracle):
    rng = np.random.default_rng(6)
    pop = 100_000
    f = uci_sr_forest(g, pop)
    lens = f[2][:, 0]
    assert lens.max() > 64 and lens.max() <= 511, "the long-row paths are not exercised"
    X, y = c2_dataset()
    full = g.sr_fitness(*f, X, y)
    h = pop // 2
    halves = np.concatenate([g.sr_fitness(*(a[:h] for a in f), X, y), g.sr_fitness(*(a[h:] for a in f), X, y)])
    same = (full.view(np.uint32) == halves.view(np.uint32)) | (np.isnan(full) & np.isnan(halves))
    assert same.all(), (np.flatnonzero(~same)[:5], "the work distribution shows in the fitness words")
    pick = np.sort(rng.choice(pop, 5000, replace=False))
    sub = tuple(a[pick] for a in f)
    # (1) against the register interpreters on the same device (the same math library): the per-row outputs of batch_evaluate, reduced
    #     in float64, must give the fused fitness to 1e-4 with identical NaN / inf classes -- EVERY tree of the sample
    pred = g.batch_evaluate(*sub, X, 1)[:, :, 0]
    with np.errstate(all="ignore"):
        d = pred - y[:, 0][None, :]
        tot = (d * d).astype(np.float64).sum(1)
        ref = np.where(tot > np.finfo(np.float32).max, np.inf, tot / X.shape[0]).astype(np.float32)
    assert_close_classes(full[pick], ref, 1e-4, what="uci_sr shape: threaded code vs register interpreters")
    # (2) against the oracle (the host's math library) within every tree's own sensitivity to 3-ulp differences of the library calls.
    #     tan(tan(x)) next to a pole can move a single row by any amount for a fourth ulp, which the probe of a few seeds does not always
    #     see: entries beyond their granted tolerance must be rare (1 in 1000), and check (1) holds for them like for all others
    want, tol, unstable = per_tree_tolerance(oracle, sub, X, y)
    got = full[pick].astype(np.float64)
    stable = ~unstable
    assert unstable.mean() <= 0.1
    assert np.array_equal(np.isnan(got[stable]), np.isnan(want[stable])) and np.array_equal(np.isinf(got[stable]), np.isinf(want[stable]))
    fin = stable & np.isfinite(want)
    beyond = np.abs(got[fin] - want[fin].astype(np.float64)) > tol[fin]
    assert beyond.mean() <= 1e-3, (int(beyond.sum()), int(fin.sum()))


def test_evolved_uci_sr_population_stays_in_the_threaded_code(g, oracle):
    """30 generations of example/uci_sr.py's own operators (DefaultCrossover, DefaultMutation(0.1, max_layer_cnt 4), TournamentSelection(20,
    0.5, 0.1)) on a population of 6 000 trees with max_tree_len 512: the rows fill up, 40 % of the trees need more operand-stack entries
    than the interpreter has in its own evaluation order, some evaluate sin / cos / tan of 2^17 and more.  Round 3 left half of such a
    population to the scratch-stack register kernel (172 ms per call at 100 k trees); now (compile_general's reordering pass, the
    library's whole trigonometric functions inside the threaded code) at most 1 % may be left, and every fitness word must be what the
    register interpreters return for the tree (batch_evaluate: the same math library, reduced in float64) -- and the oracle's within the
    trees' own sensitivity for all but 1 in 200."""
    import ctypes

    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd import _lib
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, GeneticProgramming
    from evogp_amd.algorithm.selection import TournamentSelection
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    pop = 6000
    desc = GenerateDescriptor(max_tree_len=512, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"],
                              max_layer_cnt=9, const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
    X, y = c2_dataset()
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    algo = GeneticProgramming(Forest.random_generate(pop, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)),
                              DefaultCrossover(), DefaultMutation(0.1, desc.update(max_layer_cnt=4)), TournamentSelection(20, 0.5, 0.1))
    neg = torch.full((pop,), float("-inf"), device=dev)
    for _ in range(30):
        fit = -algo.forest.SR_fitness(Xd, yd)
        algo.step(torch.where(torch.isnan(fit), neg, fit))
    f = algo.forest
    trees = (f.batch_node_value.cpu().numpy(), f.batch_node_type.cpu().numpy(), f.batch_subtree_size.cpu().numpy())
    lens = trees[2][:, 0]
    assert lens.mean() > 100 and lens.max() > 400, (lens.mean(), lens.max(), "the population did not grow: not the evolved case")
    _lib.check(_lib.lib.evogp_hip_debug_profile(2), "profile")          # the call stops behind the threaded code: marked trees keep their sentinel
    words = f.SR_fitness(Xd, yd).view(torch.int32)
    _lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile")
    left = ((words == 0x7FC0FEED) | (words == 0x7FC0BEEF) | (words == 0x7FC0DEED)).float().mean().item()
    assert left <= 0.01, f"{left:.3f} of the evolved trees were left to the register kernels"
    got = g.sr_fitness(*trees, X, y)
    assert np.array_equal(got.view(np.uint32), f.SR_fitness(Xd, yd).cpu().numpy().view(np.uint32)) or np.allclose(
        np.nan_to_num(got, nan=-1.0, posinf=-2.0), np.nan_to_num(f.SR_fitness(Xd, yd).cpu().numpy(), nan=-1.0, posinf=-2.0), rtol=1e-6)
    pred = g.batch_evaluate(*trees, X, 1)[:, :, 0]
    with np.errstate(all="ignore"):
        d = pred - y[:, 0][None, :]
        tot = (d * d).astype(np.float64).sum(1)                  # squares in fp32 (same overflow), sum in fp64 ...
        ref = np.where(tot > np.finfo(np.float32).max, np.inf, tot / X.shape[0]).astype(np.float32)   # ... which overflows where the kernel's fp32 sum does
    assert_close_classes(got, ref, 1e-4, what="evolved uci_sr population: threaded code vs register interpreters")
    pick = np.sort(np.random.default_rng(1).choice(pop, 1500, replace=False))
    sub = tuple(a[pick] for a in trees)
    want, tol, unstable = per_tree_tolerance(oracle, sub, X, y)
    stable = ~unstable
    fin = stable & np.isfinite(want) & np.isfinite(got[pick])
    beyond = np.abs(got[pick][fin].astype(np.float64) - want[fin].astype(np.float64)) > tol[fin]
    assert beyond.mean() <= 5e-3, (int(beyond.sum()), int(fin.sum()))
    # the same trees ten times over (60 000 trees: several rounds of the long-tree compiler's workgroups, whose waves run out of step --
    # round 5 found marked trees that no wave took there): every copy gets its tree's word, from the straight-line compiler of the long
    # trees (csrc/sr_tc.hip tc_compile_long_kernel) and from the general compiler's passes
    big = tuple(np.tile(a, (10, 1)) for a in trees)
    try:
        for fast in (1, 0):
            assert _lib.lib.evogp_hip_debug_long_compiler(fast) == 0
            w = g.sr_fitness(*big, X, y).view(np.uint32).reshape(10, pop)
            bad = np.nonzero((w != got.view(np.uint32)[None, :]).any(0))[0]
            assert len(bad) == 0, (fast, len(bad), bad[:5], w[:, bad[:3]], got.view(np.uint32)[bad[:3]])
    finally:
        _lib.lib.evogp_hip_debug_long_compiler(-1)


def test_headline_population_at_full_size_against_the_oracle(g, oracle):
    """BASELINE north_star / configs[2]'s population in full -- 1 M trees x 1024 datapoints, + - * / -- on the GPU against the plain-C oracle
    on the host's cores (OpenMP over trees: seconds): the same NaN and inf sets, every finite fitness within 1e-5 (the order of the sum
    over the rows is the only difference; scripts/dbg/headline_vs_oracle.py prints the figures, profiles/r06Z_headline_vs_oracle.log)."""
    import os, sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        import bench
    finally:
        sys.argv = argv
    dev = torch.device("cuda", 0)
    forest, Xd, yd, X, y = bench.sr_inputs(0, 1_000_000, dev)
    got = forest.SR_fitness(Xd, yd, True, "auto").cpu().numpy().astype(np.float64)
    want = oracle.sr_fitness(forest.batch_node_value.cpu().numpy(), forest.batch_node_type.cpu().numpy(), forest.batch_subtree_size.cpu().numpy(),
                             X, y, True, 0).astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    rel = np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-30)
    assert fin.sum() > 600_000 and rel.max() <= 1e-5, float(rel.max())


def test_headline_forest_is_the_oracles_bit_for_bit(g, oracle):
    """tree_generate at the headline's size -- 1 M rows of 64 nodes, keys [42, 0], the descriptor of bench.sr_inputs -- against the oracle's
    generator: all three arrays equal, zero tails included (the staged kernel: three waves per SIMD, byte-sized LDS arrays, sizes read back
    without the row length -- DESIGN.md section 3.4a)."""
    pop = 1_000_000
    args = (pop, 64, 10, 1, 0.0, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
    gv, gt, gs = g.generate(*args)
    ov, ot, os_ = oracle.generate(*args)
    assert np.array_equal(gt, ot) and np.array_equal(gs, os_)
    assert np.array_equal(gv.view(np.uint32), ov.view(np.uint32))


def test_crossover_and_mutate_at_the_headline_size_bit_for_bit(g, oracle):
    """tree_crossover (300 k parents -> 990 k children) and tree_mutate (1 M trees, every one with a donor) on rows of 64 nodes against
    the oracle: node indices inside and outside the live trees, right parents out of range (the copy-left fallbacks of mutation.cu:150-180,
    :256-289), children that would overflow the row."""
    rng = np.random.default_rng(6)
    pop = 1_000_000
    v, t, s = oracle.generate(pop, 64, 10, 1, 0.0, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
    n_new, n_par = 990_000, 300_000
    li = rng.integers(0, n_par, n_new).astype(np.int32); ri = rng.integers(-2, n_par + 2, n_new).astype(np.int32)
    sizes = s[:, 0].astype(np.int64)
    ln = (rng.integers(0, 1 << 30, n_new) % sizes[li]).astype(np.int32); rn = (rng.integers(0, 1 << 30, n_new) % sizes[np.clip(ri, 0, pop - 1)]).astype(np.int32)
    odd = rng.random(n_new) < 0.02
    ln[odd] = rng.integers(-3, 70, int(odd.sum())).astype(np.int32)          # outside the live tree, outside the row
    got = g.crossover(v, t, s, li, ri, ln, rn)
    want = oracle.crossover(v, t, s, li, ri, ln, rn)
    for a, b, name in zip(got, want, ("value", "type", "size")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"crossover at 1 M: {name} differs"
    dv, dt, ds = oracle.generate(pop, 64, 10, 1, 0.0, 0.5, [7, 9], depth2leaf(4), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
    idx = (rng.integers(0, 1 << 30, pop) % sizes).astype(np.int32)
    idx[rng.random(pop) < 0.02] = -1                                         # the copy-old fallback
    got = g.mutate(v, t, s, idx, dv, dt, ds)
    want = oracle.mutate(v, t, s, idx, dv, dt, ds)
    for a, b, name in zip(got, want, ("value", "type", "size")):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"mutate at 1 M: {name} differs"


@pytest.mark.parametrize("out_len", [1, 6])
def test_tree_evaluate_at_the_headline_size_bit_for_bit(g, oracle, out_len):
    """tree_evaluate -- one input row per tree (forward.cu:304-351) -- on 1 M trees of + - * /: single-output trees through the lane
    kernels, six outputs through the direct kernel (evaluate_prepared.hip); every result word equal to the oracle's (the arithmetic of
    the register interpreters and of the OUT-node reading is IEEE operation for operation)."""
    rng = np.random.default_rng(60 + out_len)
    pop, L, var_len = 1_000_000, 64, 10
    v, t, s = oracle.generate(pop, L, var_len, out_len, 0.5 if out_len > 1 else 0.0, 0.5, [42, out_len], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
    x = rng.uniform(-5, 5, (pop, var_len)).astype(np.float32)
    got = g.evaluate(v, t, s, x, out_len)
    want = oracle.evaluate(v, t, s, x, out_len)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(np.where(nan, 0, got).view(np.uint32), np.where(nan, 0, want).view(np.uint32))
