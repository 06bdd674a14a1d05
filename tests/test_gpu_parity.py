"""GPU parity tests: the HIP engine, called THROUGH THE C ABI, against the CPU oracle and the
committed golden vectors.

Bars (BASELINE.json north_star): tree encodings from generate / crossover / mutate are BIT-EXACT;
SR fitness and evaluation agree within 1e-5 relative (fp32) on the arithmetic function set —
tree_evaluate / batch_evaluate are bit-exact per datapoint there (IEEE + - * / on both sides); in
tree_SR_fitness the order of the final summation differs and the default division is the
faithfully rounded short sequence (the ieee mode is tested as well) — and within
the transcendental-library tolerance stated below for the
sin/cos/exp/pow families (device OCML vs host glibc: a few ulp per call, amplified by the tree).
"""
import glob
import os

import numpy as np
import pytest

from helpers import (ALLF, ARITH, PAPER7, assert_close_classes, assert_forest_equal, assert_within_sensitivity, bits, c2_dataset,
                     depth2leaf, fbits, per_tree_tolerance, random_crossover_indices, roulette_uniform, sensitivity, the_oracle, torch_rule_counts)

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
BATTERIES = sorted(glob.glob(os.path.join(GOLD, "battery_*.npz")))
CS3 = [-1.0, 0.0, 1.0]

RTOL_ARITH = 1e-5   # north_star bar
ARITH_MASK = 0b11110   # function ids 1..4 (+ - * /): the mask a Forest generated from such a descriptor carries


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


# ---- generate: bit-exact ---------------------------------------------------------------------
@pytest.mark.parametrize("funcs,out_len,var_len,L,mlc", [
    (ARITH, 1, 10, 64, 6), (ARITH, 1, 3, 32, 4), (PAPER7, 1, 5, 64, 6), (ALLF, 1, 4, 128, 5),
    (ARITH, 3, 6, 64, 6), (ALLF, 4, 3, 128, 5), (ARITH, 1, 2, 1024, 9), ([0], 2, 3, 1024, 6),
    # trees that outgrow their row by far (IF only, six levels: up to 1093 nodes in rows of 100 / 254 / 255: subtree sizes beyond a byte)
    ([0], 1, 3, 100, 6), (ALLF, 2, 4, 254, 7), ([0], 1, 3, 255, 6)])
def test_generate_bit_exact(g, oracle, funcs, out_len, var_len, L, mlc):
    rou, d2l = roulette_uniform(funcs), depth2leaf(mlc)
    cs = np.array([-1, 0, 1, 0.5, 2], np.float32)
    for pop, keys in ((1, [1, 2]), (63, [42, 0]), (1000, [123456, 654321]), (4097, [2**32 - 1, 7])):
        want = oracle.generate(pop, L, var_len, out_len, 0.37, 0.61, keys, d2l, rou, cs)
        got = g.generate(pop, L, var_len, out_len, 0.37, 0.61, keys, d2l, rou, cs)
        assert_forest_equal(got, want, f"generate pop={pop}")  # full rows: tails are zero on both sides


def test_generate_tree_index_offset_makes_shards_identical(g, oracle):
    rou, d2l = roulette_uniform(ARITH), depth2leaf(6)
    full = g.generate(1000, 64, 10, 1, 0.5, 0.5, [42, 0], d2l, rou, CS3)
    parts = [g.generate(250, 64, 10, 1, 0.5, 0.5, [42, 0], d2l, rou, CS3, offset=250 * r) for r in range(4)]
    assert_forest_equal(tuple(np.concatenate([p[i] for p in parts]) for i in range(3)), full, "sharded generate")
    assert_forest_equal(full, oracle.generate(1000, 64, 10, 1, 0.5, 0.5, [42, 0], d2l, rou, CS3), "vs oracle")


def test_generate_extreme_probabilities(g, oracle):
    rou = roulette_uniform(ARITH)
    for d2l in (np.array([0.0] * 5 + [1.0] * 5, np.float32), np.array([1.0] * 10, np.float32)):
        for cp in (0.0, 1.0):
            want = oracle.generate(300, 64, 3, 1, 0.0, cp, [8, 9], d2l, rou, CS3)
            got = g.generate(300, 64, 3, 1, 0.0, cp, [8, 9], d2l, rou, CS3)
            assert_forest_equal(got, want, "generate extremes")


# ---- golden batteries (reference outputs) -------------------------------------------------------
@pytest.mark.parametrize("path", BATTERIES, ids=[os.path.basename(p)[8:-4] for p in BATTERIES])
def test_golden_battery(g, path):
    z = np.load(path)
    out_len, var_len, L = int(z["out_len"]), int(z["var_len"]), int(z["gp_len"])
    pop = z["value"].shape[0]
    trans = any(int(f) not in (1, 2, 3, 4) for f in z["funcs"])
    forest = g.generate(pop, L, var_len, out_len, 0.5, 0.5, z["keys"], z["depth2leaf"], z["roulette"], z["consts"])
    assert_forest_equal(forest, (z["value"], z["type"], z["size"]), "generate", live_only=True)
    cr = g.crossover(*forest, z["left_idx"], z["right_idx"], z["left_node"], z["right_node"])
    assert_forest_equal(cr, (z["cross_value"], z["cross_type"], z["cross_size"]), "crossover", live_only=True)
    mu = g.mutate(*forest, z["mut_idx"], z["new_value"], z["new_type"], z["new_size"])
    assert_forest_equal(mu, (z["mut_value"], z["mut_type"], z["mut_size"]), "mutate", live_only=True)
    rt = RTOL_ARITH
    ev = g.evaluate(*forest, z["eval_x"], out_len)
    if trans:
        # library functions: every entry within 1e-5 plus what a 3-ulp perturbation of ITS library calls does (no allowed-bad
        # fraction); the golden values are the reference's own outputs, which the oracle reproduces bit for bit
        gold = (z["value"], z["type"], z["size"])
        oracle = the_oracle()
        want, tol, unst = sensitivity(oracle, lambda: oracle.evaluate(*gold, z["eval_x"], out_len))
        assert np.array_equal(fbits(want), fbits(z["eval_out"])), "oracle != golden vector"
        assert_within_sensitivity(ev, want, tol, unst, "evaluate")
        for mse, key in ((True, "sr_mse"), (False, "sr_mae")):
            want, tol, unst = per_tree_tolerance(oracle, gold, z["sr_x"], z["sr_y"], use_mse=mse)
            assert np.array_equal(fbits(want), fbits(z[key])), "oracle != golden vector"
            assert_within_sensitivity(g.sr_fitness(*forest, z["sr_x"], z["sr_y"], mse), want, tol, unst, key)
    else:
        assert np.array_equal(fbits(ev), fbits(z["eval_out"])), "arithmetic evaluation must be bit-exact"
        assert_close_classes(g.sr_fitness(*forest, z["sr_x"], z["sr_y"], True), z["sr_mse"], rt, what="sr mse")
        assert_close_classes(g.sr_fitness(*forest, z["sr_x"], z["sr_y"], False), z["sr_mae"], rt, what="sr mae")


# ---- crossover / mutate: bit-exact on fuzzed inputs ----------------------------------------------
@pytest.mark.parametrize("L,mlc,funcs", [(64, 6, ARITH), (32, 4, ARITH), (128, 5, ALLF), (1024, 9, ARITH), (200, 5, [0, 1, 14]),
                                         (50, 5, ARITH), (33, 4, ALLF)])  # row lengths that are no multiple of 4: one row per wave
def test_crossover_mutate_bit_exact(g, oracle, rng, L, mlc, funcs):
    rou = roulette_uniform(funcs)
    pop = 3000
    forest = oracle.generate(pop, L, 5, 1, 0.5, 0.5, [17, 4], depth2leaf(mlc), rou, CS3)
    sizes = forest[2][:, 0].astype(np.int64)
    idx = random_crossover_indices(rng, sizes, 7000)
    assert_forest_equal(g.crossover(*forest, *idx), oracle.crossover(*forest, *idx), "crossover")
    new = oracle.generate(pop, L, 5, 1, 0.5, 0.5, [4, 17], depth2leaf(max(2, mlc - 2)), rou, CS3)
    mi = (rng.integers(0, 1024, pop) % sizes).astype(np.int32)
    mi[:6] = [-1, 100000, 0, -5, L, L - 1]
    assert_forest_equal(g.mutate(*forest, mi, *new), oracle.mutate(*forest, mi, *new), "mutate")
    # second generation: operate on evolved trees (longer, results hit the length cap)
    child = oracle.crossover(*forest, *idx)
    csz = child[2][:, 0].astype(np.int64)
    idx2 = random_crossover_indices(rng, csz, 5000)
    got, want = g.crossover(*child, *idx2), oracle.crossover(*child, *idx2)
    assert_forest_equal(got, want, "crossover gen 2")
    for n in range(0, 5000, 97):
        assert oracle.validate_tree(got[1][n], got[2][n]) == 0


def test_crossover_every_position_of_one_pair(g, oracle):
    """Exhaustive: every (left position, right position) pair of two 15-node trees, including a
    ternary parent whose MIDDLE child is replaced (the reference reads an uncopied size there,
    mutation.cu:68-75 — SURVEY.md §7.3-5a)."""
    rou = roulette_uniform([0, 1, 14])
    forest = oracle.generate(64, 40, 3, 1, 0.5, 0.5, [3, 1], depth2leaf(4, 0.0), rou, CS3)
    sizes = forest[2][:, 0].astype(np.int64)
    a, b = int(np.argmax(sizes)), int(np.argsort(sizes)[-2])
    pairs = [(p, q) for p in range(sizes[a]) for q in range(sizes[b])]
    li = np.full(len(pairs), a, np.int32); ri = np.full(len(pairs), b, np.int32)
    ln = np.array([p for p, _ in pairs], np.int32); rn = np.array([q for _, q in pairs], np.int32)
    assert_forest_equal(g.crossover(*forest, li, ri, ln, rn), oracle.crossover(*forest, li, ri, ln, rn), "exhaustive")


# ---- evaluation ---------------------------------------------------------------------------------
@pytest.mark.parametrize("out_len", [1, 3])
def test_evaluate_arith_bit_exact(g, oracle, rng, out_len):
    forest = oracle.generate(5000, 64, 17, out_len, 0.5, 0.5, [1, 2], depth2leaf(6), roulette_uniform(ARITH), CS3)
    X = rng.normal(0, 1, (5000, 17)).astype(np.float32)
    assert np.array_equal(fbits(g.evaluate(*forest, X, out_len)), fbits(oracle.evaluate(*forest, X, out_len)))


@pytest.mark.parametrize("pop,L,out_len,var_len,mlc", [(5003, 128, 6, 17, 7), (777, 64, 64, 64, 6), (20001, 64, 3, 5, 6), (9, 16, 2, 3, 3), (1001, 256, 10, 30, 5), (2001, 256, 6, 17, 8)])
def test_evaluate_multi_output_trees_share_the_passes_of_a_wave(g, oracle, rng, pop, L, out_len, var_len, mlc):
    """tree_evaluate on multi-output trees: one wave per tree, every OUT node at once (evaluate_prepared.hip eval_direct_kernel) --
    populations that are no multiple of the workgroup, as many outputs as lanes, trees of more than 64 nodes (the same reading chunk
    by chunk, round 5), rows without a tree, a truncated tree and a subtree size that does not add up (the stack interpreter)."""
    forest = oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [pop % 97, L], depth2leaf(mlc), roulette_uniform(ARITH), CS3)
    v, t, s = (a.copy() for a in forest)
    bad = []
    if pop > 100:
        s[5, 0] = 0; s[pop - 1, 0] = -2; bad += [5, pop - 1]                       # no tree: NaN row
        t[6, :3] = [3, 0, 0]; s[6, :3] = [3, 1, 1]; v[6, 0] = 1; s[6, 0] = 2; bad.append(6)   # truncated: stack underflow -> NaN row
        for r in np.flatnonzero(s[:, 0] > 7)[:3]:                                 # sizes that do not describe the tree: the stack interpreter
            if r not in bad:
                s[r, 1] += 1
    X = rng.normal(0, 1, (pop, var_len)).astype(np.float32)
    got = g.evaluate(v, t, s, X, out_len)
    want = oracle.evaluate(v, t, s, X, out_len)
    ok = np.ones(pop, bool); ok[bad] = False   # (malformed trees: NaN here, undefined in the reference)
    assert np.array_equal(fbits(got[ok]), fbits(want[ok]))
    if bad:
        assert np.isnan(got[bad]).all()
    if L > 64 and mlc > 6:
        assert (s[:, 0] > 64).sum() > 10, "no tree beyond 64 nodes: the case is not covered"


def test_evaluate_each_function_alone(g, oracle, rng):
    """One function at a time (plus + to build trees): tolerances per family."""
    for f in range(29):
        funcs = [f] if f == 0 or f >= 14 else [f]
        forest = oracle.generate(400, 40, 3, 1, 0.5, 0.5, [f, 99], depth2leaf(3, 0.1), roulette_uniform(funcs + [1]), [-1.5, 0.0, 0.5, 2.0])
        X = rng.uniform(-2, 2, (400, 3)).astype(np.float32)
        got, want = g.evaluate(*forest, X, 1), oracle.evaluate(*forest, X, 1)
        exact = f in (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 23, 24, 25, 26, 27, 28)  # IEEE-exact ops (incl. / and sqrt)
        if exact:
            assert np.array_equal(fbits(got), fbits(want)), f"function {f} must be bit-exact"
        else:
            want, tol, unst = sensitivity(oracle, lambda: oracle.evaluate(*forest, X, 1))
            assert_within_sensitivity(got, want, tol, unst, f"function {f}", min_tight=0.2)


def test_deep_and_long_trees_take_the_general_path(g, oracle, rng):
    """Left-deep chains need an operand stack of (L+1)/2 > 32 entries: register path -> scratch path."""
    L = 1024
    for n_nodes in (31, 63, 65, 129, 1023):
        k = (n_nodes - 1) // 2
        # prefix of a left-deep tree: k ADD/SUB nodes, then k+1 leaves
        v = np.zeros((3, L), np.float32); t = np.zeros((3, L), np.int16); s = np.zeros((3, L), np.int16)
        for r in range(3):
            v[r, :k] = 1 + (np.arange(k) + r) % 3; t[r, :k] = 3
            s[r, :k] = n_nodes - 2 * np.arange(k)
            leaves = np.arange(k, n_nodes)
            t[r, leaves] = np.where(leaves % 2 == 0, 0, 1); v[r, leaves] = np.where(leaves % 2 == 0, leaves % 4, 0.5 + r)
            s[r, leaves] = 1
            assert oracle.validate_tree(t[r], s[r]) == 0
        X = rng.uniform(-1, 1, (3, 4)).astype(np.float32)
        assert np.array_equal(fbits(g.evaluate(v, t, s, X, 1)), fbits(oracle.evaluate(v, t, s, X, 1))), n_nodes
        Xd = rng.uniform(-1, 1, (130, 4)).astype(np.float32); yd = rng.uniform(-1, 1, (130, 1)).astype(np.float32)
        assert_close_classes(g.sr_fitness(v, t, s, Xd, yd), oracle.sr_fitness(v, t, s, Xd, yd), RTOL_ARITH, what=f"deep {n_nodes}")
        assert np.array_equal(fbits(g.batch_evaluate(v, t, s, Xd, 1)), fbits(oracle.batch_evaluate(v, t, s, Xd, 1)))


def test_malformed_trees_yield_nan(g):
    L = 16
    v = np.zeros((4, L), np.float32); t = np.zeros((4, L), np.int16); s = np.zeros((4, L), np.int16)
    t[0, :2] = [3, 0]; s[0, :2] = [2, 1]; v[0, 0] = 1          # ADD with one operand: stack underflow
    t[1, :2] = [0, 0]; s[1, 0] = 2                             # two leaves: final height 2
    s[2, 0] = 0                                                # empty tree
    t[3, :3] = [3, 0, 1]; s[3, :3] = [3, 1, 1]; v[3, :3] = [1, 0, 2]  # valid: x0 + 2
    X = np.ones((5, 1), np.float32); y = np.zeros((5, 1), np.float32)
    fit = g.sr_fitness(v, t, s, X, y)
    assert np.isnan(fit[:3]).all() and fit[3] == 9.0
    ev = g.evaluate(v, t, s, np.ones((4, 1), np.float32), 1)
    assert np.isnan(ev[:3]).all() and ev[3, 0] == 3.0


def test_malformed_trees_in_a_classifier_count_as_class_zero(g):
    """a malformed tree has NaN outputs in every row; the arg-max of a NaN row is class 0 (torch.argmax(clip(softmax(NaN)))), so
    its count is the number of rows labelled 0 -- on the compiled-program path and on the tile-group kernel alike"""
    L = 16
    v = np.zeros((4, L), np.float32); t = np.zeros((4, L), np.int16); s = np.zeros((4, L), np.int16)
    t[0, :2] = [3, 0]; s[0, :2] = [2, 1]; v[0, 0] = 1          # ADD with one operand
    t[1, :2] = [0, 0]; s[1, 0] = 2                             # two leaves
    s[2, 0] = 0                                                # empty tree
    # valid, two outputs: OUT0 += x0 + 2 (root: ADD flagged OUT, output 0)
    t[3, :3] = [3 | 0x80, 0, 1]; s[3, :3] = [3, 1, 1]
    v[3, 0] = np.array([1 | (0 << 16)], np.uint32).view(np.float32)[0]; v[3, 1:3] = [0, 2]
    X = np.array([[1.0], [-5.0], [0.5], [-2.5], [3.0]], np.float32)
    labels = np.array([0, 1, 0, 0, 1], np.int32)
    cnt = g.batch_argmax_count(v, t, s, X, labels, 2)
    # tree 3: outputs (x0 + 2, 0): class 0 where x0 + 2 >= 0 (ties go to the first class), class 1 elsewhere
    want3 = int((((X[:, 0] + 2) >= 0).astype(np.int32) == (labels == 0)).sum())
    assert cnt[:3].tolist() == [3, 3, 3] and int(cnt[3]) == want3, (cnt, want3)


# ---- SR fitness ----------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 8, 63, 64, 65, 256, 1000, 1024, 1025, 2500, 5000])
def test_sr_fitness_ragged_datapoint_counts(g, oracle, rng, D):
    forest = oracle.generate(777, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), CS3)
    X = rng.uniform(-5, 5, (D, 10)).astype(np.float32); y = rng.uniform(-5, 5, (D, 1)).astype(np.float32)
    for mse in (True, False):
        assert_close_classes(g.sr_fitness(*forest, X, y, mse), oracle.sr_fitness(*forest, X, y, mse), RTOL_ARITH, what=f"D={D} mse={mse}")


@pytest.mark.parametrize("D", [8, 100, 600])   # the 1-, 4- and 8-row interpreter builds
def test_sr_fitness_every_population_size_is_fully_evaluated(g, oracle, rng, D):
    """The interpreter hands most of the population out dynamically, one region and work counter per XCD: a region whose XCD
    runs no workgroup (few workgroups: small populations) would never be worked off.  The fitness buffer is poisoned before every call, so a tree nobody evaluated shows."""
    X = rng.uniform(-3, 3, (D, 4)).astype(np.float32); y = rng.uniform(-3, 3, (D, 1)).astype(np.float32)
    big = oracle.generate(70001, 64, 4, 1, 0.5, 0.5, [9, D], depth2leaf(6), roulette_uniform(ARITH), CS3)
    want = oracle.sr_fitness(*big, X, y)
    for pop in (1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 500, 1023, 1025, 4097, 9999, 20000, 33333, 70001):
        got = g.sr_fitness(*(a[:pop] for a in big), X, y)   # (the helper fills the fitness buffer with 12345 before the call)
        assert not (got == 12345.0).any(), f"pop {pop}: {int((got == 12345.0).sum())} trees were not evaluated"
        assert_close_classes(got, want[:pop], RTOL_ARITH, what=f"pop={pop}")


@pytest.mark.parametrize("var_len,out_len", [(1, 1), (16, 1), (17, 1), (32, 2), (33, 1), (64, 10), (5, 16), (5, 17)])
def test_sr_fitness_shapes(g, oracle, rng, var_len, out_len):
    forest = oracle.generate(500, 64, var_len, out_len, 0.5, 0.5, [5, 5], depth2leaf(6), roulette_uniform(ARITH), CS3)
    X = rng.uniform(-2, 2, (300, var_len)).astype(np.float32); y = rng.uniform(-2, 2, (300, out_len)).astype(np.float32)
    assert_close_classes(g.sr_fitness(*forest, X, y), oracle.sr_fitness(*forest, X, y), RTOL_ARITH, what=f"var={var_len} out={out_len}")
    be = g.batch_evaluate(*forest, X[:70], out_len)
    assert np.array_equal(fbits(be), fbits(oracle.batch_evaluate(*forest, X[:70], out_len)))


def test_sr_fitness_c1_xor_config(g, oracle):
    """BASELINE configs[0]: XOR-3d, pop 5000, max_tree_len 32, 8 datapoints."""
    forest = oracle.generate(5000, 32, 3, 1, 0.5, 0.5, [42, 0], depth2leaf(4), roulette_uniform(ARITH), CS3)
    X = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.float32)
    y = (X.sum(1) % 2).astype(np.float32)[:, None]
    got = g.sr_fitness(*forest, X, y)
    assert_close_classes(got, oracle.sr_fitness(*forest, X, y), RTOL_ARITH, what="C1")
    assert int(np.isnan(got).sum()) == 2430


def test_sr_fitness_c2_config_subset_and_kernel_types(g, oracle):
    """BASELINE configs[1] shape (10 variables, 1024 datapoints, L = 64) on a 20k-tree subset, and every
    kernel_type code returns the same bits."""
    forest = oracle.generate(20000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), CS3)
    X, y = c2_dataset()
    want = oracle.sr_fitness(*forest, X, y)
    got = g.sr_fitness(*forest, X, y, True, 0)
    assert_close_classes(got, want, RTOL_ARITH, what="C2")
    for kt in (1, 2, 3, 4):
        assert np.array_equal(bits(g.sr_fitness(*forest, X, y, True, kt)), bits(got))
    assert np.array_equal(bits(g.sr_fitness(*forest, X, y, True, 0)), bits(got)), "run-to-run reproducible"


@pytest.mark.parametrize("funcs,name", [(PAPER7, "paper7"), ([1, 2, 3, 4, 20, 22, 6], "exp-log-pow"), ([1, 2, 20, 27, 6, 4, 23], "vis.ipynb")])
def test_sr_fitness_library_function_setsper_tree_tolerance(g, oracle, funcs, name):
    """trees of sin cos tan / exp log pow against the oracle (host libm): EVERY tree within 1e-5 relative plus twice the
    spread a 3-ulp perturbation of its library calls produces in the oracle itself — no allowed-bad fraction.  Trees whose NaN /
    inf class flips under that perturbation are compared by class membership only."""
    forest = oracle.generate(4000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform(funcs), CS3)
    X, y = c2_dataset()
    got = g.sr_fitness(*forest, X, y).astype(np.float64)
    want, tol, unstable = per_tree_tolerance(oracle, forest, X, y)
    assert_within_sensitivity(got, want, tol, unstable, name)


def test_sr_fitness_full_c2_forest_against_the_oracle(g, oracle):
    """BASELINE configs[1] in full: all 100 000 trees x 1024 datapoints against the oracle at the north-star tolerance"""
    forest = oracle.generate(100_000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), CS3)
    X, y = c2_dataset()
    got = g.sr_fitness(*forest, X, y)
    assert_close_classes(got, oracle.sr_fitness(*forest, X, y), RTOL_ARITH, what="C2 full")
    # the call a Forest that knows its function set makes (+ - * /: the arithmetic-only compiler, ONE record array, no general compiler)
    assert np.array_equal(bits(g.sr_fitness(*forest, X, y, func_mask=ARITH_MASK)), bits(got)), "C2 full: masked call differs from the unmasked one"


def test_sr_fitness_evolved_forest_against_the_oracle(g, oracle, rng):
    """a population after 30 generations of the default GP loop (trees grow towards the length cap, deeper operand stacks,
    constants folded through several levels): fitness against the oracle at 1e-5"""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
    forest = Forest.random_generate(20_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
    X, y = c2_dataset()
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.01))
    for _ in range(30):
        f = -algo.forest.SR_fitness(Xd, yd)
        algo.step(torch.where(torch.isnan(f), torch.full_like(f, float("-inf")), f))
    f = algo.forest
    trees = (f.batch_node_value.cpu().numpy(), f.batch_node_type.cpu().numpy(), f.batch_subtree_size.cpu().numpy())
    assert trees[2][:, 0].mean() > 35, "the population did not grow: not the evolved-forest case"
    assert_close_classes(g.sr_fitness(*trees, X, y), oracle.sr_fitness(*trees, X, y), RTOL_ARITH, what="generation 30")


def test_sr_fitness_trees_too_deep_for_every_register_stack(g, oracle, rng):
    """Left-leaning trees of 18-20 levels in rows of 64 nodes, every right operand a unary function of a leaf: the operand stack of the
    interpreter's evaluation order reaches 19-21 entries -- beyond the threaded code's 9 (no reordering in rows of 64) AND beyond the FULL
    register build's 16, so the scratch-stack evaluation takes them: since round 5 inside the FULL build's launch (csrc/sr_fitness.hip
    general_tree_fitness; the scratch-stack kernel is no launch of its own behind the threaded code any more).  Mixed with ordinary trees,
    against the oracle, whole and in halves."""
    pop, L = 3000, 64
    v, t, s = (a.copy() for a in oracle.generate(pop, L, 5, 1, 0.0, 0.5, [8, 8], depth2leaf(6), roulette_uniform(ARITH), CS3))
    deep = 0
    for r in range(0, pop, 7):
        levels = int(rng.integers(18, 21))
        nodes = []
        for k in range(levels, 0, -1):          # prefix order: level k = f(level k - 1, neg(leaf)), size 3 k + 1
            nodes.append((3, float(rng.integers(1, 5)), 3 * k + 1))
        nodes.append((0, float(rng.integers(0, 5)), 1))   # level 0: a variable
        for _ in range(levels):
            nodes.append((2, 25.0, 2)); nodes.append((0, float(rng.integers(0, 5)), 1)) if rng.random() < 0.7 else nodes.append((1, float(rng.choice(CS3)), 1))
        # the right operands follow their level's left subtree: rebuild in true prefix order
        def build(k):
            if k == 0:
                return [(0, float(rng.integers(0, 5)), 1)]
            left = build(k - 1)
            right = [(2, 25.0, 2), (0, float(rng.integers(0, 5)), 1)]
            return [(3, float(rng.integers(1, 5)), 1 + len(left) + 2)] + left + right
        nodes = build(levels)
        assert len(nodes) <= L
        v[r] = 0; t[r] = 0; s[r] = 0
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
        deep += 1
    assert deep > 400
    X = rng.uniform(-2, 2, (700, 5)).astype(np.float32); y = rng.uniform(-2, 2, (700, 1)).astype(np.float32)
    want = oracle.sr_fitness(v, t, s, X, y)
    got = g.sr_fitness(v, t, s, X, y)
    assert_close_classes(got, want, RTOL_ARITH, what="trees of 19-21 stack entries")
    h = pop // 2
    halves = np.concatenate([g.sr_fitness(v[:h], t[:h], s[:h], X, y), g.sr_fitness(v[h:], t[h:], s[h:], X, y)])
    assert np.array_equal(bits(got), bits(halves))
    assert np.array_equal(bits(g.sr_fitness(v, t, s, X, y, func_mask=ARITH_MASK | (1 << 25))), bits(got)), "the masked call (no general compiler launch)"


def test_full_size_properties(g, oracle):
    """BASELINE full size (pop 100k x 1024 datapoints): size-independent checks —
    (1) fitness of a forest equals fitness of its two halves concatenated (row independence),
    (2) evaluate-then-reduce equals the fused fitness on a sampled subset,
    (3) linearity: fitness against labels y equals mean((tree - y)^2) from batch_evaluate."""
    rou, d2l = roulette_uniform(ARITH), depth2leaf(6)
    pop = 100_000
    forest = g.generate(pop, 64, 10, 1, 0.5, 0.5, [42, 0], d2l, rou, CS3)
    assert abs(forest[2][:, 0].mean() - 26.26) < 0.2  # measured mean length 26.26
    X, y = c2_dataset()
    full = g.sr_fitness(*forest, X, y)
    h = pop // 2
    halves = np.concatenate([g.sr_fitness(*(a[:h] for a in forest), X, y), g.sr_fitness(*(a[h:] for a in forest), X, y)])
    assert np.array_equal(bits(full), bits(halves))
    pick = np.arange(0, pop, 997)
    sub = tuple(a[pick] for a in forest)
    assert_close_classes(full[pick], oracle.sr_fitness(*sub, X, y), RTOL_ARITH, what="sample vs oracle")
    pred = g.batch_evaluate(*sub, X, 1)[:, :, 0]
    with np.errstate(all="ignore"):
        d = pred - y[:, 0][None, :]                    # fp32, like the kernel
        ref = (d * d).astype(np.float64).mean(1)       # squares in fp32 (same overflow), sum in fp64
    assert_close_classes(full[pick], ref.astype(np.float32), 1e-4, what="fused vs batch_evaluate")


def test_north_star_population_against_the_oracle_and_its_eight_shards(g, oracle):
    """BASELINE north_star in full: 1 M trees x 1024 datapoints (the forest and dataset bench.py times).
    (1) every fitness word against the oracle at 1e-5 with identical NaN / inf classes;
    (2) the population cut into the 8 contiguous shards `bench.py --gpus 8` gives its ranks returns the same bits:
        the interpreter's work distribution (static share, per-XCD dynamic batches, batch sizes that depend on the
        population size) must not show in the results;
    (3) the call bench.py TIMES -- evogp_hip_sr_fitness_hinted with the function mask of + - * / (Forest.SR_fitness on a forest that
        knows its descriptor): tc_compile_packed_kernel<32, false, false>, one record array, no general compiler -- returns the same
        bits as the unmasked call, whole and in the eight shards, and so meets the oracle too (VERDICT r04 weak #1a)."""
    rou, d2l = roulette_uniform(ARITH), depth2leaf(6)
    pop = 1_000_000
    forest = g.generate(pop, 64, 10, 1, 0.5, 0.5, [42, 0], d2l, rou, CS3)
    X, y = c2_dataset()
    full = g.sr_fitness(*forest, X, y)
    assert not (full == 12345.0).any()
    shards = np.concatenate([g.sr_fitness(*(a[r * pop // 8:(r + 1) * pop // 8] for a in forest), X, y) for r in range(8)])
    assert np.array_equal(bits(full), bits(shards))
    masked = g.sr_fitness(*forest, X, y, func_mask=ARITH_MASK)
    assert np.array_equal(bits(masked), bits(full)), "the masked (timed) call differs from the unmasked one"
    mshards = np.concatenate([g.sr_fitness(*(a[r * pop // 8:(r + 1) * pop // 8] for a in forest), X, y, func_mask=ARITH_MASK) for r in range(8)])
    assert np.array_equal(bits(mshards), bits(full)), "the masked call's shards differ"
    want = oracle.sr_fitness(*forest, X, y)
    assert_close_classes(full, want, RTOL_ARITH, what="north star, 1 M trees")
    assert_close_classes(masked, want, RTOL_ARITH, what="north star, 1 M trees, the masked call bench.py times")


@pytest.mark.parametrize("var_len,out_len,D,L,pop", [(64, 10, 1797, 128, 600), (40, 1, 300, 64, 800), (8, 3, 2500, 64, 700),
                                                     (5, 1, 1100, 64, 900), (100, 16, 130, 32, 300)])
def test_batch_evaluate_wide_inputs_and_long_datasets(g, oracle, rng, var_len, out_len, D, L, pop):
    """The tile-group kernel (more than 32 variables, or more rows than one workgroup keeps resident): arithmetic trees
    bit-exact per datapoint against the oracle (classifier shape of configs[3] first)."""
    f = oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [21, 4], depth2leaf(5), roulette_uniform(ARITH), [-1, 0, 1, 0.5])
    X = rng.uniform(-3, 3, (D, var_len)).astype(np.float32)
    got = g.batch_evaluate(*f, X, out_len)
    want = oracle.batch_evaluate(*f, X, out_len)
    assert np.array_equal(fbits(got), fbits(want))


def test_batch_argmax_count_matches_the_torch_rule(g, oracle, rng):
    """Fused classification epilogue: count of rows whose arg-max output equals the label, with the arg-max that
    torch.argmax(clip(softmax(x))) returns (NaN / infinite maximum -> index 0, first maximum otherwise)."""
    import torch

    pop, L, var_len, out_len, D = 1500, 64, 64, 10, 1797
    f = oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [3, 1], depth2leaf(5), roulette_uniform(ARITH), [-1, 0, 1, 0.5])
    X = rng.uniform(0, 16, (D, var_len)).astype(np.float32)
    labels = rng.integers(0, out_len, D).astype(np.int32)
    got = g.batch_argmax_count(*f, X, labels, out_len)
    want = torch_rule_counts(oracle.batch_evaluate(*f, X, out_len), labels)               # torch's kernels on the device
    # integer output: equality.  Rows where two soft-max probabilities that differ before rounding come out equal (torch then
    # takes the earlier class) are detected in the kernel and their trees recounted with torch's arithmetic (interp.hpp)
    assert np.array_equal(got, want), (np.abs(got - want).max(), (got != want).mean(), np.flatnonzero(got != want)[:5])


def _near_tie_forest(rng, pop, out_len, L=64):
    """multi-output trees whose outputs are x0 + c_o with constants a few ulps apart: ADD(OUT_0: x0 + c_0, ADD(OUT_1: x0 + c_1, ...))"""
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)
    cs = np.array([0.0, 0.0, 3e-8, -3e-8, 6e-8, -6e-8, 1.2e-7, -1.2e-7, 1.8e-7, 2.4e-7, -2.4e-7, 5e-7, 1e-3, -1e-3], np.float32)
    n = 4 * out_len - 1                                  # out_len OUT nodes of 3 nodes each + (out_len - 1) plain ADD nodes
    for r in range(pop):
        pos = 0
        for o in range(out_len):
            if o < out_len - 1:                          # plain ADD: hands its last operand up, adds to no output
                v[r, pos] = 1.0; t[r, pos] = 3; s[r, pos] = n - pos; pos += 1
            v[r, pos:pos + 1].view(np.uint32)[:] = (o << 16) | 1                  # OUT node: {function ADD, output index o}
            t[r, pos] = 3 | 0x80; s[r, pos] = 3
            v[r, pos + 1] = 0.0; t[r, pos + 1] = 0; s[r, pos + 1] = 1             # x0
            v[r, pos + 2] = cs[rng.integers(0, len(cs))]; t[r, pos + 2] = 1; s[r, pos + 2] = 1
            pos += 3
        assert pos == n
    return v, t, s


@pytest.mark.parametrize("out_len,D,var_len", [(2, 300, 3), (3, 64, 3), (5, 1000, 3), (10, 1797, 64), (16, 700, 3)])
def test_batch_argmax_count_on_near_ties_equals_torch(g, oracle, rng, out_len, D, var_len):
    """The adversarial case for the fused classification count: outputs a few ulps apart, where torch's fp32 soft-max rounds two
    different outputs to the SAME probability and torch.argmax returns the earlier class although a later output is larger
    (classification.py:62-67).  Rows like that are detected in END_CLS (threaded code) and in the tile-group kernel, their trees
    recounted with aten's arithmetic (softmax_warp_forward: expf, xor-butterfly sum, IEEE division) — counts equal torch's."""
    pop = 1500
    f = _near_tie_forest(rng, pop, out_len)
    for r in range(0, pop, 97):
        assert oracle.validate_tree(f[1][r], f[2][r]) == 0
    X = rng.uniform(0.1, 1.9, (D, var_len)).astype(np.float32)
    X[::7, 0] = rng.uniform(0.01, 0.03, len(X[::7]))           # finer ulps: differences of 3e-8 survive the addition
    labels = rng.integers(0, out_len, D).astype(np.int32)
    outs = oracle.batch_evaluate(*f, X, out_len)
    want = torch_rule_counts(outs, labels)
    raw = (np.argmax(outs, axis=2) == labels[None, :]).sum(1)
    assert (raw != want).mean() > 0.01, "the forest does not exercise the near-tie rule"
    got = g.batch_argmax_count(*f, X, labels, out_len)
    assert np.array_equal(got, want), (np.abs(got - want).max(), (got != want).mean(), np.flatnonzero(got != want)[:5])


def test_sr_fitness_repeated_calls_streams_and_graph_replay(g, oracle, rng):
    """The call chain's scratch block alternates per stream and is zeroed by the previous call's first kernel; a call
    recorded into a HIP graph carries its own memset.  Repeated calls, calls on a side stream, calls that take the
    fallback kernels in between, and graph replays must all give the same fitness."""
    import torch

    rou, d2l = roulette_uniform(ARITH), depth2leaf(6)
    v, t, s = g.generate(3000, 64, 4, 1, 0.5, 0.3, [5, 6], d2l, rou, CS3)
    X = rng.standard_normal((1024, 4)).astype(np.float32)
    y = rng.standard_normal((1024, 1)).astype(np.float32)
    want = oracle.sr_fitness(v, t, s, X, y)
    a = [g.dev(v, np.float32), g.dev(t, np.int16), g.dev(s, np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]

    def call(out, stream):
        rc = g.L.evogp_hip_sr_fitness(3000, 1024, 64, 4, 1, 1, *[x.data_ptr() for x in a], out.data_ptr(), 0, stream.cuda_stream)
        assert rc == 0, g.L.evogp_hip_error_string(rc)

    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i in range(7):
        out = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
        if i % 3 == 2:
            side.wait_stream(main)   # (the fill above is main's work: without this the side stream's result can be overwritten by it)
        call(out, side if i % 3 == 2 else main)
        if i == 3:  # a call of another shape (fallback kernels, different scratch use) in between
            g.sr_fitness(v[:50], t[:50], s[:50], X[:100], np.tile(y[:100], (1, 1)), use_mse=False, kernel_type=2)
        outs.append(out)
    torch.cuda.synchronize()
    for i, out in enumerate(outs):
        assert_close_classes(out.cpu().numpy(), want, RTOL_ARITH, 0.0, f"call {i}")

    gout = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    cap = torch.cuda.Stream()
    cap.wait_stream(main)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        call(gout, cap)  # warm-up outside the capture
        cap.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            call(gout, torch.cuda.current_stream())
    for i in range(3):
        gout.fill_(555.0)
        graph.replay()
        torch.cuda.synchronize()
        assert_close_classes(gout.cpu().numpy(), want, RTOL_ARITH, 0.0, f"replay {i}")
    # eager calls after the capture still work
    out = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    call(out, cap); call(out, main)
    torch.cuda.synchronize()
    assert_close_classes(out.cpu().numpy(), want, RTOL_ARITH, 0.0, "after capture")
    # a capture per generation with eager calls in between (a new forest every generation means a new capture): the engine holds one
    # record buffer for eager calls and one for the graphs, however often the two alternate (round 4 leaked one per alternation)
    import evogp_amd
    held = evogp_amd.program_buffer_bytes()
    graphs = []
    for i in range(5):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(gr, stream=cap):
                call(gout, torch.cuda.current_stream())
        graphs.append(gr)
        call(out, main)
        gout.fill_(555.0)
        gr.replay()
        torch.cuda.synchronize()
        assert_close_classes(gout.cpu().numpy(), want, RTOL_ARITH, 0.0, f"capture {i} of the alternation")
        assert_close_classes(out.cpu().numpy(), want, RTOL_ARITH, 0.0, f"eager call {i} of the alternation")
    assert evogp_amd.program_buffer_bytes() == held, "record memory grew while captures and eager calls alternated"


def test_graph_replay_survives_a_later_call_that_grows_the_record_buffer(g, oracle, rng):
    """A captured tree_SR_fitness keeps pointing at the engine's record memory.  A later eager call with a population a hundred
    times as large must not free or overwrite what the graph reads (sr_tc.hip: the buffer a capture used is left to the graphs, the
    eager call takes a fresh one; a call that cannot get records inside a capture runs on the ring-based kernel): replays before
    and after the large call, and interleaved with eager calls of the small population, give the oracle's values."""
    import torch

    import evogp_amd

    rou, d2l = roulette_uniform(ARITH), depth2leaf(6)
    v, t, s = g.generate(3000, 64, 4, 1, 0.5, 0.3, [15, 16], d2l, rou, CS3)
    V, T, S = g.generate(300_000, 64, 4, 1, 0.5, 0.3, [17, 18], d2l, rou, CS3)
    X = rng.standard_normal((1024, 4)).astype(np.float32)
    y = rng.standard_normal((1024, 1)).astype(np.float32)
    want = oracle.sr_fitness(v, t, s, X, y)
    sample = rng.choice(300_000, 2000, replace=False)
    want_big = oracle.sr_fitness(V[sample], T[sample], S[sample], X, y)
    small = [g.dev(v, np.float32), g.dev(t, np.int16), g.dev(s, np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]
    big = [g.dev(V, np.float32), g.dev(T, np.int16), g.dev(S, np.int16), small[3], small[4]]

    def call(args, pop, out, stream):
        rc = g.L.evogp_hip_sr_fitness(pop, 1024, 64, 4, 1, 1, *[x.data_ptr() for x in args], out.data_ptr(), 0, stream.cuda_stream)
        assert rc == 0, g.L.evogp_hip_error_string(rc)

    evogp_amd.release_workspaces()
    main = torch.cuda.current_stream()
    gout = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    cap = torch.cuda.Stream()
    cap.wait_stream(main)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        call(small, 3000, gout, cap)           # warm-up outside the capture (a stream's first call allocates its scratch block)
        cap.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            call(small, 3000, gout, torch.cuda.current_stream())
    gout.fill_(555.0)
    graph.replay()
    torch.cuda.synchronize()
    assert_close_classes(gout.cpu().numpy(), want, RTOL_ARITH, 0.0, "replay before the large call")
    bout = torch.full((300_000,), 777.0, dtype=torch.float32, device=g.DEV)
    call(big, 300_000, bout, main)           # grows the record memory a hundred times
    gout.fill_(555.0)
    graph.replay()                          # ... while this one still reads the old
    eout = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    call(small, 3000, eout, main)
    graph.replay()
    torch.cuda.synchronize()
    assert_close_classes(gout.cpu().numpy(), want, RTOL_ARITH, 0.0, "replay after the large call")
    assert_close_classes(eout.cpu().numpy(), want, RTOL_ARITH, 0.0, "eager call between replays")
    assert_close_classes(bout.cpu().numpy()[sample], want_big, RTOL_ARITH, 0.0, "the large call")
    del graph
    # a capture that needs MORE record memory than exists cannot allocate: it runs on the ring-based kernel (records in L2), whose
    # ring the eager warm-up left behind
    evogp_amd.release_workspaces()
    main = torch.cuda.current_stream()
    eout = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    call(small, 3000, eout, main)
    torch.cuda.synchronize()
    held = evogp_amd.program_buffer_bytes()
    bout = torch.full((300_000,), 777.0, dtype=torch.float32, device=g.DEV)
    cap = torch.cuda.Stream()
    cap.wait_stream(main)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        call(small, 3000, eout, cap)           # (the stream's scratch block)
        cap.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            call(big, 300_000, bout, torch.cuda.current_stream())
    assert evogp_amd.program_buffer_bytes() == held, "the capture allocated record memory"
    for i in range(2):
        bout.fill_(555.0)
        graph.replay()
        call(small, 3000, eout, main)
        torch.cuda.synchronize()
        assert_close_classes(bout.cpu().numpy()[sample], want_big, RTOL_ARITH, 0.0, f"large call captured, replay {i}")
        assert_close_classes(eout.cpu().numpy(), want, RTOL_ARITH, 0.0, "eager call next to it")
    del graph
    evogp_amd.release_workspaces()


@pytest.mark.parametrize("D", [1, 300])
def test_sr_fitness_division_modes_on_special_operands(g, oracle, D):
    """Every ordered pair of special operands through `x_i / x_j` in the threaded-code path, in the three division modes
    (include/evogp_hip.h: evogp_hip_set_sr_division).  IEEE and SHORT must reproduce the oracle's |quotient| — classes
    exactly, values to the last bit for D = 1 — over the whole range; FAST on the pairs inside its documented range."""
    vals = np.array([0.0, -0.0, 1.0, -1.0, 3.0, -7.5, 0.1, np.inf, -np.inf, np.nan, 1e-45, -3e-39, 1.1754944e-38, 2e-38,
                     1e-30, 4e-20, 1e20, -6e29, 8.5e37, 1.7e38, -3.4e38, 3.4028235e38, 2.0**-126, 2.0**126, 2.0**127,
                     5e-324, 1.0000001, 0.99999994, 16777216.0, 1.0 / 3.0], dtype=np.float32)
    n = len(vals)
    assert n <= 32
    pop, L = n * n, 8
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)
    ii, jj = np.divmod(np.arange(pop), n)
    v[:, 0] = 4.0; t[:, 0] = 3; s[:, 0] = 3          # DIV, binary function
    v[:, 1] = ii; t[:, 1] = 0; s[:, 1] = 1           # left operand  x_i
    v[:, 2] = jj; t[:, 2] = 0; s[:, 2] = 1           # right operand x_j
    X = np.tile(vals[None, :], (D, 1)).astype(np.float32)
    y = np.zeros((D, 1), np.float32)
    want = oracle.sr_fitness(v, t, s, X, y, use_mse=False)
    with np.errstate(all="ignore"):
        q = np.where(vals[jj] == 0, np.float32(np.nan), vals[ii] / vals[jj]).astype(np.float32)
    in_range = np.isfinite(q) & (np.abs(vals[jj]) <= 2.0**126) & (np.abs(vals[jj]) >= 2.0**-126) & \
        ((np.abs(q) >= 2.0**-126) | (q == 0)) & (np.abs(vals[ii]) >= 2.0**-126)
    try:
        for mode, code in (("ieee", 0), ("short", 2), ("fast", 1)):
            assert g.L.evogp_hip_set_sr_division(code) == 0 and g.L.evogp_hip_get_sr_division() == code
            got = g.sr_fitness(v, t, s, X, y, use_mse=False)
            sel = np.ones(pop, bool) if mode != "fast" else in_range
            assert_close_classes(got[sel], want[sel], 0.0 if D == 1 else 1e-6, 0.0, f"{mode} division, D={D}")
    finally:
        assert g.L.evogp_hip_set_sr_division(2) == 0
    assert g.L.evogp_hip_set_sr_division(7) < 0


def test_wide_kernel_deep_multi_output_trees_are_redone(g, oracle, rng):
    """Trees whose operand stack exceeds the register stack of the tile-group kernel (16 entries for multi-output): the
    store mode leaves a sentinel for the scratch-stack kernel, the fused accuracy flags the count word and recounts in
    its follow-up kernel.  Left-deep chains mixed into a generated multi-output forest."""
    import torch

    pop, L, var_len, out_len, D = 400, 128, 40, 4, 700
    v, t, s = (a.copy() for a in oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [8, 9], depth2leaf(5), roulette_uniform(ARITH), [-1, 0, 1, 0.5]))
    for r, n_nodes in ((3, 41), (57, 127), (399, 75)):   # operand stack of (n + 1) / 2 = 21, 64, 38 entries
        k = (n_nodes - 1) // 2
        v[r] = 0; t[r] = 0; s[r] = 0
        t[r, :k] = 3 | 0x80                                                     # output-flagged binary functions
        ids = 1 + (np.arange(k) + r) % 3                                        # + - *
        oi = np.arange(k) % out_len
        v[r, :k] = ((oi.astype(np.uint32) << 16) | ids.astype(np.uint32)).view(np.float32)
        s[r, :k] = n_nodes - 2 * np.arange(k)
        leaves = np.arange(k, n_nodes)
        t[r, leaves] = np.where(leaves % 2 == 0, 0, 1); v[r, leaves] = np.where(leaves % 2 == 0, leaves % var_len, 0.25 * (1 + r % 3))
        s[r, leaves] = 1
        assert oracle.validate_tree(t[r], s[r]) == 0
    X = rng.uniform(-2, 2, (D, var_len)).astype(np.float32)
    want = oracle.batch_evaluate(v, t, s, X, out_len)
    got = g.batch_evaluate(v, t, s, X, out_len)
    assert np.array_equal(fbits(got), fbits(want))
    labels = rng.integers(0, out_len, D).astype(np.int32)
    cnt = g.batch_argmax_count(v, t, s, X, labels, out_len)
    ref = torch_rule_counts(want, labels)
    assert np.array_equal(cnt, ref), (np.abs(cnt - ref).max(), np.flatnonzero(cnt != ref)[:5])


@pytest.mark.parametrize("D", [100, 1024, 1500])
def test_sr_fitness_unary_functions_in_the_threaded_code(g, oracle, rng, D):
    """neg / abs (and, where the build carries them, sin / cos / tan) have handlers in the threaded-code interpreter:
    operand on the stack (replaced in place) or a variable (pushed); a function of a constant is folded by the compiler
    kernel; chains of more than 31 instructions fall back to the register kernels."""
    funcs = [1, 2, 3, 4, 25, 26]
    f = oracle.generate(4000, 64, 6, 1, 0.0, 0.4, [31, 7], depth2leaf(6), roulette_uniform(funcs), [-1, 0, 1, 0.5, -2.5])
    v, t, s = (a.copy() for a in f)
    # hand-made rows: neg(abs(neg(x0))) chains of growing length (the longest exceed 31 program words), abs of a
    # constant, neg of a constant inside a division
    for r, n in enumerate((1, 2, 5, 30, 31, 32, 40, 63)):
        v[r] = 0; t[r] = 0; s[r] = 0
        v[r, :n] = np.where(np.arange(n) % 2 == 0, 25, 26); t[r, :n] = 2; s[r, :n] = n + 1 - np.arange(n)
        v[r, n] = 3; t[r, n] = 0; s[r, n] = 1                                # x3
    v[8, :3] = [26, 25, -1.5]; t[8, :3] = [2, 2, 1]; s[8, :3] = [3, 2, 1]     # abs(neg(-1.5))
    v[9, :4] = [4, 0, 25, 0.0]; t[9, :4] = [3, 0, 2, 1]; s[9, :4] = [4, 1, 2, 1]  # x0 / neg(0.0): division by -0 -> NaN
    for r in range(10):
        assert oracle.validate_tree(t[r], s[r]) == 0
    X = rng.uniform(-3, 3, (D, 6)).astype(np.float32)
    X[0, :] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1.0][:6]
    y = rng.uniform(-1, 1, (D, 1)).astype(np.float32)
    for use_mse in (True, False):
        assert_close_classes(g.sr_fitness(v, t, s, X, y, use_mse), oracle.sr_fitness(v, t, s, X, y, use_mse), RTOL_ARITH,
                             what=f"unary D={D} mse={use_mse}")


@pytest.mark.parametrize("D", [64, 1024])
def test_sr_fitness_trigonometric_handlers_match_the_register_kernels(g, oracle, rng, D):
    """sin / cos / tan in the threaded code are the device math library's small-argument path, instruction for
    instruction: the fitness must agree with the per-datapoint outputs of batch_evaluate (the C++ interpreter calling the
    library) to summation order, operands of 2^17 and more (Payne-Hanek territory) hand the tree to the register kernels
    at RUN time, and non-finite operands give NaN."""
    funcs = [1, 2, 3, 4, 14, 15, 16]
    f = oracle.generate(3000, 64, 5, 1, 0.0, 0.4, [77, 5], depth2leaf(6), roulette_uniform(funcs), [-1, 0, 1, 0.5, 3.0])
    v, t, s = (a.copy() for a in f)
    # sin / cos / tan of one variable (V form), of a product (S form), of a large constant (folded), nested
    rows = {0: ([14, 2], [2, 0]), 1: ([15, 2], [2, 0]), 2: ([16, 2], [2, 0]),
            3: ([14, 3, 0, 1], [2, 3, 0, 0]), 4: ([15, 3, 0, 1], [2, 3, 0, 0]), 5: ([16, 3, 0, 1], [2, 3, 0, 0]),
            6: ([1, 14, 1.0e6, 0], [3, 2, 1, 0]), 7: ([14, 15, 16, 4], [2, 2, 2, 0]), 8: ([16, 16, 3], [2, 2, 0])}
    for r, (vals, types) in rows.items():
        n = len(vals)
        v[r] = 0; t[r] = 0; s[r] = 0
        v[r, :n] = vals; t[r, :n] = types
        s[r, :n] = {2: [2, 1], 3: [3, 2, 1], 4: [4, 3, 2, 1]}[n] if types[1] != 3 and types[0] != 3 else 0
    s[3, :4] = [4, 3, 1, 1]; s[4, :4] = [4, 3, 1, 1]; s[5, :4] = [4, 3, 1, 1]; s[6, :4] = [4, 2, 1, 1]
    for r in rows:
        assert oracle.validate_tree(t[r], s[r]) == 0, r
    X = rng.uniform(-4, 4, (D, 5)).astype(np.float32)
    X[:, 2] = rng.uniform(-1.3e5, 1.3e5, D)        # straddles 2^17 = 131072: run-time bail-out for trees that feed it to sin
    X[:, 3] = rng.uniform(-300, 300, D)
    X[1, :] = [0.0, -0.0, 131071.9, np.inf, np.nan]
    X[2, :] = [1e-30, -1e-42, -131072.0, -np.inf, 3.0]
    y = rng.uniform(-1, 1, (D, 1)).astype(np.float32)
    got = g.sr_fitness(v, t, s, X, y, True)
    outs = g.batch_evaluate(v, t, s, X, 1)[:, :, 0]
    with np.errstate(all="ignore"):
        d = outs - y[:, 0][None, :]
        ref = (d * d).astype(np.float64).mean(1).astype(np.float32)
    assert_close_classes(got, ref, 1e-4, what=f"trig vs batch_evaluate, D={D}")
    # and against the CPU oracle (glibc): the usual tolerance for transcendental trees
    want, tol, unst = per_tree_tolerance(oracle, (v, t, s), X, y)
    assert_within_sensitivity(got, want, tol, unst, "trig vs oracle", max_unstable=0.15, min_tight=0.2)


@pytest.mark.parametrize("D", [64, 1024])
def test_sr_fitness_sqrt_exp_log_inv_handlers_match_the_register_kernels(g, oracle, rng, D):
    """sqrt / loose sqrt / exp / log / loose log in the threaded code are the device math library's sequences (full range,
    no bail-out); inv is the division handler with a constant numerator.  Same comparison as for sin / cos / tan: against
    the per-datapoint outputs of batch_evaluate, and against the CPU oracle with the transcendental tolerance."""
    funcs = [1, 2, 3, 4, 20, 21, 22, 23, 25, 26, 27, 28]
    f = oracle.generate(3000, 64, 5, 1, 0.0, 0.4, [12, 99], depth2leaf(6), roulette_uniform(funcs), [-1, 0, 1, 0.5, 3.0, -2.0])
    v, t, s = (a.copy() for a in f)
    # one function of one variable each (V form), of a product (S form), of a constant (folded)
    r = 0
    for fid in (20, 21, 22, 23, 27, 28):
        for child in ("var", "prod", "const"):
            v[r] = 0; t[r] = 0; s[r] = 0
            if child == "var":
                v[r, :2] = [fid, r % 5]; t[r, :2] = [2, 0]; s[r, :2] = [2, 1]
            elif child == "prod":
                v[r, :4] = [fid, 3, 0, 1]; t[r, :4] = [2, 3, 0, 0]; s[r, :4] = [4, 3, 1, 1]
            else:
                v[r, :4] = [1, fid, [-2.0, 0.0, 7.5][r % 3], 2]; t[r, :4] = [3, 2, 1, 0]; s[r, :4] = [4, 2, 1, 1]
            assert oracle.validate_tree(t[r], s[r]) == 0
            r += 1
    X = rng.uniform(-4, 4, (D, 5)).astype(np.float32)
    X[:, 3] = rng.uniform(-100, 100, D)                  # exp over- and underflows
    X[0, :] = [0.0, -0.0, np.inf, -np.inf, np.nan]
    X[1, :] = [1e-40, -1e-40, 1e38, 88.8, -104.0]          # denormals, exp at its range ends
    X[2, :] = [1e-30, 4.0, 2.5e-39, -88.8, 89.0]
    y = rng.uniform(-1, 1, (D, 1)).astype(np.float32)
    got = g.sr_fitness(v, t, s, X, y, True)
    outs = g.batch_evaluate(v, t, s, X, 1)[:, :, 0]
    with np.errstate(all="ignore"):
        d = outs - y[:, 0][None, :]
        ref = (d * d).astype(np.float64).mean(1).astype(np.float32)
    assert_close_classes(got, ref, 1e-4, what=f"sqrt/exp/log/inv vs batch_evaluate, D={D}")
    want, tol, unst = per_tree_tolerance(oracle, (v, t, s), X, y)
    assert_within_sensitivity(got, want, tol, unst, "sqrt/exp/log/inv vs oracle", max_unstable=0.15, min_tight=0.2)
