"""CPU: fuzz the plain-C oracle against the reference's own device code compiled for the host
(oracle/_ref).  Skipped where /root/reference was never available (the GPU box uses the committed
golden vectors instead)."""
import numpy as np
import pytest

from helpers import (ALLF, assert_forest_equal, avoid_middle_child_roots, bits, fbits, depth2leaf, middle_child_roots,
                     random_crossover_indices, roulette_uniform)


@pytest.mark.parametrize("seed", range(120))
def test_fuzz_all_ops(oracle, reference, seed):
    rng = np.random.default_rng(1000 + seed)
    out_len = int(rng.choice([1, 1, 2, 5]))
    var_len = int(rng.integers(1, 12))
    funcs = list(rng.choice(ALLF, int(rng.integers(1, 10)), replace=False))
    rou = roulette_uniform(funcs)
    mlc = int(rng.integers(2, 7))
    arity = 3 if 0 in funcs else (2 if any(f <= 13 for f in funcs) else 1)
    L = max(8, (arity**mlc - 1) // (arity - 1) if arity > 1 else mlc)
    L = min(1024, L + int(rng.integers(0, 30)))
    d2l = depth2leaf(mlc, float(rng.uniform(0, 0.6)))
    cs = rng.uniform(-3, 3, int(rng.integers(1, 8))).astype(np.float32)
    keys = rng.integers(0, 2**32, 2, dtype=np.uint64).astype(np.uint32)
    op, cp = float(rng.uniform()), float(rng.uniform())
    pop = 200
    a = oracle.generate(pop, L, var_len, out_len, op, cp, keys, d2l, rou, cs)
    b = reference.generate(pop, L, var_len, out_len, op, cp, keys, d2l, rou, cs)
    assert_forest_equal(a, b, "generate", live_only=True)
    X = rng.uniform(-3, 3, (pop, var_len)).astype(np.float32)
    assert np.array_equal(fbits(oracle.evaluate(*a, X, out_len)), fbits(reference.evaluate(*a, X, out_len)))
    D = int(rng.choice([1, 7, 64, 100, 1025, 2048, 2500] if seed >= 24 else [1, 7, 64, 100]))   # beyond one 1024-row block: forward.cu:456-471
    Xd = rng.uniform(-3, 3, (D, var_len)).astype(np.float32)
    yd = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
    mse = bool(rng.integers(0, 2))
    assert np.array_equal(fbits(oracle.sr_fitness(*a, Xd, yd, mse)), fbits(reference.sr_fitness(*a, Xd, yd, mse)))
    sizes = a[2][:, 0].astype(np.int64)
    idx = list(random_crossover_indices(rng, sizes, 400))
    idx[2] = avoid_middle_child_roots(a[1], a[2], idx[0], idx[2])       # undefined in the reference (helpers.middle_child_roots)
    assert_forest_equal(oracle.crossover(*a, *idx), reference.crossover(*a, *idx), "crossover", live_only=True)
    new = oracle.generate(pop, L, var_len, out_len, op, cp, keys[::-1].copy(), depth2leaf(max(2, mlc - 2)), rou, cs)
    mi = (rng.integers(0, 1024, pop) % sizes).astype(np.int32)
    mi = avoid_middle_child_roots(a[1], a[2], np.arange(pop), mi)
    mi[:3] = [-1, 2000, 0]
    assert_forest_equal(oracle.mutate(*a, mi, *new), reference.mutate(*a, mi, *new), "mutate", live_only=True)


def test_sr_reduction_order_beyond_one_block(oracle, reference):
    """D > 1024: pairwise tree inside each 1024-block, blocks added in order (forward.cu:456-471)."""
    rng = np.random.default_rng(5)
    a = oracle.generate(40, 64, 4, 1, 0.5, 0.5, [3, 4], depth2leaf(5), roulette_uniform([1, 2, 3]), [-1, 0.5, 2])
    X = rng.uniform(-2, 2, (2500, 4)).astype(np.float32)
    y = rng.uniform(-2, 2, (2500, 1)).astype(np.float32)
    assert np.array_equal(fbits(oracle.sr_fitness(*a, X, y, True)), fbits(reference.sr_fitness(*a, X, y, True)))


@pytest.mark.parametrize("out_len,funcs", [(1, [1, 2, 3, 4]), (4, [0, 1, 2, 3, 4, 14, 18]), (10, [1, 2, 3, 4, 9, 10])])
def test_long_rows_and_multi_output(oracle, reference, out_len, funcs):
    """L = 1024 (MAX_STACK): trees of up to ~1000 nodes, single and multi output, every op"""
    rng = np.random.default_rng(77 + out_len)
    L, pop, var_len = 1024, 60, 6
    d2l = depth2leaf(9, 0.05)
    rou = roulette_uniform(funcs)
    a = oracle.generate(pop, L, var_len, out_len, 0.4, 0.5, [9, out_len], d2l, rou, [-1.0, 0.5, 2.0])
    b = reference.generate(pop, L, var_len, out_len, 0.4, 0.5, [9, out_len], d2l, rou, [-1.0, 0.5, 2.0])
    assert_forest_equal(a, b, "generate", live_only=True)
    assert a[2][:, 0].max() > 300
    X = rng.uniform(-2, 2, (pop, var_len)).astype(np.float32)
    assert np.array_equal(fbits(oracle.evaluate(*a, X, out_len)), fbits(reference.evaluate(*a, X, out_len)))
    for D in (64, 1025, 2048):
        Xd = rng.uniform(-2, 2, (D, var_len)).astype(np.float32)
        yd = rng.uniform(-2, 2, (D, out_len)).astype(np.float32)
        for mse in (True, False):
            assert np.array_equal(fbits(oracle.sr_fitness(*a, Xd, yd, mse)), fbits(reference.sr_fitness(*a, Xd, yd, mse))), (D, mse)
    sizes = a[2][:, 0].astype(np.int64)
    idx = list(random_crossover_indices(rng, sizes, 300))
    idx[2] = avoid_middle_child_roots(a[1], a[2], idx[0], idx[2])
    assert_forest_equal(oracle.crossover(*a, *idx), reference.crossover(*a, *idx), "crossover", live_only=True)
    new = oracle.generate(pop, L, var_len, out_len, 0.4, 0.5, [5, 5], depth2leaf(4), rou, [-1.0, 0.5, 2.0])
    mi = avoid_middle_child_roots(a[1], a[2], np.arange(pop), (rng.integers(0, 1024, pop) % sizes).astype(np.int32))
    assert_forest_equal(oracle.mutate(*a, mi, *new), reference.mutate(*a, mi, *new), "mutate", live_only=True)


def test_configs1_scale(oracle, reference):
    """BASELINE configs[1] at its own scale: the 100 k-tree forest of bench.py (generate, live prefix), the SR fitness of 3 000 of
    its trees on the 1024-row dataset, 50 k crossovers — oracle and reference agree bit for bit (VERDICT r02 ran this by hand)."""
    from helpers import c2_dataset

    args = (100_000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    a = oracle.generate(*args)
    b = reference.generate(*args)
    assert_forest_equal(a, b, "generate", live_only=True)
    assert abs(float(a[2][:, 0].mean()) - 26.26) < 0.05
    X, y = c2_dataset()
    sub = tuple(x[:3000] for x in a)
    assert np.array_equal(fbits(oracle.sr_fitness(*sub, X, y, True)), fbits(reference.sr_fitness(*sub, X, y, True)))
    rng = np.random.default_rng(3)
    idx = random_crossover_indices(rng, a[2][:, 0].astype(np.int64), 50_000)
    assert_forest_equal(oracle.crossover(*a, *idx), reference.crossover(*a, *idx), "crossover", live_only=True)


def test_middle_child_positions_are_the_intended_result_in_the_oracle(oracle):
    """the positions the reference leaves undefined (root of a ternary node's middle operand): the oracle updates exactly the
    ancestors' sizes, like everywhere else — the result is a well-formed tree"""
    rng = np.random.default_rng(9)
    a = oracle.generate(300, 64, 3, 1, 0.5, 0.5, [8, 8], depth2leaf(4, 0.1), roulette_uniform([0, 1]), [1.0])
    mids = middle_child_roots(a[1], a[2])
    assert mids.sum() > 100
    t, pos = np.nonzero(mids)
    new = oracle.generate(len(t), 64, 3, 1, 0.5, 0.5, [1, 1], depth2leaf(2), roulette_uniform([1]), [2.0])
    sub = tuple(x[t] for x in a)
    out = oracle.mutate(*sub, pos.astype(np.int32), *new)
    for k in range(len(t)):
        assert oracle.validate_tree(out[1][k], out[2][k]) == 0
        grown = int(out[2][k, 0]) - int(sub[2][k, 0])
        assert grown in (0, int(new[2][k, 0]) - int(sub[2][k, pos[k]]))
