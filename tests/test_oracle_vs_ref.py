"""CPU: fuzz the plain-C oracle against the reference's own device code compiled for the host
(oracle/_ref).  Skipped where /root/reference was never available (the GPU box uses the committed
golden vectors instead)."""
import numpy as np
import pytest

from helpers import ALLF, assert_forest_equal, bits, fbits, depth2leaf, random_crossover_indices, roulette_uniform


@pytest.mark.parametrize("seed", range(24))
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
    D = int(rng.choice([1, 7, 64, 100]))
    Xd = rng.uniform(-3, 3, (D, var_len)).astype(np.float32)
    yd = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
    mse = bool(rng.integers(0, 2))
    assert np.array_equal(fbits(oracle.sr_fitness(*a, Xd, yd, mse)), fbits(reference.sr_fitness(*a, Xd, yd, mse)))
    sizes = a[2][:, 0].astype(np.int64)
    idx = random_crossover_indices(rng, sizes, 400)
    assert_forest_equal(oracle.crossover(*a, *idx), reference.crossover(*a, *idx), "crossover", live_only=True)
    new = oracle.generate(pop, L, var_len, out_len, op, cp, keys[::-1].copy(), depth2leaf(max(2, mlc - 2)), rou, cs)
    mi = (rng.integers(0, 1024, pop) % sizes).astype(np.int32)
    mi[:3] = [-1, 2000, 0]
    assert_forest_equal(oracle.mutate(*a, mi, *new), reference.mutate(*a, mi, *new), "mutate", live_only=True)


def test_sr_reduction_order_beyond_one_block(oracle, reference):
    """D > 1024: pairwise tree inside each 1024-block, blocks added in order (forward.cu:456-471)."""
    rng = np.random.default_rng(5)
    a = oracle.generate(40, 64, 4, 1, 0.5, 0.5, [3, 4], depth2leaf(5), roulette_uniform([1, 2, 3]), [-1, 0.5, 2])
    X = rng.uniform(-2, 2, (2500, 4)).astype(np.float32)
    y = rng.uniform(-2, 2, (2500, 1)).astype(np.float32)
    assert np.array_equal(fbits(oracle.sr_fitness(*a, X, y, True)), fbits(reference.sr_fitness(*a, X, y, True)))
