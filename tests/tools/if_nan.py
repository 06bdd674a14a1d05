"""debug: trees of the IF function set whose fitness differs from the oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import gpu_capi as g
from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform
o = Oracle("port")
rng = np.random.default_rng(20260925)
for funcs, out_len, D in (([0, 1, 2, 10], 1, 520), (list(range(0, 6)) + [8, 9, 10, 11, 12, 13, 23, 24, 25, 26, 27, 28], 4, 1024)):
    if out_len == 1:
        forest = o.generate(6000, 64, 4, 1, 0.5, 0.4, [sum(funcs), 1], depth2leaf(5, 0.15), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5, 2.0, 1e-10, -1e-10])
        X = rng.uniform(-2, 2, (D, 4)).astype(np.float32); y = rng.uniform(-2, 2, (D, 1)).astype(np.float32)
    else:
        forest = o.generate(5000, 64, 7, out_len, 0.5, 0.5, [out_len, D], depth2leaf(6, 0.15), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5, 2.0])
        X = rng.uniform(-2, 2, (D, 7)).astype(np.float32); y = rng.uniform(-2, 2, (D, out_len)).astype(np.float32)
    got = g.sr_fitness(*forest, X, y); want = o.sr_fitness(*forest, X, y)
    bad = np.flatnonzero((np.isnan(got) != np.isnan(want)) | (np.isfinite(want) & (np.abs(got - want) > 1e-5 * np.abs(want))))
    print("funcs", funcs, "out", out_len, "mismatches", len(bad), bad[:10])
    for i in bad[:4]:
        n = forest[2][i, 0]
        print(" tree", i, "len", n, "got", got[i], "want", want[i])
        print("  type", forest[1][i, :n].tolist()); print("  val ", [float(x) if forest[1][i, k] & 0x80 == 0 else hex(np.float32(x).view(np.uint32)) for k, x in enumerate(forest[0][i, :n])]); print("  size", forest[2][i, :n].tolist())
        one = tuple(a[i:i + 1] for a in forest)
        print("  alone:", g.sr_fitness(*one, X, y), " first 8 rows:", g.sr_fitness(*one, X[:8], y[:8]), o.sr_fitness(*one, X[:8], y[:8]))
