"""Which trees of the example/uci_sr.py-shaped forest differ from the oracle beyond their own sensitivity, and what are they?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_capi as g
from helpers import PAPER7, c2_dataset, per_tree_tolerance, roulette_uniform
from oracle.pyoracle import Oracle

oracle = Oracle("port")
d2l = np.array([0.3] * 8 + [1.0] * 2, np.float32)
cs = np.random.default_rng(5).uniform(-5, 5, 10_000).astype(np.float32)
pop = 100_000
f = g.generate(pop, 512, 10, 1, 0.5, 0.5, [42, 0], d2l, roulette_uniform(PAPER7), cs)
X, y = c2_dataset()
full = g.sr_fitness(*f, X, y)
pick = np.sort(np.random.default_rng(6).choice(pop, 5000, replace=False))
sub = tuple(a[pick] for a in f)
want, tol, unstable = per_tree_tolerance(oracle, sub, X, y)
got = full[pick].astype(np.float64)
fin = ~unstable & np.isfinite(want)
bad = np.flatnonzero(fin & (np.abs(got - want) > tol))
print("bad entries:", bad, "of", len(pick))
names = {1: "+", 2: "-", 3: "*", 4: "/", 14: "sin", 15: "cos", 16: "tan"}
for b in bad[:5]:
    t = pick[b]
    n = int(f[2][t, 0])
    print(f"tree {t} (sample entry {b}): len {n}, got {got[b]!r}, want {want[b]!r}, tol {tol[b]:.4g}")
    print("  nodes:", " ".join((f"x{int(f[0][t, i])}" if f[1][t, i] == 0 else f"{f[0][t, i]:.6g}" if f[1][t, i] == 1 else names.get(int(f[0][t, i]), f"f{int(f[0][t, i])}")) + f"/{f[2][t, i]}" for i in range(n)))
    one = tuple(a[t:t + 1] for a in f)
    print("  alone:", g.sr_fitness(*one, X, y), " batch_evaluate mean sq:", np.mean((g.batch_evaluate(*one, X, 1)[0, :, 0].astype(np.float64) - y[:, 0]) ** 2))
    pred = oracle.batch_evaluate(*one, X, 1)[0, :, 0]
    gp = g.batch_evaluate(*one, X, 1)[0, :, 0]
    d = np.abs(pred.astype(np.float64) - gp)
    k = np.argsort(-np.nan_to_num(d))[:3]
    print("  rows with the largest difference:", [(int(i), float(pred[i]), float(gp[i])) for i in k])
