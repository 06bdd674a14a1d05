#!/usr/bin/env python3
"""Randomised sweep of evogp_hip_sr_fitness against the CPU oracle (test infrastructure, run by hand on a GPU box; not collected
by pytest):   python tests/tools/fuzz_sr.py [iterations] [seed]

Every iteration draws a shape -- population, row length, inputs, outputs, datapoints (so all three interpreter builds, the
pieces path and ragged tiles come up), an IEEE-exact function subset, constants with special values -- generates the forest
with the oracle's generator, breeds it once (crossover products reach the length cap), and compares both losses at 1e-5 with
identical NaN / inf classes."""
import os
import sys

import numpy as np

TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(TESTS)); sys.path.insert(0, TESTS)
from helpers import assert_close_classes, depth2leaf, roulette_uniform  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
import gpu_capi as g  # noqa: E402

IF, ADD, SUB, MUL, DIV, LDIV, POW, LPOW, MAX, MIN, LT, GT, LE, GE = range(14)
SIN, COS, TAN, SINH, COSH, TANH, LOG, LLOG, EXP, INV, LINV, NEG, ABS, SQRT, LSQRT = range(14, 29)
EXACT = [IF, ADD, SUB, MUL, DIV, LDIV, MAX, MIN, LT, GT, LE, GE, INV, LINV, NEG, ABS, SQRT, LSQRT]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    o = Oracle("port")
    for it in range(iters):
        L = int(rng.choice([8, 16, 31, 32, 64, 65, 100, 128, 200, 256]))
        funcs = sorted(set([ADD, DIV] + list(rng.choice(EXACT, size=int(rng.integers(1, 7)), replace=False))))
        has_if = IF in funcs
        mlc = 2
        while ((3 if has_if else 2) ** (mlc + 1) - 1) // (2 if has_if else 1) <= L and mlc < 9:
            mlc += 1
        out_len = int(rng.choice([1, 1, 1, 2, 3, 5]))
        var_len = int(rng.choice([1, 2, 3, 7, 10, 16, 33, 64]))
        D = int(rng.choice([1, 3, 8, 33, 64, 65, 200, 256, 257, 700, 1024, 1500, 6000]))
        pop = int(rng.choice([1, 5, 64, 300, 1000, 2500, 7000]))
        consts = [-1.0, 0.0, 1.0, 0.5, 2.0, 3.0, -0.25, 1e-10, 1e20]
        forest = o.generate(pop, L, var_len, out_len, 0.5, 0.5, [it, seed], depth2leaf(mlc, 0.1), roulette_uniform(funcs), consts)
        if pop >= 2:   # one round of crossover: products up to exactly L nodes
            sizes = forest[2][:, 0].astype(np.int64)
            li, ri = rng.integers(0, pop, pop).astype(np.int32), rng.integers(0, pop, pop).astype(np.int32)
            ln = (rng.integers(0, 1 << 30, pop) % sizes[li]).astype(np.int32); rn = (rng.integers(0, 1 << 30, pop) % sizes[ri]).astype(np.int32)
            forest = o.crossover(*forest, li, ri, ln, rn)
        X = rng.uniform(-3, 3, (D, var_len)).astype(np.float32)
        X[rng.random((D, var_len)) < 0.05] = 0.0
        y = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
        what = f"it {it}: pop {pop} L {L} in {var_len} out {out_len} D {D} funcs {funcs}"
        for mse in (True, False):
            got, want = g.sr_fitness(*forest, X, y, mse), o.sr_fitness(*forest, X, y, mse)
            assert not (got == 12345.0).any(), what + ": trees not evaluated"
            assert_close_classes(got, want, 1e-5, what=what + f" mse={mse}")
        print("ok", what, flush=True)
    print(f"{iters} random shapes: all agree with the oracle")


if __name__ == "__main__":
    main()
