#!/usr/bin/env python3
"""Randomised cross-check of the threaded-code fitness path against the per-datapoint outputs of batch_evaluate (the C++
interpreter with the device math library): random function subsets, dataset sizes, variable counts, constants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_capi as g
from helpers import depth2leaf, roulette_uniform
from oracle.pyoracle import Oracle

o = Oracle("port")
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
ALL = [1, 2, 3, 4, 14, 15, 16, 20, 21, 22, 23, 25, 26, 27, 28]
bad = 0
for it in range(int(os.environ.get("ITERS", "30"))):
    k = int(rng.integers(2, len(ALL) + 1))
    funcs = sorted(set([1, 2] + list(rng.choice(ALL, k, replace=False))))
    var_len = int(rng.integers(1, 12)); D = int(rng.choice([1, 7, 64, 100, 256, 257, 777, 1024, 2000]))
    L = int(rng.choice([16, 32, 64])); mlc = int(rng.integers(2, 7)); pop = int(rng.integers(200, 3000))
    consts = rng.choice([-1, 0, 1, 0.5, -2.5, 3.0, 1e-3, 100.0], 4).astype(np.float32)
    f = o.generate(pop, L, var_len, 1, 0.0, float(rng.uniform(0.1, 0.6)), [int(rng.integers(1, 1 << 30)), it], depth2leaf(mlc), roulette_uniform(funcs), consts)
    X = (rng.standard_normal((D, var_len)) * float(rng.choice([0.5, 3.0, 50.0]))).astype(np.float32)
    y = rng.standard_normal((D, 1)).astype(np.float32)
    for mse in (True, False):
        got = g.sr_fitness(*f, X, y, mse).astype(np.float64)
        outs = g.batch_evaluate(*f, X, 1)[:, :, 0]
        with np.errstate(all="ignore"):
            d = outs - y[:, 0][None, :]
            ref = ((d * d) if mse else np.abs(d)).astype(np.float64).mean(1).astype(np.float32).astype(np.float64)
        cls = (np.isnan(got) != np.isnan(ref)) | (np.isposinf(got) != np.isposinf(ref))
        # the fitness is an fp32 sum (as in the reference and the oracle): D terms whose fp64 mean is finite can overflow it
        cls &= ~(np.isposinf(got) & np.isfinite(ref) & (ref * D > 3.0e38))
        fin = np.isfinite(got) & np.isfinite(ref)
        rel = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30)
        nb = int(cls.sum()) + int((rel > 1e-4).sum())
        bad += nb
        print(f"it {it:2d} funcs {funcs} var {var_len} D {D} L {L} pop {pop} mse {mse}: class mismatches {int(cls.sum())}, rel > 1e-4: {int((rel > 1e-4).sum())}, max rel {rel.max() if rel.size else 0:.2e}", flush=True)
print("FUZZ_OK" if bad == 0 else f"FUZZ_BAD {bad}")
