#!/usr/bin/env python3
"""Launch time of the SR-fitness path for forests with different operator mixes (what bounds the interpreter?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_capi as g
from helpers import c2_dataset, depth2leaf, roulette_uniform
from oracle.pyoracle import Oracle
o = Oracle("port")
X, y = c2_dataset()
pop = 100_000
def timed(f, X, y, reps=10):
    a = [g.dev(f[0], np.float32), g.dev(f[1], np.int16), g.dev(f[2], np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]
    fit = torch.empty(pop, dtype=torch.float32, device=g.DEV)
    D, vl = X.shape
    def run():
        rc = g.L.evogp_hip_sr_fitness(pop, D, f[0].shape[1], vl, 1, 1, *[x.data_ptr() for x in a], fit.data_ptr(), 0, g._stream()); assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, funcs, consts, cp in (("+-*/ (configs[1])", [1, 2, 3, 4], [-1, 0, 1], 0.5), ("+-* only", [1, 2, 3], [-1, 0, 1], 0.5), ("/ only", [4], [-1, 0, 1], 0.5),
                                ("+ only, variables only", [1], [1], 0.0), ("+ only, constants only", [1], [-1, 0, 1], 1.0)):
    f = o.generate(pop, 64, 10, 1, 0.5, cp, [42, 0], depth2leaf(6), roulette_uniform(funcs), consts)
    nodes = f[2][:, 0].astype(np.int64).sum()
    ms = timed(f, X, y)
    print(f"{name:28s} mean len {nodes / pop:6.2f}  launch {ms:.4f} ms  {pop * 1024 / ms / 1e6:.1f} G tree-evals/s", flush=True)
for D in (256, 512, 2048):
    Xd, yd = c2_dataset(D=D)
    f = o.generate(pop, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    ms = timed(f, Xd, yd)
    print(f"+-*/ D={D:5d}                 launch {ms:.4f} ms  {pop * D / ms / 1e6:.1f} G tree-evals/s", flush=True)
