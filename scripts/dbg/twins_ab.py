"""The handlers' twins that do not prefetch, always against never (evogp_hip_debug_twins), alternating in ONE process: time of a fitness
call on the first N trees of the headline population for several N.  python scripts/dbg/twins_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from evogp_amd import _lib

dev = torch.device("cuda", 0)
for n in (60_000, 100_000, 125_000, 180_000, 250_000, 1_000_000):
    forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
    res = {0: [], 1: []}
    for rnd in range(6):
        for tw in (0, 1):
            assert _lib.lib.evogp_hip_debug_twins(tw) == 0
            for _ in range(10):
                forest.SR_fitness(Xd, yd, True, "auto")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 100 if n <= 250_000 else 30
            e0.record()
            for _ in range(reps):
                forest.SR_fitness(Xd, yd, True, "auto")
            e1.record(); torch.cuda.synchronize()
            res[tw].append(e0.elapsed_time(e1) / reps * 1e3)
    _lib.lib.evogp_hip_debug_twins(-1)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{n:8d} trees: never {med[0]:7.1f} us  always {med[1]:7.1f} us  ({med[1] - med[0]:+.1f})   rounds never {[round(x, 1) for x in res[0]]} always {[round(x, 1) for x in res[1]]}", flush=True)
