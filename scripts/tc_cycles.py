#!/usr/bin/env python3
"""Cycle accounting of the threaded-code fitness kernel on configs[1] (profiling hook of the C ABI; the accounting
build of the interpreter adds s_memtime pairs around its waits)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
POP = int(os.environ.get("POP", "100000"))
forest, Xd, yd, X, y = bench.sr_inputs(0, POP, dev)
stats = torch.zeros(8, dtype=torch.int64, device=dev)
for _ in range(3):
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(stats.data_ptr())
reps = 5
for _ in range(reps + 1):   # the first call queries the handler table of the accounting build
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(None)
c = stats.cpu().tolist()
rec, work, trees, disp4, ticks, waves = c[:6]
print(json.dumps({"waves_per_launch": waves / (reps + 1), "trees_per_wave": trees / waves, "later_block_wait_ticks_per_tree": disp4 / trees,
                  "ticks_per_wave": ticks / waves, "frac_record_wait": rec / ticks, "frac_work_wait": work / ticks,
                  "record_wait_ticks_per_tree": rec / trees, "work_wait_ticks_per_wave": work / waves,
                  "ticks_per_tree": (ticks - work) / trees}))
