#!/usr/bin/env python3
"""Where the waves of sr_fused_kernel spend their clocks (a library built with -DEVOGP_FUSED_STATS): the wait for a batch's first
nodes, the compilation, the wait for the work counter's answer, the assembly block.   POP=1000000 python scripts/fused_cycles.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
POP = int(os.environ.get("POP", "1000000"))
forest, Xd, yd, X, y = bench.sr_inputs(0, POP, dev)
stats = torch.zeros(16, dtype=torch.int64, device=dev)
for _ in range(3):
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(stats.data_ptr())
reps = 5
for _ in range(reps):
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(None)
first, comp, grab, asm, trees, batches, ticks, waves, cpc, plan = stats.cpu().tolist()[:10]
if waves == 0:
    print(json.dumps({"error": "no counters: the library was not built with -DEVOGP_FUSED_STATS (or the call did not take the fused kernel)"}))
    sys.exit(0)
print(json.dumps({"pop": POP, "waves_per_launch": waves / reps, "trees_per_wave": trees / waves, "batches_per_wave": batches / waves,
                  "clocks_per_wave": ticks / waves, "frac_first_nodes": first / ticks, "frac_compile": comp / ticks, "frac_grab_wait": grab / ticks,
                  "frac_asm": asm / ticks, "frac_outside": 1 - (first + comp + grab + asm) / ticks,
                  "clocks_per_batch": {"args_copy": cpc / batches, "plan_first_pass": plan / batches, "first_nodes_all": first / batches, "asm": asm / batches, "all": ticks / batches},
                  "clocks_per_tree": {"first_nodes": first / trees, "compile": comp / trees, "grab": grab / trees, "asm": asm / trees, "all": ticks / trees}}))
