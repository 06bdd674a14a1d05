#!/usr/bin/env python3
"""Cycle accounting of the threaded-code fitness kernel on configs[1] (profiling hook of the C ABI)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd
from evogp_amd import _lib
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
forest, Xd, yd, X, y = bench.c2_inputs(0, 100_000, dev)
stats = torch.zeros(8, dtype=torch.int64, device=dev)
for _ in range(3):
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(stats.data_ptr())
reps = 10
for _ in range(reps):
    forest.SR_fitness(Xd, yd)
torch.cuda.synchronize()
_lib.lib.evogp_hip_debug_set_stats(None)
c = stats.cpu().tolist()
asm, loop, wait, trees, nodes, kern, waves = c[:7]
print(json.dumps({
    "waves_per_launch": waves / reps, "trees_per_wave": trees / waves, "nodes_per_tree": nodes / max(trees, 1),
    "cycles_per_wave_kernel": kern / waves, "cycles_asm_per_tree": asm / max(trees, 1), "cycles_asm_per_node": asm / max(nodes, 1),
    "cycles_loop_per_tree": loop / max(trees, 1), "cycles_barrier_wait_per_tree": wait / max(trees, 1),
    "frac_asm": asm / kern, "frac_loop": loop / kern, "frac_wait": wait / kern}))
