#!/usr/bin/env python3
"""configs[1] generation loop (fitness + default operators, pop 100 k) for `rocprofv3 --kernel-trace --stats`: which kernels a
generation consists of and what each costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import evogp_amd  # noqa: F401
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); set_default_device(dev)
forest, X, y, _, _ = bench.sr_inputs(0, 100_000, dev)
mdesc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=3, const_samples=[-1, 0, 1])
algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, mdesc), DefaultSelection(0.3, elite_rate=0.01))
neg = torch.full((100_000,), float("-inf"), device=dev)
import time
for g in range(25):
    if g == 5: torch.cuda.synchronize(); t0 = time.perf_counter()
    f = -algo.forest.SR_fitness(X, y, True, "auto")
    algo.step(torch.where(torch.isnan(f), neg, f))
torch.cuda.synchronize()
print("ms per generation (20 generations):", (time.perf_counter() - t0) / 20 * 1e3)
