#!/usr/bin/env python3
"""How does the fitness path behave on EVOLVED populations (bloat, deeper operand stacks)?  Runs the default GP loop on
configs[1] for a number of generations and prints, per generation: mean tree length, launch time of tree_SR_fitness,
whole-generation time, best fitness, and the share of trees the threaded-code path handed to the register kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
sys.argv = [sys.argv[0]]
import bench
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import GenerateDescriptor

dev = torch.device("cuda", 0)
gens = int(os.environ.get("GENS", "40"))
forest, Xd, yd, X, y = bench.sr_inputs(0, 100_000, dev)
mdesc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=3, const_samples=[-1, 0, 1])
algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, mdesc), DefaultSelection(0.3, elite_rate=0.01))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for g in range(gens):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record()
    fit = algo.forest.SR_fitness(Xd, yd)
    e1.record()
    f = torch.where(torch.isnan(fit), torch.full_like(fit, float('-inf')), -fit)
    sizes = algo.forest.batch_subtree_size[:, 0].float()
    algo.step(f)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    if g % 4 == 0 or g == gens - 1:
        print(f"gen {g:3d}  mean len {float(sizes.mean()):5.1f}  max {int(sizes.max()):2d}  fitness launch {e0.elapsed_time(e1):.3f} ms  generation {dt:.3f} ms  "
              f"best {float(f.max()):.4g}  nan share {float(torch.isnan(fit).float().mean()):.3f}", flush=True)
