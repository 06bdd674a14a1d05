#!/usr/bin/env python3
"""The notebook configuration (scripts/vis_profile.py) evolved for 30 generations, then ten fitness calls on that forest:
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of an evolved small-dataset call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); set_default_device(dev)
POP = 100_000
desc = GenerateDescriptor(max_tree_len=128, input_len=3, output_len=1, using_funcs=["+", "-", "log", "sqrt", "pow", "/", "inv"],
                          max_layer_cnt=2, const_samples=[-1, 0, 1])
X = torch.tensor([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], device=dev)
y = (X.sum(1) % 2)[:, None].contiguous()
algo = GeneticProgramming(Forest.random_generate(POP, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)),
                          DefaultCrossover(), DefaultMutation(0.2, desc), DefaultSelection(0.3, elite_rate=0.01))
neg = torch.full((POP,), float("-inf"), dtype=torch.float32, device=dev)
for g in range(30):
    f = -algo.forest.SR_fitness(X, y, True, "auto")
    algo.step(torch.where(torch.isnan(f), neg, f))
torch.cuda.synchronize()
sizes = algo.forest.batch_subtree_size[:, 0].float()
print(f"generation 30: mean tree length {float(sizes.mean()):.1f}, share above 64 nodes {float((sizes > 64).float().mean()):.2f}")
torch.cuda.synchronize()
for _ in range(10):
    algo.forest.SR_fitness(X, y, True, "auto")
torch.cuda.synchronize()
