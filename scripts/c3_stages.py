#!/usr/bin/env python3
"""BASELINE configs[3] (classifier trees, pop 200 k, 10 outputs, L 128, sklearn digits 1797 x 64): the fitness pass alone, PASSES times, for a
kernel trace (rocprofv3 --kernel-trace --stats): which kernels the 1.7 ms are."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.problem import Classification
from evogp_amd.tree import Forest, GenerateDescriptor

dev = torch.device("cuda", 0)
desc = GenerateDescriptor(max_tree_len=128, input_len=64, output_len=10, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
prob = Classification(dataset="digits")
forest = Forest.random_generate(200_000, desc, keys=torch.tensor([7, 0], dtype=torch.uint32, device=dev))
for _ in range(3):
    prob.evaluate(forest)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("PASSES", "10"))
for _ in range(n):
    prob.evaluate(forest)
torch.cuda.synchronize()
print(f"configs[3] fitness pass: {(time.perf_counter() - t0) / n * 1e3:.3f} ms; mean tree length {float(forest.batch_subtree_size[:, 0].float().mean()):.1f}")
