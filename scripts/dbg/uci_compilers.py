"""The uci_sr.py-shaped forest (L 512, + - * / sin cos tan, 10 000 constants), fresh and after some generations: fitness words under the
one-tree program compiler and under the packed one (evogp_hip_debug_compile_batch) -- which trees differ, and how."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, GeneticProgramming
from evogp_amd.algorithm.selection import TournamentSelection
from evogp_amd.tree import Forest, GenerateDescriptor

sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(42)
POP = 100_000
_, Xd, yd, X, y = bench.sr_inputs(0, 1000, dev)
desc = GenerateDescriptor(max_tree_len=512, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"],
                          max_layer_cnt=9, const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
algo = GeneticProgramming(Forest.random_generate(POP, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)),
                          DefaultCrossover(), DefaultMutation(0.1, desc.update(max_layer_cnt=4)), TournamentSelection(20, 0.5, 0.1))
neg = torch.full((POP,), float("-inf"), device=dev)
for g in range(0, 13):
    if g in (0, 4, 8, 12):
        f = algo.forest
        w = {}
        for b in (0, -1):
            _lib.check(_lib.lib.evogp_hip_debug_compile_batch(b), "batch")
            w[b] = f.SR_fitness(Xd, yd).cpu().numpy().view(np.uint32).copy()
        _lib.lib.evogp_hip_debug_compile_batch(-1)
        d = np.nonzero(w[0] != w[-1])[0]
        sz = f.batch_subtree_size[:, 0].cpu().numpy()
        print(f"gen {g}: mean len {sz.mean():.1f}; {len(d)} fitness words differ between the compilers", d[:8], w[0][d[:4]], w[-1][d[:4]], "lens", sz[d[:8]])
        if len(d):
            t = int(d[0]); n = int(sz[t])
            ty = f.batch_node_type[t, :n].cpu().numpy(); va = f.batch_node_value[t, :n].cpu().numpy(); ss = f.batch_subtree_size[t, :n].cpu().numpy()
            print("   first:", " ".join(f"{int(a)}/{b:g}/{int(c)}" for a, b, c in zip(ty, va, ss))[:1500])
    fit = -algo.forest.SR_fitness(Xd, yd)
    algo.step(torch.where(torch.isnan(fit), neg, fit))
