"""debug: captured fitness call replayed, several population sizes"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import evogp_amd
from evogp_amd.tree import Forest, GenerateDescriptor
dev = torch.device("cuda", 0)
desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
X = torch.rand(1024, 4, device=dev) * 4 - 2
y = (X[:, 0] - X[:, 1] * X[:, 2]).unsqueeze(1).contiguous()
for pop in (3000, 20000, 60000, 100000, 300000):
    f = Forest.random_generate(pop, desc, keys=torch.tensor([3, 4], dtype=torch.uint32, device=dev))
    ref = f.SR_fitness(X, y).clone(); torch.cuda.synchronize()
    main = torch.cuda.current_stream()
    cap = torch.cuda.Stream(); cap.wait_stream(main)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        f.SR_fitness(X, y); cap.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            out = f.SR_fitness(X, y)
    res = []
    for i in range(3):
        out.fill_(555.0)
        graph.replay(); torch.cuda.synchronize()
        res.append(int((out == 555.0).sum().item()))
    print(f"pop {pop}: untouched per replay {res}")
