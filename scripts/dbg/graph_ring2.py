"""debug: which captured fitness paths survive replays?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_capi as g
import evogp_amd
from evogp_amd.tree import Forest, GenerateDescriptor
dev = torch.device("cuda", 0)
desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
pop = 300_000
f = Forest.random_generate(pop, desc, keys=torch.tensor([3, 4], dtype=torch.uint32, device=dev))
plain = Forest(f.input_len, f.output_len, f.batch_node_value, f.batch_node_type, f.batch_subtree_size)
X = torch.rand(1024, 4, device=dev) * 4 - 2
y = (X[:, 0] - X[:, 1] * X[:, 2]).unsqueeze(1).contiguous()
ref = f.SR_fitness(X, y).clone()
torch.cuda.synchronize()
for name, forest in (("masked (fused when EVOGP_TC_FUSED=1)", f), ("plain", plain)):
    main = torch.cuda.current_stream()
    cap = torch.cuda.Stream(); cap.wait_stream(main)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        forest.SR_fitness(X, y); cap.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            out = forest.SR_fitness(X, y)
    for i in range(3):
        out.fill_(555.0)
        graph.replay(); torch.cuda.synchronize()
        o = out
        same = (o.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(o) & torch.isnan(ref))
        print(f"{name}: replay {i}: untouched {(o == 555.0).sum().item()}, differ {(~same).sum().item()}; rings {evogp_amd.record_ring_bytes()} records {evogp_amd.program_buffer_bytes()}")
