"""Fitness words of forests beyond 120 k trees -- where the packed compiler runs one wave per workgroup -- for a checksum line per case;
run under EVOGP_TC_PACKED_WG=256 and =64 and diff (multi-output trees, long rows, generic functions: the modes scripts/dbg/pool_soak.py does not reach)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); torch.cuda.set_device(0); set_default_device(dev)
cases = [(200_000, 64, 4, ["+", "-", "*", "/"], 5), (150_000, 128, 1, ["+", "-", "*", "/", "sin"], 6), (130_000, 256, 6, ["+", "-", "*", "/"], 6),
         (250_000, 32, 1, ["+", "-", "*", "/", "max", "min", "if"], 3), (121_000, 64, 10, ["+", "*", "neg"], 5)]
for i, (pop, L, out, funcs, mlc) in enumerate(cases):
    desc = GenerateDescriptor(max_tree_len=L, input_len=7, output_len=out, using_funcs=funcs, max_layer_cnt=mlc, const_samples=[-1, 0, 1, 0.5])
    f = Forest.random_generate(pop, desc, keys=torch.tensor([77 + i, 3], dtype=torch.uint32, device=dev))
    g = torch.Generator(device="cpu").manual_seed(i)
    X = (torch.rand((600, 7), generator=g) * 6 - 3).to(dev); y = (torch.rand((600, out), generator=g) * 2 - 1).to(dev)
    a = f.SR_fitness(X, y); b = f.SR_fitness(X, y)
    w = torch.where(torch.isnan(a), torch.full_like(a, 7.0), a).view(torch.int32).to(torch.int64)
    print(f"case {i} pop {pop} L {L} out {out} {'|'.join(funcs)}: repeat equal {bool(torch.equal(w, torch.where(torch.isnan(b), torch.full_like(b, 7.0), b).view(torch.int32).to(torch.int64)))} sum {int(w.sum())} wsum {int((w * torch.arange(1, pop + 1, device=dev)).sum())}", flush=True)
