#!/usr/bin/env python3
"""tree_SR_fitness on configs[1]-sized forests of other function sets and output counts (the paths outside the threaded
code): launch time and tree-evals/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); set_default_device(dev)
g = torch.Generator(device="cpu").manual_seed(1234)
X = (torch.rand(1024, 10, generator=g) * 10 - 5).to(dev)


def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, funcs, out_len in (("+ - * /", ["+", "-", "*", "/"], 1), ("+ - * / sin cos tan (uci_sr.py)", ["+", "-", "*", "/", "sin", "cos", "tan"], 1),
                             ("+ - * / neg abs sqrt", ["+", "-", "*", "/", "neg", "abs", "sqrt"], 1),
                             ("+ - * / exp log pow", ["+", "-", "*", "/", "exp", "log", "pow"], 1),
                             ("+ - * /, 4 outputs", ["+", "-", "*", "/"], 4)):
    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=out_len, using_funcs=funcs, max_layer_cnt=6, const_samples=[-1, 0, 1])
    forest = Forest.random_generate(100_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
    y = torch.randn(1024, out_len, device=dev)
    ms = timed(lambda: forest.SR_fitness(X, y, True, "auto"))
    print(f"| {name} | {float(forest.batch_subtree_size[:, 0].float().mean()):.1f} | {ms:.3f} | {100_000 * 1024 / ms / 1e6:.0f} G |", flush=True)
