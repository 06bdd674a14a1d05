#!/usr/bin/env python3
"""The default GP loop on a `+ - * / sin cos tan` population (the function set of the reference's example/uci_sr.py):
per generation the fitness launch time, the generation time and the best fitness; at the end the best tree's fitness is
recomputed with batch_forward (the C++ interpreter) as a cross-check of the threaded-code handlers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); set_default_device(dev)
torch.manual_seed(0)
funcs = ["+", "-", "*", "/", "sin", "cos", "tan"]
desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=funcs, max_layer_cnt=6, const_samples=[-1, 0, 1, 0.5, 2])
g = torch.Generator(device="cpu").manual_seed(3)
X = (torch.rand(1024, 4, generator=g) * 6 - 3).to(dev)
y = (torch.sin(X[:, 0] * X[:, 1]) + 0.5 * torch.cos(X[:, 2]) * X[:, 3])[:, None].contiguous()
forest = Forest.random_generate(100_000, desc, keys=torch.tensor([9, 1], dtype=torch.uint32, device=dev))
algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.01))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gens = int(os.environ.get("GENS", "40"))
for gen in range(gens):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0.record(); fit = algo.forest.SR_fitness(X, y); e1.record()
    f = torch.where(torch.isnan(fit), torch.full_like(fit, float("-inf")), -fit)
    sizes = algo.forest.batch_subtree_size[:, 0].float()
    old = algo.forest
    algo.step(f)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    if gen % 5 == 0 or gen == gens - 1:
        print(f"gen {gen:3d}  mean len {float(sizes.mean()):5.1f}  fitness launch {e0.elapsed_time(e1):.3f} ms  generation {dt:.3f} ms  best {float(f.max()):.5g}", flush=True)
# the share of the evolved population the threaded code leaves to the register kernels (the call stops behind it: sentinels stay)
import ctypes
from evogp_amd import _lib
_lib.check(_lib.lib.evogp_hip_debug_profile(2), "profile")
words = old.SR_fitness(X, y).view(torch.int32)
st = (ctypes.c_float * 3)(); nc = ctypes.c_int(0)
_lib.check(_lib.lib.evogp_hip_debug_profile_read(st, ctypes.byref(nc)), "profile read")
_lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile")
left = ((words == 0x7FC0FEED) | (words == 0x7FC0BEEF) | (words == 0x7FC0DEED)).float().mean().item()
print(f"last population: {left:.4%} of the trees left to the register kernels; compilers {st[0] * 1e3:.0f} us, interpreter {st[1] * 1e3:.0f} us")
best = int(torch.argmax(f))
out = old[best:best + 1].batch_forward(X)[0, :, 0]
print("best tree:", old[best], " fitness", float(fit[best]), " recomputed", float(((out - y[:, 0]) ** 2).mean()))
