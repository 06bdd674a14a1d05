#!/usr/bin/env python3
"""The reference's published configuration (test/vis.ipynb:171,181: XOR-3d, pop 100 k, max_tree_len 128, 8 datapoints, functions
+ - log sqrt pow / inv, default operators): where a generation's time goes as the trees grow — fitness call (compilers /
interpreter / follow-up kernels) against the generation step."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

L = _lib.lib
dev = torch.device("cuda", 0); set_default_device(dev)
POP = int(os.environ.get("VIS_POP", 100_000))
desc = GenerateDescriptor(max_tree_len=128, input_len=3, output_len=1, using_funcs=["+", "-", "log", "sqrt", "pow", "/", "inv"],
                          max_layer_cnt=2, const_samples=[-1, 0, 1])
X = torch.tensor([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], device=dev)
y = (X.sum(1) % 2)[:, None].contiguous()
forest = Forest.random_generate(POP, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc), DefaultSelection(0.3, elite_rate=0.01))
neg = torch.full((POP,), float("-inf"), dtype=torch.float32, device=dev)


def ev(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("| generation | mean / longest tree | generation ms | fitness call ms | compilers | interpreter | follow-up kernels | best fitness |\n|---|---|---|---|---|---|---|---|")
for g in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = -algo.forest.SR_fitness(X, y, True, "auto")
    fit = torch.where(torch.isnan(f), neg, f)
    if g in (0, 5, 10, 20, 30, 39):
        torch.cuda.synchronize()
        sizes = algo.forest.batch_subtree_size[:, 0].float()
        fo = algo.forest
        call = ev(lambda: fo.SR_fitness(X, y, True, "auto"))
        L.evogp_hip_debug_profile(1)
        for _ in range(5): fo.SR_fitness(X, y, True, "auto")
        st = (ctypes.c_float * 3)(); n = ctypes.c_int(0)
        L.evogp_hip_debug_profile_read(st, ctypes.byref(n)); L.evogp_hip_debug_profile(0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        algo.step(fit); torch.cuda.synchronize(); step_ms = (time.perf_counter() - t1) * 1e3
        print(f"| {g} | {float(sizes.mean()):.1f} / {int(sizes.max())} | {call + step_ms:.2f} | {call:.3f} | {st[0]:.3f} | {st[1]:.3f} | {st[2]:.3f} | {float(fit.max()):.4g} |", flush=True)
    else:
        algo.step(fit)
