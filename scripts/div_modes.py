#!/usr/bin/env python3
"""IEEE vs fast division in the threaded-code fitness path: launch time and the distribution of the fitness differences
on configs[1] (random forest) and on an evolved forest.  Run on the GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import evogp_amd
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming

dev = torch.device("cuda", 0); set_default_device(dev)
torch.manual_seed(0)
desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
forest = Forest.random_generate(100_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
g = torch.Generator(device="cpu").manual_seed(1234)
X = (torch.rand(1024, 10, generator=g) * 10 - 5).to(dev)
y = (X[:, 0] * X[:, 1] + X[:, 2] * X[:, 3] - X[:, 4] + 0.5 * X[:, 5] ** 2)[:, None].contiguous()


def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def compare(forest, tag):
    out = {}
    fits = {}
    for mode in ("ieee", "fast", "short"):
        evogp_amd.set_sr_division(mode)
        fits[mode] = forest.SR_fitness(X, y, True, "auto").double().cpu().numpy()
        out[f"launch_ms_{mode}"] = timed(lambda: forest.SR_fitness(X, y, True, "auto"))
    evogp_amd.set_sr_division("ieee")
    out["short_identical_to_ieee"] = bool(np.array_equal(fits["ieee"], fits["short"], equal_nan=True))
    a, b = fits["ieee"], fits["fast"]
    cls = (np.isnan(a) != np.isnan(b)) | (np.isposinf(a) != np.isposinf(b))
    fin = np.isfinite(a) & np.isfinite(b)
    rel = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(a[fin]), 1e-300)
    out.update(trees=int(a.size), finite=int(fin.sum()), class_mismatch=int(cls.sum()), identical=int((a[fin] == b[fin]).sum()),
               rel_max=float(rel.max()), rel_p999=float(np.quantile(rel, 0.999)), rel_median=float(np.median(rel)),
               over_1e5=int((rel > 1e-5).sum()), over_1e6=int((rel > 1e-6).sum()),
               mean_len=float(forest.batch_subtree_size[:, 0].float().mean()))
    print(tag, json.dumps(out), flush=True)


compare(forest, "random")
algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.01))
for gen in range(30):
    f = -algo.forest.SR_fitness(X, y, True, "auto")
    f[torch.isnan(f)] = -torch.inf
    algo.step(f)
compare(algo.forest, "evolved30")
