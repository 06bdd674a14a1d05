#!/usr/bin/env python3
"""One generation with the reference's OTHER operator sets, timed on one MI355X:

  * brax:    pop 50 k, L 256, 17 inputs, 6 outputs, DefaultCrossover + CombinedMutation[DefaultMutation(0.2), DeleteMutation(0.8)]
             (/root/reference/example/brax_task.py:38-45 at BASELINE configs[4]'s population)
  * hoist:   the same forest with CombinedMutation[HoistMutation(0.2), InsertMutation(0.2)] (README.md:233-251 variants)
  * point:   pop 100 k, L 64, single output, SinglePointMutation(0.2) (+ MultiConstMutation for the constants)
  * default: the fused default step on the same forests, for the ratio

Per case: milliseconds per generation (HIP events over `reps` generations, fitness handed in as a random vector -- selection,
crossover and mutation only) and kernel launches per generation (torch.profiler kernel count over one generation)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.algorithm import (CombinedMutation, DefaultCrossover, DefaultMutation, DefaultSelection, DeleteMutation, GeneticProgramming,
                                 HoistMutation, InsertMutation, MultiConstMutation, SinglePointMutation)
from evogp_amd.tree import Forest, GenerateDescriptor

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def launches_of(fn):
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name)


def run(name, forest, mutation, reps=10):
    algo = GeneticProgramming(forest, DefaultCrossover(), mutation, DefaultSelection(survival_rate=0.3, elite_rate=0.01))
    fit = torch.rand(forest.pop_size, device=dev)
    for _ in range(3):
        algo.step(fit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        algo.step(fit)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    try:
        n = launches_of(lambda: algo.step(fit))
    except Exception as ex:   # (the profiler is a convenience here, the timing is the measurement)
        n = f"n/a ({type(ex).__name__})"
    mean_len = float(algo.forest.batch_subtree_size[:, 0].float().mean())
    print(f"| {name} | {forest.pop_size} | {forest.max_tree_len} | {ms:.3f} | {n} | {mean_len:.1f} |", flush=True)
    return {"ms_per_generation": round(ms, 4), "launches": n, "mean_len_after": round(mean_len, 1)}


out = {}
print("| operator set | pop | L | ms per generation | kernel launches | mean length after |\n|---|---|---|---|---|---|")
d6 = GenerateDescriptor(max_tree_len=256, input_len=17, output_len=6, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_range=[-1, 1], sample_cnt=100)
mk6 = lambda: Forest.random_generate(50_000, d6, keys=torch.tensor([7, 1], dtype=torch.uint32, device=dev))
out["brax_default"] = run("default step (DefaultMutation 0.2)", mk6(), DefaultMutation(0.2, d6.update(max_layer_cnt=3)))
out["brax_combined_delete"] = run("brax_task.py: Combined[Default 0.2, Delete 0.8]", mk6(),
                                  CombinedMutation([DefaultMutation(0.2, d6.update(max_layer_cnt=3)), DeleteMutation(0.8)]))
out["hoist_insert"] = run("Combined[Hoist 0.2, Insert 0.2]", mk6(), CombinedMutation([HoistMutation(0.2), InsertMutation(0.2, d6.update(max_layer_cnt=3))]))
d1 = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
mk1 = lambda: Forest.random_generate(100_000, d1, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
out["sr_default"] = run("default step (DefaultMutation 0.2)", mk1(), DefaultMutation(0.2, d1.update(max_layer_cnt=3)))
out["single_point"] = run("SinglePointMutation 0.2", mk1(), SinglePointMutation(0.2, d1))
out["point_and_const"] = run("Combined[SinglePoint 0.2, MultiConst 0.2]", mk1(), CombinedMutation([SinglePointMutation(0.2, d1), MultiConstMutation(0.2, d1)]))
print("N3_JSON " + json.dumps(out))
