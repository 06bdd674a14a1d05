#!/usr/bin/env python3
"""BASELINE configs[4] shape (policy trees: pop 50 k, 17 observations, 6 actions, max_tree_len 256, 1000 steps per generation,
example/brax_task.py:19-32): microseconds per environment step of the forward pass — the stack interpreter
(evogp_hip_evaluate), the prepared operation lists (evogp_hip_evaluate_prepared), and a whole graph-replayed rollout step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.problem import RolloutProblem
from evogp_amd.problem.rollout import LinearTrackingEnv
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); set_default_device(dev)
pop = 50_000
desc = GenerateDescriptor(max_tree_len=256, input_len=17, output_len=6, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6,
                          const_samples=torch.linspace(-1, 1, 100).tolist())
forest = Forest.random_generate(pop, desc, keys=torch.tensor([7, 0], dtype=torch.uint32, device=dev))
obs = torch.randn(pop, 17, generator=torch.Generator().manual_seed(7)).to(dev)
sizes = forest.batch_subtree_size[:, 0].float()
print(f"pop {pop}, in 17, out 6, L 256: mean tree length {float(sizes.mean()):.1f}, longest {int(sizes.max())}")


def timed(f, reps=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


args = (pop, 256, 17, 6, *forest._tensors())
us_stack = timed(lambda: torch.ops.evogp_cuda.tree_evaluate(*args, obs))
ws, with_fallback = forest.prepare_forward()
us_prep = timed(lambda: torch.ops.evogp_hip.tree_evaluate_prepared(*args, ws, with_fallback, obs))
t0 = time.perf_counter(); torch.ops.evogp_hip.tree_evaluate_prepare(*args); torch.cuda.synchronize(); prep_ms = (time.perf_counter() - t0) * 1e3
recs = ws[-(pop * 4 + 64):-64].view(torch.int32).float()
nodes = float(sizes.sum())
alg = 16 * float(recs.clamp(min=0).sum()) + 4 * pop + 4 * pop * (17 + 6)
print(f"| forward pass | us per step | note |\n|---|---|---|")
print(f"| stack interpreter (evogp_hip_evaluate) | {us_stack:.1f} | reads 8 B x {nodes / pop:.1f} nodes per tree, decodes, interprets |")
print(f"| operation lists (evogp_hip_evaluate_prepared) | {us_prep:.1f} | {float(recs.clamp(min=0).mean()):.1f} records of 16 B per tree, {alg / 1e6:.1f} MB per step = {alg / us_prep / 1e6:.2f} TB/s; "
      f"{int((recs == -2).sum())} trees left to the stack interpreter; built once per forest in {prep_ms:.2f} ms |")
for graph in (False, True):
    prob = RolloutProblem(LinearTrackingEnv(device=dev), 1000, use_graph=graph)
    prob.evaluate(forest); torch.cuda.synchronize()
    t0 = time.perf_counter(); prob.evaluate(forest); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"| whole rollout step ({'HIP graph replay' if graph else 'eager'}), 1000 steps | {dt * 1e3:.1f} | policy forward + tanh + the linear environment's kernels; {dt * 1e3:.0f} ms per generation |")
