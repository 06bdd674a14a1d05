#!/usr/bin/env python3
"""Tuning aid for the small-dataset (one row per lane) interpreter: `make` evolves the notebook configuration and saves the
forests of generations 5 and 30 to /tmp; `time` loads them and prints the stage split of a fitness call (run it under
different EVOGP_TC_* settings: they are read once per process)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

L = _lib.lib
dev = torch.device("cuda", 0); set_default_device(dev)
POP = 100_000
X = torch.tensor([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], device=dev)
y = (X.sum(1) % 2)[:, None].contiguous()
if sys.argv[1] == "make":
    desc = GenerateDescriptor(max_tree_len=128, input_len=3, output_len=1, using_funcs=["+", "-", "log", "sqrt", "pow", "/", "inv"],
                              max_layer_cnt=2, const_samples=[-1, 0, 1])
    algo = GeneticProgramming(Forest.random_generate(POP, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)),
                              DefaultCrossover(), DefaultMutation(0.2, desc), DefaultSelection(0.3, elite_rate=0.01))
    neg = torch.full((POP,), float("-inf"), dtype=torch.float32, device=dev)
    for g in range(31):
        if g in (5, 30):
            torch.save([t.cpu() for t in algo.forest._tensors()], f"/tmp/vis_forest_{g}.pt")
        f = -algo.forest.SR_fitness(X, y, True, "auto")
        algo.step(torch.where(torch.isnan(f), neg, f))
    sys.exit(0)
out = []
for g in (5, 30):
    v, t, s = (a.to(dev) for a in torch.load(f"/tmp/vis_forest_{g}.pt"))
    fo = Forest(3, 1, v, t, s)
    for _ in range(3): fo.SR_fitness(X, y, True, "auto")
    L.evogp_hip_debug_profile(1)
    for _ in range(10): fo.SR_fitness(X, y, True, "auto")
    st = (ctypes.c_float * 3)(); n = ctypes.c_int(0)
    L.evogp_hip_debug_profile_read(st, ctypes.byref(n)); L.evogp_hip_debug_profile(0)
    if len(sys.argv) > 2 and sys.argv[2] == "hist":
        import json
        tab = json.load(open(os.path.join(ROOT, "evogp_amd", "lib", "tc_handlers.json")))["K1_short"]["handlers"]
        nh = L.evogp_hip_debug_tc_nhandlers()
        hist = torch.zeros(2 * nh, dtype=torch.int64, device=dev)
        L.evogp_hip_debug_tc_histogram(POP, ctypes.c_void_p(hist.data_ptr()), 2 * nh, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        h = hist.cpu().numpy(); h = h[:nh] + h[nh:]
        rows = sorted(((int(h[v["id"]]), k, v) for k, v in tab.items()), reverse=True)
        tot = {c: sum(n_ * v[c] for n_, _, v in rows) / POP for c in ("valu", "salu", "lds", "smem")}
        print(f"gen {g}: words per tree {h.sum() / POP:.1f}; per tree (stub paths only, bodies behind branches not counted): {tot}")
        print("   ", ", ".join(f"{k} {n_ / POP:.2f}" for n_, k, v in rows[:16]))
    out.append(f"gen {g} (mean {float(s[:, 0].float().mean()):.1f}): compilers {st[0]*1e3:.0f} interpreter {st[1]*1e3:.0f} follow-ups {st[2]*1e3:.0f} us")
print(" | ".join(out))
