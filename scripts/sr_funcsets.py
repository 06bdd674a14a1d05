#!/usr/bin/env python3
"""tree_SR_fitness on configs[1]-sized forests of other function sets and output counts (the paths outside the threaded
code): launch time and tree-evals/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

import ctypes, json
from evogp_amd import _lib
L = _lib.lib
_t = json.load(open(os.path.join(ROOT, "evogp_amd", "lib", "tc_handlers.json")))["K8_short"]["handlers"]
names = [None] * (max(v["id"] for v in _t.values()) + 1)
for k, v in _t.items(): names[v["id"]] = k
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
                             ("+ - * / max min < > if (generic stubs)", ["+", "-", "*", "/", "max", "min", "<", ">", "if"], 1),
                             ("+ - * /, 4 outputs", ["+", "-", "*", "/"], 4), ("+ - * /, 6 outputs", ["+", "-", "*", "/"], 6),
                             ("+ - * /, 10 outputs (8 rows, wide stack)", ["+", "-", "*", "/"], 10),
                             ("+ - * / sin cos tan, 4 outputs", ["+", "-", "*", "/", "sin", "cos", "tan"], 4),
                             ("vis.ipynb set: + - log sqrt pow / inv", ["+", "-", "log", "sqrt", "pow", "/", "inv"], 1)):
    mlc = 4 if "if" in funcs else 6   # a full tree must fit the 64-node row (descriptor.py:19-31)
    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=out_len, using_funcs=funcs, max_layer_cnt=mlc, const_samples=[-1, 0, 1])
    forest = Forest.random_generate(100_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
    y = torch.randn(1024, out_len, device=dev)
    ms = timed(lambda: forest.SR_fitness(X, y, True, "auto"))
    # where the time goes: per-stage events inside the call, handler histogram of the compiled programs
    L.evogp_hip_debug_profile(1)
    for _ in range(5): forest.SR_fitness(X, y, True, "auto")
    st = (ctypes.c_float * 3)(); n = ctypes.c_int(0)
    L.evogp_hip_debug_profile_read(st, ctypes.byref(n)); L.evogp_hip_debug_profile(0)
    nh = L.evogp_hip_debug_tc_nhandlers()
    hist = torch.zeros(2 * nh, dtype=torch.int64, device=dev)
    L.evogp_hip_debug_tc_histogram(100_000, ctypes.c_void_p(hist.data_ptr()), 2 * nh, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    h = hist.cpu().numpy(); h = h[:nh] + h[nh:]
    words = int(h.sum()); skip = int(h[names.index("skip")])
    print(f"| {name} | {float(forest.batch_subtree_size[:, 0].float().mean()):.1f} | {ms:.3f} | {100_000 * 1024 / ms / 1e6:.0f} G | compilers {st[0]*1e3:.0f} us, interpreter {st[1]*1e3:.0f} us, "
          f"register kernels {st[2]*1e3:.0f} us | {words / 1e5:.1f} words per tree, {skip / 1e3:.1f} % of the trees left to the register kernels |", flush=True)
