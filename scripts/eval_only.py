#!/usr/bin/env python3
"""tree_evaluate at the policy shape (C5) alone, for kernel traces: pop 50k, L 256, 17 inputs, 6 outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_capi as g
from helpers import depth2leaf, roulette_uniform
L_ = g.L; S = g._stream
layers = int(os.environ.get("EVAL_LAYERS", "6"))
pe, Le = 50_000, 256
keys = g.dev([42, 0], np.uint32); d2l = g.dev(depth2leaf(layers), np.float32); rou = g.dev(roulette_uniform([1, 2, 3, 4]), np.float32)
cse = g.dev(np.linspace(-1, 1, 100), np.float32)
ev = torch.empty((pe, Le), dtype=torch.float32, device=g.DEV); et = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV); es = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV)
assert L_.evogp_hip_generate(pe, Le, 17, 6, 100, 0.5, 0.5, keys.data_ptr(), d2l.data_ptr(), rou.data_ptr(), cse.data_ptr(), ev.data_ptr(), et.data_ptr(), es.data_ptr(), 0, S()) == 0
obs = torch.randn(pe, 17, device=g.DEV); res = torch.empty(pe, 6, device=g.DEV)
sz = es[:, 0].float()
print("len mean %.1f max %d" % (float(sz.mean()), int(sz.max())))
for _ in range(30):
    assert L_.evogp_hip_evaluate(pe, Le, 17, 6, ev.data_ptr(), et.data_ptr(), es.data_ptr(), obs.data_ptr(), res.data_ptr(), S()) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    assert L_.evogp_hip_evaluate(pe, Le, 17, 6, ev.data_ptr(), et.data_ptr(), es.data_ptr(), obs.data_ptr(), res.data_ptr(), S()) == 0
e1.record(); torch.cuda.synchronize()
print("us per call %.1f" % (e0.elapsed_time(e1) / 50 * 1e3))
