"""tree_evaluate (multi-output, policy shape) at several row widths and populations: device time per call from a HIP-graph replay.
python scripts/dbg/eval_shape.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import gpu_capi as g
from helpers import depth2leaf, roulette_uniform

L_ = g.L
keys = g.dev([42, 0], np.uint32); rou = g.dev(roulette_uniform([1, 2, 3, 4]), np.float32)
d2l6 = g.dev(depth2leaf(6), np.float32); cse = g.dev(np.linspace(-1, 1, 100), np.float32)
for pe, Le in ((50_000, 256), (50_000, 64), (200_000, 64), (12_500, 256)):
    ev = torch.empty((pe, Le), dtype=torch.float32, device=g.DEV); et = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV); es = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV)
    assert L_.evogp_hip_generate(pe, Le, 17, 6, 100, 0.5, 0.5, keys.data_ptr(), d2l6.data_ptr(), rou.data_ptr(), cse.data_ptr(), ev.data_ptr(), et.data_ptr(), es.data_ptr(), 0, g._stream()) == 0
    obs = torch.randn(pe, 17, device=g.DEV); res = torch.empty(pe, 6, device=g.DEV)
    gs = torch.cuda.Stream()
    with torch.cuda.stream(gs):
        st = g._stream()
        def evaluate():
            assert L_.evogp_hip_evaluate(pe, Le, 17, 6, ev.data_ptr(), et.data_ptr(), es.data_ptr(), obs.data_ptr(), res.data_ptr(), st) == 0
        evaluate(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=gs):
            for _ in range(20): evaluate()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
    print(f"pop {pe} L {Le}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call, mean len {float(es[:, 0].float().mean()):.1f}")
