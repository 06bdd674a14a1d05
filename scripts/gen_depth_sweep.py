"""tree_generate at 100 k x 64 under descriptors of growing depth: how much of a launch is the serial loop (grows with the longest tree of a
wave) and how much is fixed (tables, row flushes, the launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_capi as g
from bench_ops_common import depth2leaf, roulette_uniform, timed

keys = g.dev([42, 0], np.uint32); rou = g.dev(roulette_uniform([1, 2, 3, 4]), np.float32); cs = g.dev([-1, 0, 1], np.float32)
for pop in (100_000, 1_000_000):
    L = 64
    v = torch.empty((pop, L), dtype=torch.float32, device=g.DEV); t = torch.empty((pop, L), dtype=torch.int16, device=g.DEV); s = torch.empty((pop, L), dtype=torch.int16, device=g.DEV)
    for depth in (1, 2, 3, 4, 5, 6):
        d2l = g.dev(depth2leaf(depth), np.float32)
        def gen():
            assert g.L.evogp_hip_generate(pop, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d2l.data_ptr(), rou.data_ptr(), cs.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), 0, g._stream()) == 0
        us = timed(gen)
        lens = s[:, 0].float()
        wmax = lens[: pop // 64 * 64].view(-1, 64).max(1).values.mean()
        print(f"pop {pop:>8} max_layer_cnt {depth}: {us:7.1f} us   mean len {float(lens.mean()):5.2f}  mean over waves of the longest tree {float(wmax):5.1f}")
