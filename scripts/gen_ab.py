"""tree_generate per launch at 100 k and 1 M rows of 64 nodes (and two other shapes), the masked donor launch of a generation, and a hash
of the rows -- for A/B runs of two builds of csrc/generate.hip (round 6: the persistent-lane and straight-line kernels against the
staged one, profiles/r06_generate_depth_sweep.log): the hashes must agree."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_capi as g
from bench_ops_common import depth2leaf, roulette_uniform, timed

L_ = g.L
S = g._stream
keys = g.dev([42, 0], np.uint32); d2l = g.dev(depth2leaf(6), np.float32); rou = g.dev(roulette_uniform([1, 2, 3, 4]), np.float32); cs = g.dev([-1, 0, 1], np.float32)
d2l3 = g.dev(depth2leaf(3), np.float32)


def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()[:12]


for pop, L, out_len in ((100_000, 64, 1), (1_000_000, 64, 1), (200_000, 32, 1), (100_000, 64, 3)):
    v = torch.empty((pop, L), dtype=torch.float32, device=g.DEV); t = torch.empty((pop, L), dtype=torch.int16, device=g.DEV); s = torch.empty((pop, L), dtype=torch.int16, device=g.DEV)
    def gen():
        assert L_.evogp_hip_generate(pop, L, 10, out_len, 3, 0.5, 0.5, keys.data_ptr(), d2l.data_ptr(), rou.data_ptr(), cs.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), 0, S()) == 0
    us = timed(gen)
    print(f"tree_generate pop {pop:>8} L {L} out {out_len}: {us:8.1f} us   mean len {float(s[:, 0].float().mean()):.2f}  rows {digest(v, t, s)}")
    # the donors of a generation: one row in five, small trees (max_layer_cnt 3), hashed mask
    n_new = pop * 99 // 100
    below = int(0.2 * (2**31 - 1))
    def donors():
        assert L_.evogp_hip_generate_masked_hashed(n_new, L, 10, out_len, 3, 0.5, 0.5, d2l3.data_ptr(), rou.data_ptr(), cs.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), 0, 1234, 7, below, S()) == 0
    v.zero_(); t.zero_(); s.zero_()
    us = timed(donors)
    print(f"  generate_masked_hashed {n_new:>8} rows, 20 % live, max_layer_cnt 3: {us:8.1f} us   rows {digest(v, t, s)}")
