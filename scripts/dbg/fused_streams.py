"""tree_SR_fitness on two streams without synchronisation in between (EVOGP_TC_FUSED=1 for the one-kernel call): modes one / two / sync / other / race.
mode race leaves out the side stream's wait for the fill of its output on the main stream -- the 777s that then survive are the script's, not the engine's."""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gpu_capi as g
from helpers import ARITH, depth2leaf, roulette_uniform
from oracle.pyoracle import Oracle
CS3 = [-1.0, 0.0, 1.0]
rng = np.random.default_rng(0)
o = Oracle("port")
v, t, s = g.generate(3000, 64, 4, 1, 0.5, 0.3, [5, 6], depth2leaf(6), roulette_uniform(ARITH), CS3)
X = rng.standard_normal((1024, 4)).astype(np.float32); y = rng.standard_normal((1024, 1)).astype(np.float32)
want = o.sr_fitness(v, t, s, X, y)
a = [g.dev(v, np.float32), g.dev(t, np.int16), g.dev(s, np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]
def call(out, stream):
    rc = g.L.evogp_hip_sr_fitness(3000, 1024, 64, 4, 1, 1, *[x.data_ptr() for x in a], out.data_ptr(), 0, stream.cuda_stream); assert rc == 0
main = torch.cuda.current_stream(); side = torch.cuda.Stream(); torch.cuda.synchronize()
mode = sys.argv[1]
outs = []
for i in range(9):
    out = torch.full((3000,), 777.0, dtype=torch.float32, device=g.DEV)
    st = side if (mode != "one" and i % 3 == 2) else main
    if st is side and mode != "race": side.wait_stream(main)   # (the fill of `out` is main's work; mode "race": the side stream does not wait for it)
    call(out, st)
    if mode == "sync": torch.cuda.synchronize()
    if mode == "other" and i == 3:
        g.sr_fitness(v[:50], t[:50], s[:50], X[:100], np.tile(y[:100], (1, 1)), use_mse=False, kernel_type=2)
    outs.append(out)
torch.cuda.synchronize()
for i, out in enumerate(outs):
    got = out.cpu().numpy()
    print(mode, "call", i, "nan", int(np.isnan(got).sum()), "want nan", int(np.isnan(want).sum()), "777s", int((got == 777.0).sum()))
