"""debug: the call-scratch counters around replays of a captured fitness call"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_capi as g
import evogp_amd
from helpers import ARITH, depth2leaf, roulette_uniform
L = ctypes.CDLL(os.path.join(ROOT, "evogp_amd/lib/libevogp_hip.so"))
rng = np.random.default_rng(1)
pop = int(os.environ.get("POP", "3000"))
v, t, s = g.generate(pop, 64, 4, 1, 0.5, 0.3, [15, 16], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.0, 1.0])
X = rng.standard_normal((1024, 4)).astype(np.float32); y = rng.standard_normal((1024, 1)).astype(np.float32)
a = [g.dev(v, np.float32), g.dev(t, np.int16), g.dev(s, np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]
def call(out, stream):
    rc = g.L.evogp_hip_sr_fitness(pop, 1024, 64, 4, 1, 1, *[x.data_ptr() for x in a], out.data_ptr(), 0, stream.cuda_stream)
    assert rc == 0
def scratch(stream, tag):
    w = (ctypes.c_uint * (2 * 2304))(); cur = ctypes.c_int(0)
    rc = L.evogp_hip_debug_call_scratch(ctypes.c_void_p(stream.cuda_stream), w, ctypes.byref(cur))
    w = np.array(w[:]).reshape(2, 2304)
    print(f"  {tag}: rc {rc} current block {cur.value}; block0 counters {[int(w[0, 32 * (1 + x)]) for x in range(8)]} flags {w[0, :5].tolist()}; block1 counters {[int(w[1, 32 * (1 + x)]) for x in range(8)]} flags {w[1, :5].tolist()}")
main = torch.cuda.current_stream()
out = torch.full((pop,), 777.0, device=g.DEV)
cap = torch.cuda.Stream(); cap.wait_stream(main)
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    call(out, cap); cap.synchronize()
    scratch(cap, "after the warm-up")
    with torch.cuda.graph(graph, stream=cap):
        call(out, torch.cuda.current_stream())
scratch(cap, "after the capture")
for i in range(3):
    out.fill_(555.0)
    graph.replay(); torch.cuda.synchronize()
    print(f"pop {pop} replay {i}: untouched {(out == 555.0).sum().item()}")
    scratch(cap, f"after replay {i}")
