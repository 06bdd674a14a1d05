"""tree_SR_fitness on the shards of the headline population that N = 8, 4, 2, 1 ranks get (125 k ... 1 M trees), measured on ONE
GPU: per-call wall time, the stage split (program compilers | interpreter | follow-ups) and the ideal (the 1 M time / N)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from evogp_amd.tree import set_default_device
set_default_device(dev)
res = {}
for n in (1_000_000, 500_000, 250_000, 125_000):
    forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
    for _ in range(5): forest.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): forest.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
    _lib.lib.evogp_hip_debug_profile(1)
    for _ in range(10): forest.SR_fitness(Xd, yd, True, "auto")
    st = (ctypes.c_float * 3)(); nc = ctypes.c_int(0)
    _lib.lib.evogp_hip_debug_profile_read(st, ctypes.byref(nc)); _lib.lib.evogp_hip_debug_profile(0)
    res[n] = (ms, list(st))
    del forest
base = res[1_000_000][0]
for n, (ms, st) in res.items():
    print(f"{n:>8} trees: {ms:.4f} ms  (ideal {base * n / 1e6:.4f}, efficiency {base * n / 1e6 / ms:.1%})  compilers {st[0]*1e3:.0f} us | interpreter {st[1]*1e3:.0f} us | follow-ups {st[2]*1e3:.0f} us")
