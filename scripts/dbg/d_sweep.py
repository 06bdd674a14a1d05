import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.tree import set_default_device
L = _lib.lib
dev = torch.device("cuda", 0); set_default_device(dev)
forest, X, y, _, _ = bench.sr_inputs(0, 1_000_000, dev)
for D in (256, 512, 1024, 1536, 2048, 3072):
    g = torch.Generator().manual_seed(1)
    Xd = (torch.rand(D, 10, generator=g) * 10 - 5).to(dev); yd = torch.randn(D, 1, generator=g).to(dev)
    for _ in range(3): forest.SR_fitness(Xd, yd, True, "auto")
    L.evogp_hip_debug_profile(1)
    for _ in range(10): forest.SR_fitness(Xd, yd, True, "auto")
    st = (ctypes.c_float * 3)(); n = ctypes.c_int(0)
    L.evogp_hip_debug_profile_read(st, ctypes.byref(n)); L.evogp_hip_debug_profile(0)
    print(f"D {D}: compilers {st[0]*1e3:.0f} us, interpreter {st[1]*1e3:.0f} us ({st[1]*1e6/D:.0f} ns per row-block of 1 M trees), follow-ups {st[2]*1e3:.0f} us")
