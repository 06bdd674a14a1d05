"""tree_SR_fitness on a shard of N trees of the headline population: the interpreter's time for one setting of the work-distribution
knobs (environment: EVOGP_TC_DYNSHIFT, EVOGP_TC_STATIC, EVOGP_TC_BATCH), one process per setting (the knobs are read once)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from evogp_amd.tree import set_default_device
set_default_device(dev)
for n in [int(a) for a in sys.argv[1:]] or [125_000]:
    forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
    for _ in range(5): forest.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): forest.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    _lib.lib.evogp_hip_debug_profile(1)
    for _ in range(10): forest.SR_fitness(Xd, yd, True, "auto")
    st = (ctypes.c_float * 3)(); nc = ctypes.c_int(0)
    _lib.lib.evogp_hip_debug_profile_read(st, ctypes.byref(nc)); _lib.lib.evogp_hip_debug_profile(0)
    knobs = " ".join(f"{k[9:]}={os.environ[k]}" for k in ("EVOGP_TC_DYNSHIFT", "EVOGP_TC_STATIC", "EVOGP_TC_BATCH") if k in os.environ) or "defaults"
    print(f"{n:>8} trees [{knobs}]: {ms:.4f} ms  compilers {st[0]*1e3:.0f} us | interpreter {st[1]*1e3:.1f} us | follow-ups {st[2]*1e3:.0f} us", flush=True)
