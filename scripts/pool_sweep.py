"""tree_SR_fitness at 100 k / 125 k / 1 M trees: per-call time and the stage split, for the share of the population that the workgroups'
pools hand out (EVOGP_TC_STATIC, read once per process: one process per value, see scripts/gpu_session.sh extra:)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from evogp_amd.tree import set_default_device
set_default_device(dev)
tag = f"STATIC={os.environ.get('EVOGP_TC_STATIC', '-')} DYNSHIFT={os.environ.get('EVOGP_TC_DYNSHIFT', '-')}"
ref = {}
for n in (100_000, 125_000, 250_000, 1_000_000):
    forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
    for _ in range(30): forest.SR_fitness(Xd, yd, True, "auto")
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): forest.SR_fitness(Xd, yd, True, "auto")
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
    _lib.lib.evogp_hip_debug_profile(1)
    for _ in range(10): forest.SR_fitness(Xd, yd, True, "auto")
    st = (ctypes.c_float * 3)(); nc = ctypes.c_int(0)
    _lib.lib.evogp_hip_debug_profile_read(st, ctypes.byref(nc)); _lib.lib.evogp_hip_debug_profile(0)
    f = forest.SR_fitness(Xd, yd, True, "auto")
    h = int(torch.where(torch.isnan(f), torch.zeros_like(f), f).view(torch.int32).to(torch.int64).sum())
    print(f"{tag} {n:>8} trees: {best:.4f} ms  compilers {st[0]*1e3:.0f} us | interpreter {st[1]*1e3:.0f} us | follow-ups {st[2]*1e3:.0f} us   words checksum {h}")
    del forest
