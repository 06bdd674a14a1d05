"""time of evogp_hip::select_survivors (elite 1 %, keep 30 %) at a few population sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import evogp_amd  # noqa: F401
dev = torch.device("cuda", 0)
for n in (10_000, 100_000, 125_000, 1_000_000):
    x = torch.randn(n, device=dev)
    f = lambda: torch.ops.evogp_hip.select_survivors(x, n // 100, n * 3 // 10)
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): f()
    torch.cuda.synchronize(); print(n, round((time.perf_counter() - t0) / 300 * 1e6, 1), "us")
