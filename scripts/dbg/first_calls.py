"""Duration of the first N tree_SR_fitness calls of a fresh process on the headline population (one HIP event pair per call):
how long the device takes to reach its steady state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import evogp_amd  # noqa: F401
import bench

dev = torch.device("cuda", 0)
forest, Xd, yd, _, _ = bench.sr_inputs(0, 1_000_000, dev)
N = 400
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for a, b in ev:
    a.record(); forest.SR_fitness(Xd, yd, True, "auto"); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
for lo in (0, 1, 2, 5, 10, 20, 40, 80, 160, 320):
    hi = min(N, max(lo + 1, lo * 2))
    seg = t[lo:hi]
    print(f"calls {lo:3d}..{hi - 1:3d}: mean {sum(seg) / len(seg):.4f} ms  min {min(seg):.4f}  max {max(seg):.4f}")
