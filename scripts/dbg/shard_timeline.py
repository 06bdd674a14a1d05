"""tree_SR_fitness on the first N trees of the headline population, in a loop (for a rocprofv3 --kernel-trace timeline of ONE shard's
call: which launches it makes, how long each lasts, the gaps between them).  python scripts/dbg/shard_timeline.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
dev = torch.device("cuda", 0)
forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
for _ in range(10):
    forest.SR_fitness(Xd, yd, True, "auto")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    forest.SR_fitness(Xd, yd, True, "auto")
torch.cuda.synchronize()
print(f"{n} trees: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per call")
