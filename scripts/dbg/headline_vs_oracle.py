"""The headline population (1 M trees x 1024 rows) on the GPU against the plain-C oracle on the host's cores: NaN / inf sets and the largest
relative difference of the finite fitness values (the tolerance of tests/: 1e-5 on + - * /)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import evogp_amd  # noqa: F401
sys.argv = [sys.argv[0]]
import bench
from oracle.pyoracle import Oracle

dev = torch.device("cuda", 0)
n = int(os.environ.get("POP", "1000000"))
forest, Xd, yd, X, y = bench.sr_inputs(0, n, dev)
got = forest.SR_fitness(Xd, yd, True, "auto").cpu().numpy().astype(np.float64)
o = Oracle("port", native=True)
want = o.sr_fitness(forest.batch_node_value.cpu().numpy(), forest.batch_node_type.cpu().numpy(), forest.batch_subtree_size.cpu().numpy(), X, y, True, 0).astype(np.float64)
nan_eq = bool(np.array_equal(np.isnan(got), np.isnan(want))); inf_eq = bool(np.array_equal(np.isinf(got), np.isinf(want)))
fin = np.isfinite(got) & np.isfinite(want)
rel = np.abs(got[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-30)
print(f"{n} trees: NaN sets equal {nan_eq} ({int(np.isnan(want).sum())}), inf sets equal {inf_eq} ({int(np.isinf(want).sum())}), finite {int(fin.sum())}: max rel diff {rel.max():.3e}, "
      f"99.99th percentile {np.quantile(rel, 0.9999):.3e}, bit-equal {int((got[fin] == want[fin]).sum())}")
sys.exit(0 if nan_eq and inf_eq and rel.max() <= 1e-5 else 1)
