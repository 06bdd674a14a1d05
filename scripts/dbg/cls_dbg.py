import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform
import gpu_capi as g
o = Oracle("port"); rng = np.random.default_rng(1)
for (pop, L, var_len, out_len, D) in ((300, 64, 4, 3, 8), (300, 64, 4, 3, 64), (300, 64, 4, 3, 200), (300, 64, 4, 3, 512), (300, 64, 4, 3, 700), (300, 64, 4, 10, 200), (300, 64, 64, 10, 300), (300, 64, 64, 10, 1797)):
    f = o.generate(pop, L, var_len, out_len, 0.5, 0.5, [3, 1], depth2leaf(5), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1, 0.5])
    X = rng.uniform(0, 16, (D, var_len)).astype(np.float32)
    labels = rng.integers(0, out_len, D).astype(np.int32)
    got = g.batch_argmax_count(*f, X, labels, out_len)
    outs = torch.from_numpy(o.batch_evaluate(*f, X, out_len))
    pred = torch.argmax(torch.clip(torch.softmax(outs, dim=2), 1e-15, 1 - 1e-15), dim=2)
    want = (pred == torch.from_numpy(labels.astype(np.int64))[None, :]).sum(1).numpy()
    bad = np.flatnonzero(np.abs(got - want) > 2)
    print((pop, L, var_len, out_len, D), "max diff", int(np.abs(got - want).max()), "bad trees", len(bad), "first", [(int(i), int(got[i]), int(want[i])) for i in bad[:5]])
