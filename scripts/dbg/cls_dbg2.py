import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform
import gpu_capi as g
o = Oracle("port"); rng = np.random.default_rng(1)
pop, L, var_len, out_len, D = 24, 16, 2, 3, 8
f = o.generate(pop, L, var_len, out_len, 0.5, 0.5, [3, 1], depth2leaf(3), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1, 0.5])
X = rng.uniform(0, 16, (D, var_len)).astype(np.float32)
labels = rng.integers(0, out_len, D).astype(np.int32)
got = g.batch_argmax_count(*f, X, labels, out_len)
outs = o.batch_evaluate(*f, X, out_len)
pred = torch.argmax(torch.clip(torch.softmax(torch.from_numpy(outs), dim=2), 1e-15, 1 - 1e-15), dim=2).numpy()
want = (pred == labels[None, :]).sum(1)
print("labels", labels)
for t in range(pop):
    n = f[2][t, 0]
    print(t, "len", n, "types", f[1][t, :n].tolist(), "vals", np.round(f[0][t, :n], 2).tolist() if n < 8 else "...", "got", got[t], "want", want[t], "pred", pred[t].tolist())
