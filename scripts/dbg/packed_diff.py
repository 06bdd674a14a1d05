"""Which trees does the packed program compiler (EVOGP_TC_PACKED=32, default) treat differently from the one-tree-per-pass compiler
(EVOGP_TC_PACKED=0)?  Fitness words of the headline forest under both (two processes: the switch is read once), the differing trees
with their nodes, and the 32-tree batch around the first one evaluated alone with EVOGP_DEBUG_MARKS.

    python scripts/dbg/packed_diff.py            (parent: runs both children, compares)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def child(tag, lo, n):
    import torch

    import bench

    dev = torch.device("cuda", 0)
    forest, Xd, yd, X, y = bench.sr_inputs(0, 1_000_000, dev)
    if n:
        forest = forest[lo:lo + n]
    for rep in range(3):
        f = forest.SR_fitness(Xd, yd, True, "auto")
        torch.cuda.synchronize()
        np.save(os.path.join(OUT, f"pd_{tag}_{rep}.npy"), f.cpu().numpy().view(np.uint32))
    if not n:
        np.savez(os.path.join(OUT, "pd_forest.npz"), ty=forest.batch_node_type.cpu().numpy(), va=forest.batch_node_value.cpu().numpy(),
                 sz=forest.batch_subtree_size.cpu().numpy())


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    os.makedirs(OUT, exist_ok=True)
    for tag, env in (("one", {"EVOGP_TC_PACKED": "0"}), ("pk", {})):
        subprocess.run([sys.executable, __file__, tag, "0", "0"], env={**os.environ, **env}, check=True)
    a = [np.load(os.path.join(OUT, f"pd_one_{r}.npy")) for r in range(3)]
    b = [np.load(os.path.join(OUT, f"pd_pk_{r}.npy")) for r in range(3)]
    print("one: repeats equal", all(np.array_equal(a[0], x) for x in a), " packed: repeats equal", all(np.array_equal(b[0], x) for x in b))
    for r in range(3):
        d = np.nonzero(a[0] != b[r])[0]
        print(f"packed run {r}: {len(d)} words differ", d[:10], a[0][d[:10]], b[r][d[:10]])
    d = np.nonzero(a[0] != b[0])[0]
    if not len(d):
        d = np.nonzero(a[0] != b[1])[0]
    if len(d):
        fz = np.load(os.path.join(OUT, "pd_forest.npz"))
        t = int(d[0])
        n = int(fz["sz"][t, 0])
        print("tree", t, "len", n, "batch position", t % 32)
        print(" ".join(f"{int(fz['ty'][t, i])}/{fz['va'][t, i]:g}/{int(fz['sz'][t, i])}" for i in range(n)))
        lo = t - t % 32
        print("lengths of its batch:", fz["sz"][lo:lo + 32, 0].tolist())
        for tag, env in (("one32", {"EVOGP_TC_PACKED": "0"}), ("pk32", {})):
            subprocess.run([sys.executable, __file__, tag, str(lo), "32"], env={**os.environ, **env, "EVOGP_DEBUG_MARKS": "1"}, check=True)
        x, yv = np.load(os.path.join(OUT, "pd_one32_0.npy")), np.load(os.path.join(OUT, "pd_pk32_0.npy"))
        print("batch alone: one", x[t - lo], "packed", yv[t - lo], "full-run one", a[0][t], "full-run packed", b[0][t], "differ in batch:", np.nonzero(x != yv)[0])
    for f in os.listdir(OUT):
        if f.startswith("pd_"):
            os.remove(os.path.join(OUT, f))


if __name__ == "__main__":
    main()
