#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with the REFERENCE's own device code compiled for
the host (oracle/_ref/libevogp_ref.so, built by oracle/build_ref.py from /root/reference).

Run in the dev container (the reference does not exist on the GPU box):
    python tests/golden/make_golden.py
Outputs (committed, small):
    fixtures.json       hand-checkable vectors: SURVEY.md Appendix B (test/test_bind_success.py,
                        test/fix_bug.py argument sets) + hash / taus88 known answers
    battery_*.npz       seeded random batteries: generate / crossover / mutate / evaluate / SR fitness
                        for several function sets, single- and multi-output
Only the LIVE prefix of each tree row is meaningful in reference outputs (the tail is uninitialised
memory in the reference; the harness pre-zeroes it).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import build_ref  # noqa: E402
from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform  # noqa: E402

build_ref.build()
R = Oracle("reference")
ARITH = roulette_uniform([1, 2, 3, 4])
CS = [-1.0, 0.0, 1.0]


def tolist(a):
    a = np.asarray(a)
    if a.dtype == np.float32:
        return {"f32_bits": a.view(np.uint32).tolist()}
    return a.tolist()


fx = {}
fx["hash"] = [{"n": n, "k1": k1, "k2": k2, "out": R.hash(n, k1, k2)} for n, k1, k2 in [(0, 42, 0), (1, 42, 0), (123456, 7, 9), (2**32 - 1, 2**32 - 1, 1)]]
u, f = R.taus88(12345, 8)
fx["taus88_12345"] = {"u32": u.tolist(), "uniform": tolist(f)}
u, _ = R.taus88(341, 10000)
fx["taus88_default_10000th"] = int(u[9999])  # Thrust's documented known answer: 3535848941

# B1: test/test_bind_success.py generate / evaluate / crossover fixture
v, t, s = R.generate(2, 64, 2, 1, 0.3, 0.5, [42, 0], [0.1, 0.1] + [1.0] * 8, ARITH, CS)
ev = R.evaluate(v, t, s, [[1, 2], [3, 4]], 1)
cv, ct, cs_ = R.crossover(v, t, s, [0], [1], [2], [4])
fx["B1"] = {"value": tolist(v[:, :8]), "type": t[:, :8].tolist(), "size": s[:, :8].tolist(), "evaluate": tolist(ev),
            "crossover": {"value": tolist(cv[:, :10]), "type": ct[:, :10].tolist(), "size": cs_[:, :10].tolist()}}
# B2: SR fixture
v, t, s = R.generate(2, 64, 1, 1, 0.3, 0.5, [42, 0], [0.1, 0.1] + [1.0] * 8, ARITH, CS)
fx["B2"] = {"value": tolist(v[:, :8]), "type": t[:, :8].tolist(), "size": s[:, :8].tolist(),
            "sr_fitness": tolist(R.sr_fitness(v, t, s, [[1], [2]], [[1], [3]], True))}
# B3: multi-output
v, t, s = R.generate(2, 32, 3, 2, 0.5, 0.5, [7, 9], [0.2, 0.2] + [1.0] * 8, ARITH, CS)
ev = R.evaluate(v, t, s, [[1, 2, 3], [0.5, -1.5, 2]], 2)
mv, mt, ms = R.mutate(v, t, s, [1, 0], v[::-1].copy(), t[::-1].copy(), s[::-1].copy())
fx["B3"] = {"value": tolist(v[:, :8]), "type": t[:, :8].tolist(), "size": s[:, :8].tolist(), "evaluate": tolist(ev),
            "mutate": {"value": tolist(mv[:, :12]), "type": mt[:, :12].tolist(), "size": ms[:, :12].tolist()}}
# B4: test/fix_bug.py — (x0-x2)*(x0-x2) on four XOR rows: MSE 0.5
v = np.array([[3, 2, 0, 2, 2, 0, 2, 0]], np.float32)
t = np.array([[3, 3, 0, 0, 3, 0, 0, 0]], np.int16)
s = np.array([[7, 3, 1, 1, 3, 1, 1, 0]], np.int16)
fx["B4"] = {"sr_fitness": tolist(R.sr_fitness(v, t, s, [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1]], [[0], [1], [1], [0]], True))}
# B5: statistics of the C1 forest
v, t, s = R.generate(5000, 32, 3, 1, 0.5, 0.5, [42, 0], depth2leaf(4), ARITH, CS)
XOR_X = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.float32)
XOR_Y = (XOR_X.sum(1) % 2).astype(np.float32)[:, None]
fit = R.sr_fitness(v, t, s, XOR_X, XOR_Y, True)
fx["B5"] = {"mean_len": float(s[:, 0].mean()), "nan_count": int(np.isnan(fit).sum()),
            "len_checksum": int(s[:, 0].astype(np.int64).sum())}
with open(os.path.join(HERE, "fixtures.json"), "w") as fjs:
    json.dump(fx, fjs, indent=1)

# ---- seeded batteries ---------------------------------------------------------------------------
CONFIGS = {
    # name: (funcs, out_len, var_len, gp_len, max_layer_cnt, keys)
    "arith_so": ([1, 2, 3, 4], 1, 10, 64, 6, [42, 0]),
    "paper7_so": ([1, 2, 3, 4, 14, 15, 16], 1, 5, 64, 6, [11, 22]),
    "allfuncs_so": (list(range(29)), 1, 4, 128, 5, [5, 6]),
    "arith_mo": ([1, 2, 3, 4], 3, 6, 64, 6, [9, 1]),
    "allfuncs_mo": (list(range(29)), 4, 3, 128, 5, [77, 3]),
}
rng = np.random.default_rng(7)
for name, (funcs, out_len, var_len, L, mlc, keys) in CONFIGS.items():
    pop = 160
    rou = roulette_uniform(funcs)
    d2l = depth2leaf(mlc)
    cs = np.array([-1.0, 0.0, 1.0, 0.5, 2.0], np.float32)
    v, t, s = R.generate(pop, L, var_len, out_len, 0.5, 0.5, keys, d2l, rou, cs)
    sizes = s[:, 0].astype(np.int64)
    n = 256
    li = rng.integers(0, pop, n).astype(np.int32)
    ri = rng.integers(0, pop, n).astype(np.int32)
    ri[:4] = [-1, pop, pop + 3, -9]
    ln = (rng.integers(0, 2**31 - 1, n) % sizes[li]).astype(np.int32)
    rn = (rng.integers(0, 2**31 - 1, n) % sizes[np.clip(ri, 0, pop - 1)]).astype(np.int32)
    cv, ct, cs_ = R.crossover(v, t, s, li, ri, ln, rn)
    nv, nt, ns = R.generate(pop, L, var_len, out_len, 0.5, 0.5, keys[::-1], depth2leaf(3), rou, cs)
    mi = (rng.integers(0, 1024, pop) % sizes).astype(np.int32)
    mi[:3] = [-1, 5000, 0]
    mv, mt, ms = R.mutate(v, t, s, mi, nv, nt, ns)
    Xp = rng.uniform(-3, 3, (pop, var_len)).astype(np.float32)
    ev = R.evaluate(v, t, s, Xp, out_len)
    D = 100
    X = rng.uniform(-3, 3, (D, var_len)).astype(np.float32)
    y = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
    f_mse = R.sr_fitness(v, t, s, X, y, True)
    f_mae = R.sr_fitness(v, t, s, X, y, False)
    np.savez_compressed(
        os.path.join(HERE, f"battery_{name}.npz"),
        funcs=np.array(funcs), out_len=out_len, var_len=var_len, gp_len=L, max_layer_cnt=mlc, keys=np.array(keys, np.uint32),
        roulette=rou, depth2leaf=d2l, consts=cs, depth2leaf_new=depth2leaf(3),
        value=v, type=t, size=s,
        left_idx=li, right_idx=ri, left_node=ln, right_node=rn, cross_value=cv, cross_type=ct, cross_size=cs_,
        new_value=nv, new_type=nt, new_size=ns, mut_idx=mi, mut_value=mv, mut_type=mt, mut_size=ms,
        eval_x=Xp, eval_out=ev, sr_x=X, sr_y=y, sr_mse=f_mse, sr_mae=f_mae,
    )
    print(name, "mean len", sizes.mean(), "nan mse", int(np.isnan(f_mse).sum()))
print("golden vectors written to", HERE)
