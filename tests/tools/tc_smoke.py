#!/usr/bin/env python3
"""First contact with the threaded-code SR-fitness path (sr_tc.hip): small populations first, then ragged
datasets, MAE, both row widths, deep and malformed trees; everything checked against the CPU oracle.
Run under `timeout`: a wrong jump target would hang the wave."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

os.environ["EVOGP_SR_ASM"] = "3"
if len(sys.argv) > 1:
    os.environ["EVOGP_TC_K"] = sys.argv[1]
import gpu_capi as g  # noqa: E402
from helpers import c2_dataset, depth2leaf, roulette_uniform  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

o = Oracle("port")


def check(f, X, y, mse, what):
    want = o.sr_fitness(*f, X, y, use_mse=mse)
    got = g.sr_fitness(*f, X, y, use_mse=mse)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (what, "nan sets differ", int(np.isnan(got).sum()), int(np.isnan(want).sum()), np.argwhere(np.isnan(got) != np.isnan(want))[:5].ravel())
    assert np.array_equal(np.isposinf(got), np.isposinf(want)), (what, "inf sets differ")
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-5, atol=0, err_msg=what)
    print(f"tc {os.environ.get('EVOGP_TC_K', 'auto')}: {what} ok ({int(ok.sum())} finite of {len(want)})", flush=True)


X, y = c2_dataset()
for pop, mlc in ((1, 1), (1, 2), (4, 2), (64, 3), (1000, 6), (20000, 6)):
    f = o.generate(pop, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(mlc), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    check(f, X, y, True, f"pop {pop} layers {mlc} D 1024")
f = o.generate(3000, 64, 10, 1, 0.5, 0.5, [7, 1], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1, 0.5, 2.0])
for D in (1, 7, 64, 255, 256, 257, 511, 513, 1000, 1500, 2048):
    Xd, yd = c2_dataset(D=D)
    check(f, Xd, yd, True, f"pop 3000 D {D} mse")
    check(f, Xd, yd, False, f"pop 3000 D {D} mae")
for vl in (1, 3, 17, 32):
    Xd, yd = c2_dataset(D=700, var_len=max(vl, 6))
    Xd = np.ascontiguousarray(Xd[:, :vl])
    fv = o.generate(2000, 64, vl, 1, 0.5, 0.3, [9, 9], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    check(fv, Xd, yd, True, f"var_len {vl} D 700")
# all 63-node trees (two program blocks), the full function set (marked trees take the register kernels), short rows
ff = o.generate(2000, 64, 10, 1, 0.5, 0.5, [3, 3], [0.0] * 5 + [1.0] * 5, roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
check(ff, X, y, True, "full 63-node trees")
fa = o.generate(2000, 64, 10, 1, 0.5, 0.5, [5, 5], depth2leaf(5), roulette_uniform(list(range(29))), [-1, 0, 1])
got = g.sr_fitness(*fa, X, y)
want = o.sr_fitness(*fa, X, y)
assert np.array_equal(np.isnan(got), np.isnan(want)) or (np.isnan(got) != np.isnan(want)).mean() < 0.02, "all-function forest: NaN sets"
print("all-function forest ok (marked trees fall through)")
fs = o.generate(5000, 32, 3, 1, 0.5, 0.5, [42, 0], depth2leaf(4), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
Xx = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.float32)
yx = (Xx.sum(1) % 2).astype(np.float32)[:, None]
check(fs, Xx, yx, True, "configs[0] XOR-3d pop 5000 L 32")
print("TC_SMOKE_OK")
