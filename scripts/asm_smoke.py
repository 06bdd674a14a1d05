#!/usr/bin/env python3
"""First contact with the threaded-code (assembly) interpreter core: tiny populations, both stack depths,
checked against the CPU oracle.  Run under `timeout`: a wrong jump target would hang the wave."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

depth = sys.argv[1] if len(sys.argv) > 1 else "10"
os.environ["EVOGP_SR_ASM"] = depth
import gpu_capi as g  # noqa: E402
from helpers import c2_dataset, depth2leaf, roulette_uniform  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

o = Oracle("port")
X, y = c2_dataset()
for pop, mlc in ((4, 2), (64, 3), (1000, 6), (20000, 6)):
    f = o.generate(pop, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(mlc), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    want = o.sr_fitness(*f, X, y)
    got = g.sr_fitness(*f, X, y)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (pop, "nan sets differ", int(np.isnan(got).sum()), int(np.isnan(want).sum()))
    ok = np.isfinite(want)
    assert np.array_equal(np.isposinf(got), np.isposinf(want))
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-5, atol=0)
    be = g.batch_evaluate(*(a[:50] for a in f), X, 1)
    bw = o.batch_evaluate(*(a[:50] for a in f), X, 1)
    bu, wu = be.view(np.uint32).copy(), bw.view(np.uint32).copy()
    bu[np.isnan(be)] = 0; wu[np.isnan(bw)] = 0
    assert np.array_equal(bu, wu), (pop, "batch_evaluate differs")
    print(f"asm depth {depth}: pop {pop} ok", flush=True)
print("ASM_SMOKE_OK")
