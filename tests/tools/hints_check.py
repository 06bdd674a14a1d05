"""EVOGP_TC_HINTS=1 (csrc/sr_tc.hip tc_hints): after a run of calls that marked nothing, the general compiler and the FULL register
build are no longer launched, and the last follow-up kernel takes whatever the next call marks after all.  A population of
another kind right after such a run -- functions only the general compiler knows (max, if), run-time bail-outs (sin of 2^17 and
more), operand stacks beyond the register stack -- must come out within the contract; so must the calls after the hints have
caught up.  Run by tests/test_gpu_parity.py in a process of its own (the switch is read once per process)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gpu_capi as g  # noqa: E402
from helpers import ARITH, assert_close_classes, assert_within_sensitivity, c2_dataset, depth2leaf, per_tree_tolerance, roulette_uniform, the_oracle  # noqa: E402

print("EVOGP_TC_HINTS =", os.environ.get("EVOGP_TC_HINTS"), flush=True)
oracle = the_oracle()
CS3 = [-1.0, 0.0, 1.0]
X, y = c2_dataset()
plain = oracle.generate(3000, 64, 10, 1, 0.5, 0.5, [5, 5], depth2leaf(6), roulette_uniform(ARITH), CS3)
want_plain = oracle.sr_fitness(*plain, X, y)
mixed = [a.copy() for a in oracle.generate(3000, 64, 10, 1, 0.5, 0.5, [6, 6], depth2leaf(4, 0.1), roulette_uniform([0, 1, 2, 3, 4, 9, 10, 14]), CS3)]   # (ternary: 4 layers are at most 40 nodes)
assert mixed[2][:, 0].max() <= 64
# a left-deep chain of 31 subtractions: operand stack of 32 entries (beyond the register stacks)
n = 63
mixed[0][7] = 0; mixed[1][7] = 0; mixed[2][7] = 0
k = (n - 1) // 2
mixed[1][7, :k] = 3; mixed[0][7, :k] = 2.0; mixed[2][7, :k] = n - 2 * np.arange(k)
mixed[1][7, k:n] = 0; mixed[0][7, k:n] = np.arange(n - k) % 10; mixed[2][7, k:n] = 1
assert oracle.validate_tree(mixed[1][7], mixed[2][7]) == 0
Xb = X.copy(); Xb[:, 3] *= 1.0e5                         # sin of arguments around 2^17: run-time bail-outs
want_mixed, tol, unstable = per_tree_tolerance(oracle, tuple(mixed), Xb, y)
for round_ in range(2):
    for _ in range(24):                                  # more calls than the hints remember
        got = g.sr_fitness(*plain, X, y)
    assert_close_classes(got, want_plain, 1e-5, what="plain forest")
    for call in range(3):                                # the first call meets stale hints, the later ones fresh ones
        got = g.sr_fitness(*mixed, Xb, y)
        bad = np.flatnonzero((np.isnan(got) != np.isnan(want_mixed)) & ~unstable)
        if len(bad):
            print(f"round {round_} call {call}: {len(bad)} trees with another NaN class, e.g.", bad[:8], "got", got[bad[:8]], "want", want_mixed[bad[:8]],
                  "bits", [hex(int(b)) for b in got[bad[:8]].view(np.uint32)], "lengths", mixed[2][bad[:8], 0], flush=True)
            for t in bad[:3]:
                print("tree", t, "types", mixed[1][t, :mixed[2][t, 0]].tolist(), "values", mixed[0][t, :mixed[2][t, 0]].tolist(), flush=True)
        assert_within_sensitivity(got, want_mixed, tol, unstable, f"mixed forest, round {round_}, call {call}", max_unstable=0.2, min_tight=0.2)

print("hints ok")
