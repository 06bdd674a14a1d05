"""Randomised cross-check (fixed seeds) of tree_SR_fitness against the per-datapoint outputs of batch_evaluate: random
subsets of the functions the threaded code handles, dataset sizes on both sides of the K = 4 / K = 8 switch and of the
tile boundaries, variable counts, constants, tree lengths.  Both sides run the device math library, so the only
difference allowed is the summation order (tests/tools/fuzz_tc.py is the open-ended version)."""
import numpy as np
import pytest

from helpers import depth2leaf, roulette_uniform

pytestmark = pytest.mark.gpu
HANDLED = [1, 2, 3, 4, 14, 15, 16, 20, 21, 22, 23, 25, 26, 27, 28]


@pytest.mark.parametrize("seed", range(12))
def test_sr_fitness_random_function_sets_match_batch_evaluate(oracle, seed):
    import gpu_capi as g

    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(2, len(HANDLED) + 1))
    funcs = sorted({1, 2} | {int(x) for x in rng.choice(HANDLED, k, replace=False)})
    var_len = int(rng.integers(1, 12)); D = int(rng.choice([1, 7, 64, 100, 256, 257, 777, 1024, 2000]))
    L = int(rng.choice([16, 32, 64])); mlc = int(rng.integers(2, 7)); pop = int(rng.integers(200, 2500))
    consts = rng.choice([-1, 0, 1, 0.5, -2.5, 3.0, 1e-3, 100.0], 4).astype(np.float32)
    f = oracle.generate(pop, L, var_len, 1, 0.0, float(rng.uniform(0.1, 0.6)), [int(rng.integers(1, 1 << 30)), seed], depth2leaf(mlc),
                        roulette_uniform(funcs), consts)
    X = (rng.standard_normal((D, var_len)) * float(rng.choice([0.5, 3.0, 50.0]))).astype(np.float32)
    y = rng.standard_normal((D, 1)).astype(np.float32)
    outs = g.batch_evaluate(*f, X, 1)[:, :, 0]
    for mse in (True, False):
        got = g.sr_fitness(*f, X, y, mse).astype(np.float64)
        with np.errstate(all="ignore"):
            d = outs - y[:, 0][None, :]
            ref = ((d * d) if mse else np.abs(d)).astype(np.float64).mean(1).astype(np.float32).astype(np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isposinf(got), np.isposinf(ref)), (funcs, D)
        fin = np.isfinite(got) & np.isfinite(ref)
        rel = np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30)
        assert rel.size == 0 or rel.max() <= 1e-4, (funcs, D, float(rel.max()))


def test_fitness_words_do_not_depend_on_the_work_distribution():
    """scripts/dbg/pool_soak.py in two processes -- the round-6 workgroup pools (default) and every batch drawn from the XCD counters
    (EVOGP_TC_STATIC=0, round 5's distribution): random populations up to 450 k trees, five function sets, rows of 16 .. 128 nodes; each
    configuration called through the reference's operator, under the forest's mask and on a second stream.  Inside a process all words
    must be equal (the script's exit code); between the processes the per-configuration checksums must be."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs = []
    for extra in ({}, {"EVOGP_TC_STATIC": "0"}):
        env = dict(os.environ, SOAK_CASES="14", SOAK_SEED="3", **extra)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dbg", "pool_soak.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        logs.append([l for l in r.stdout.splitlines() if l.startswith("case")])
    assert len(logs[0]) == 14 and logs[0] == logs[1]


def test_fitness_words_do_not_depend_on_the_compilers_workgroup_size():
    """scripts/dbg/packed_wg_check.py in two processes: the packed program compiler as workgroups of 256 threads and of 64 (one wave, the
    default from 120 k trees on) on forests beyond that size in the modes the soak above does not reach -- 4, 6 and 10 outputs, rows of
    128 and 256 nodes, functions behind the generic stubs.  The checksums of the fitness words must be equal."""
    import os, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs = []
    for wg in ("256", "64"):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dbg", "packed_wg_check.py")], env=dict(os.environ, EVOGP_TC_PACKED_WG=wg),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        logs.append([l for l in r.stdout.splitlines() if l.startswith("case")])
    assert len(logs[0]) == 5 and all("repeat equal True" in l for l in logs[0]) and logs[0] == logs[1]
