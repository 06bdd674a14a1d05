"""GPU: tree_SR_fitness WITHOUT a function mask -- the operator the reference's own Python calls (tree/forest.py:340-351,
torch_wrapper.cu:235-284) -- takes the kernels a caller's mask would choose, from what the engine observes on the device (round 6,
csrc/sr_tc.hip tc_learned_class / tc_detect_class, csrc/sr_fitness.hip run_population):

  * the first call on a population shape looks at its own forest (one pass, waited for) and is exact;
  * later calls read the observation the last completed call left in host memory -- the arithmetic line and ONE array of records for a
    forest of + - * /, bit-equal to the call under a mask;
  * an observation that is out of date (another function set at the same shape) costs one call its speed, never a result: the trees
    the chosen line cannot take go to the register kernels, and the call's own observation puts the next one right;
  * inside a stream capture nothing is waited for: the capture takes round 5's generic line and replays correctly."""
import numpy as np
import pytest

from helpers import ARITH, assert_close_classes, assert_within_sensitivity, c2_dataset, depth2leaf, per_tree_tolerance, roulette_uniform
from test_gpu_tc_wide import handler_histogram

pytestmark = pytest.mark.gpu
SIN, NEG, MAX = 14, 25, 8
CS = [-1.0, 0.0, 1.0]
ARITH_MASK = sum(1 << f for f in ARITH)


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


def forest_of(oracle, pop, funcs, key):
    return oracle.generate(pop, 64, 10, 1, 0.0, 0.5, [key, 1], depth2leaf(6), roulette_uniform(funcs), CS)


def same_words(a, b):
    return np.array_equal(np.where(np.isnan(a), 0, a).view(np.uint32), np.where(np.isnan(b), 0, b).view(np.uint32)) and np.array_equal(np.isnan(a), np.isnan(b))


def test_unhinted_call_takes_the_kernels_of_the_mask(g, oracle):
    import evogp_amd

    pop = 30_000
    X, y = c2_dataset()
    f = forest_of(oracle, pop, ARITH, 11)
    evogp_amd.release_workspaces()
    first = g.sr_fitness(*f, X, y, forget=True)                    # no observation: the call looks at its forest
    array = (pop * 256 + 4095) // 4096 * 4096
    assert array <= evogp_amd.program_buffer_bytes() <= array + array // 8 + 8 * 256, "an unhinted call on + - * / holds more than one array of records"
    h1 = handler_histogram(g, pop, fold_twins=False)
    later = g.sr_fitness(*f, X, y, forget=False)                   # the observation of the first call
    h2 = handler_histogram(g, pop, fold_twins=False)
    masked = g.sr_fitness(*f, X, y, func_mask=ARITH_MASK)
    h3 = handler_histogram(g, pop, fold_twins=False)
    assert h1 == h2 == h3 and h1["skip"] <= 2
    assert same_words(first, later) and same_words(first, masked)
    assert array <= evogp_amd.program_buffer_bytes() <= array + array // 8 + 8 * 256
    assert_close_classes(first, oracle.sr_fitness(*f, X, y), 1e-5, what="unhinted + - * /")


@pytest.mark.parametrize("funcs,name", [(ARITH + [SIN, NEG], "unary"), (ARITH + [MAX, SIN], "generic")])
def test_out_of_date_observation_costs_speed_not_results(g, oracle, funcs, name):
    import evogp_amd

    pop = 20_000
    X, y = c2_dataset()
    fa, fu = forest_of(oracle, pop, ARITH, 5), forest_of(oracle, pop, funcs, 6)
    want_a = oracle.sr_fitness(*fa, X, y)
    want_u, tol_u, unstable_u = per_tree_tolerance(oracle, fu, X, y)   # (sin: the device library against the host's, tests/helpers.py)
    evogp_amd.release_workspaces()
    exact = g.sr_fitness(*fu, X, y, forget=True)                   # looked at: every tree through the threaded code
    h_exact = handler_histogram(g, pop)
    assert h_exact["skip"] <= 0.02 * pop, h_exact["skip"]
    a0 = g.sr_fitness(*fa, X, y, forget=True)                      # observation: + - * /
    stale = g.sr_fitness(*fu, X, y, forget=False)                  # ... which is wrong for this forest
    h_stale = handler_histogram(g, pop)
    assert h_stale["skip"] > 0.3 * pop, "the out-of-date observation was not used"
    assert_within_sensitivity(exact.astype(np.float64), want_u, tol_u, unstable_u, f"{name}: the call that looked")
    assert_within_sensitivity(stale.astype(np.float64), want_u, tol_u, unstable_u, f"{name}: call under an out-of-date observation")
    healed = g.sr_fitness(*fu, X, y, forget=False)                 # the stale call's own observation
    assert handler_histogram(g, pop)["skip"] == h_exact["skip"]
    assert same_words(healed, exact)
    back = g.sr_fitness(*fa, X, y, forget=False)                   # the other way round: the wider line takes + - * / as it is
    assert same_words(back, a0)
    again = g.sr_fitness(*fa, X, y, forget=False)                  # ... and the arithmetic line is back
    assert same_words(again, a0)
    assert_close_classes(a0, want_a, 1e-5, what="+ - * /")


def test_shapes_keep_their_own_observations(g, oracle):
    """two populations of different sizes alternate: neither spoils the other's observation"""
    X, y = c2_dataset()
    fa, fu = forest_of(oracle, 9000, ARITH, 7), forest_of(oracle, 7000, ARITH + [SIN], 8)
    ea, eu = g.sr_fitness(*fa, X, y, forget=True), g.sr_fitness(*fu, X, y, forget=True)
    g.L.evogp_hip_debug_forget_function_classes()
    for _ in range(3):
        a = g.sr_fitness(*fa, X, y, forget=False)
        assert handler_histogram(g, 9000)["skip"] <= 2 and same_words(a, ea)
        u = g.sr_fitness(*fu, X, y, forget=False)
        assert handler_histogram(g, 7000)["skip"] <= 0.02 * 7000 and same_words(u, eu)


def test_unhinted_call_inside_a_capture_does_not_wait(g, oracle):
    import torch

    pop = 5000
    X, y = c2_dataset()
    f = forest_of(oracle, pop, ARITH + [SIN], 9)
    eager = g.sr_fitness(*f, X, y, forget=True)
    a = [g.dev(f[0], np.float32), g.dev(f[1], np.int16), g.dev(f[2], np.int16), g.dev(X, np.float32), g.dev(y, np.float32)]
    fit = torch.zeros(pop, dtype=torch.float32, device=g.DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                 # (a stream's first fitness call allocates its scratch block: not inside a capture)
        rc = g.L.evogp_hip_sr_fitness_hinted(pop, X.shape[0], 64, X.shape[1], 1, 1, *[x.data_ptr() for x in a], fit.data_ptr(), 0, ARITH_MASK | (1 << SIN),
                                             side.cuda_stream)
        assert rc == 0
    side.synchronize()
    g.L.evogp_hip_debug_forget_function_classes()                 # no observation: a capture cannot look (nothing is waited for inside one)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = g.L.evogp_hip_sr_fitness(pop, X.shape[0], 64, X.shape[1], 1, 1, *[x.data_ptr() for x in a], fit.data_ptr(), 0, side.cuda_stream)
            assert rc == 0
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        fit.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert same_words(fit.cpu().numpy(), eager)
