"""GPU: the rollout path (BASELINE configs[4]): a multi-output policy population evaluated once per
environment step (src/evogp/problem/brax_problem.py:54-93).

  * the prepared forward pass (csrc/evaluate_prepared.hip: operation lists built once per forest) must return exactly what
    evogp_hip_evaluate returns — same operations in the same order, so bit for bit — and what the oracle returns;
  * a 50-step rollout through RolloutProblem (prepared forward pass inside a replayed HIP graph) against the same loop in
    numpy with oracle.evaluate as the policy."""
import numpy as np
import pytest

from helpers import ARITH, depth2leaf, fbits, roulette_uniform

pytestmark = pytest.mark.gpu
IF, LDIV, MAX, MIN, LT, SIN, TANH, NEG, ABS, SQRT = 0, 5, 8, 9, 10, 14, 19, 25, 26, 27


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


@pytest.mark.parametrize("funcs,out_len,var_len,L,mlc", [(ARITH, 6, 17, 256, 6), (ARITH + [IF, LDIV, MAX, MIN, LT, NEG, ABS, SQRT], 6, 17, 256, 5),
                                                         (ARITH + [SIN, TANH], 3, 4, 64, 6), (ARITH, 2, 1, 32, 5), (ARITH, 32, 40, 128, 6)])
def test_prepared_forward_is_what_evaluate_returns(g, oracle, rng, funcs, out_len, var_len, L, mlc):
    pop = 6000
    cs = np.linspace(-1, 1, 100).astype(np.float32)
    forest = oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [out_len, L], depth2leaf(mlc), roulette_uniform(funcs), cs)
    v, t, s = (a.copy() for a in forest)
    # trees the operation lists cannot express or must answer specially
    s[5, 0] = 0                                            # empty: NaN row
    t[6, :3] = [3, 0, 0]; s[6, :3] = [3, 1, 1]; v[6, 0] = 1; s[6, 0] = 2   # truncated: stack underflow -> NaN row
    bigger = np.flatnonzero(s[:, 0] > 7)[:3]
    for r in bigger:                                      # subtree sizes that do not describe the tree: the stack interpreter takes it
        s[r, 1] += 1
    X = rng.normal(0, 1, (pop, var_len)).astype(np.float32)
    got, left = g.evaluate_prepared(v, t, s, X, out_len, steps=3)
    want = g.evaluate(v, t, s, X, out_len)
    assert left >= len(bigger)
    assert np.array_equal(fbits(got), fbits(want)), "prepared forward pass differs from evogp_hip_evaluate"
    if SIN not in funcs:
        ora = oracle.evaluate(v, t, s, X, out_len)
        ok = np.ones(pop, bool); ok[[5, 6]] = False   # malformed trees: NaN here, undefined in the reference
        assert np.array_equal(fbits(got[ok]), fbits(ora[ok])), "prepared forward pass differs from the oracle"
        assert np.isnan(got[[5, 6]]).all()


def test_forest_forward_switches_to_the_prepared_pass_and_invalidates(g, oracle, rng):
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.tree import Forest

    f = oracle.generate(3000, 64, 5, 3, 0.5, 0.5, [1, 2], depth2leaf(6), roulette_uniform(ARITH), [-1.0, 0.5, 2.0])
    forest = Forest(5, 3, *(torch.from_numpy(a).cuda() for a in f))
    x = torch.from_numpy(rng.normal(0, 1, (3000, 5)).astype(np.float32)).cuda()
    first = forest.forward(x)                      # stack interpreter
    assert getattr(forest, "_prepared", None) is None
    second = forest.forward(x)                     # second call on the unchanged forest: operation lists
    assert forest._prepared is not None
    assert torch.equal(first.view(torch.int32), second.view(torch.int32))
    forest.batch_node_value[0, 0] = 3.0            # an in-place edit invalidates the lists (version counter)
    third = forest.forward(x)
    want = oracle.evaluate(forest.batch_node_value.cpu().numpy(), f[1], f[2], x.cpu().numpy(), 3)
    assert np.array_equal(fbits(third.cpu().numpy()), fbits(want))


def test_rollout_of_50_steps_against_the_oracle_loop(g, oracle, rng):
    """RolloutProblem (prepared forward pass, one step captured as a HIP graph and replayed) vs the same loop in numpy with
    oracle.evaluate as the policy.  The environment is elementwise (no matrix product: the two sides then differ only in the
    order of two small sums), the action squashing is a clamp (exact)."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.problem import RolloutProblem
    from evogp_amd.tree import Forest

    pop, obs_dim, act_dim, steps = 4000, 17, 6, 50

    class Env:
        def __init__(self, device=None):
            self.device, self.obs_dim, self.act_dim = device, obs_dim, act_dim
            self.x0 = torch.linspace(-1, 1, obs_dim).to(device)

        def reset(self, n):
            return self.x0[None, :].repeat(n, 1)

        def observe(self, state):
            return state

        def step(self, state, action):
            push = torch.cat([action, action, action[:, :obs_dim - 2 * act_dim]], dim=1)
            nxt = 0.95 * state + 0.1 * push
            reward = -(nxt * nxt).sum(1) - 0.1 * (action * action).sum(1)
            return nxt, reward, nxt.abs().amax(1) > 5.0

    cs = np.linspace(-1, 1, 100).astype(np.float32)
    f = oracle.generate(pop, 256, obs_dim, act_dim, 0.5, 0.5, [4, 2], depth2leaf(6), roulette_uniform(ARITH), cs)
    forest = Forest(obs_dim, act_dim, *(torch.from_numpy(a).cuda() for a in f))
    clamp = lambda a: a.clamp(-1, 1)  # noqa: E731
    for use_graph in (True, False):
        got = RolloutProblem(Env("cuda"), steps, output_transform=clamp, use_graph=use_graph).evaluate(forest).cpu().numpy()
        assert forest._prepared is not None, "the rollout must run from the operation lists"
        # the same loop in numpy (float32 throughout), oracle.evaluate as the policy
        state = np.tile(np.linspace(-1, 1, obs_dim, dtype=np.float32)[None, :], (pop, 1))
        total = np.zeros(pop, np.float32); done = np.zeros(pop, bool)
        for _ in range(steps):
            with np.errstate(all="ignore"):
                action = np.clip(oracle.evaluate(*f, state, act_dim), -1, 1).astype(np.float32)
                action = np.where(np.isnan(action), np.float32(np.nan), action)
                push = np.concatenate([action, action, action[:, :obs_dim - 2 * act_dim]], 1)
                nxt = (np.float32(0.95) * state + np.float32(0.1) * push).astype(np.float32)
                reward = (-(nxt * nxt).sum(1) - np.float32(0.1) * (action * action).sum(1)).astype(np.float32)
                now_done = np.abs(nxt).max(1) > 5.0
                reward = np.nan_to_num(reward, nan=-1e6, posinf=-1e6, neginf=-1e6).astype(np.float32)
                total = total + np.where(done, np.float32(0), reward)
                done = done | now_done | ~np.isfinite(nxt).all(1)
                state = np.where(done[:, None], state, np.nan_to_num(nxt)).astype(np.float32)
        assert np.allclose(got, total, rtol=2e-4, atol=1e-3), (use_graph, np.abs(got - total).max())
