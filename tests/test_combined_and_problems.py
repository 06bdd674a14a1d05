"""Host-logic tests (CPU, TEST-ONLY oracle-backed ops) of the callers on either side of the hot path that are pure
routing: CombinedForest / CombinedTree with their default operators, the Transformation and CustomLoss problems
(reference: tree/combined_forest.py, tree/combined_tree.py, crossover/combined_dafault.py, mutation/combined_default.py,
problem/transformation.py, problem/custom_loss.py)."""
import numpy as np
import pytest
import torch

import cpu_ops
from test_mutation_variants import _check_well_formed


@pytest.fixture(scope="module", autouse=True)
def _cpu_ops():
    cpu_ops.register()
    from evogp_amd.tree import set_default_device, default_device

    old = default_device()
    set_default_device("cpu")
    yield
    set_default_device(old)


INFO = {"A": ["x", "y"], "B": ["y", "z", "x"]}


def _combined(pop=300, seed=1):
    from evogp_amd.tree import CombinedForest, GenerateDescriptor

    torch.manual_seed(seed)
    ds = [GenerateDescriptor(max_tree_len=64, input_len=len(cols), output_len=1, using_funcs=["+", "-", "*"], max_layer_cnt=4,
                             const_samples=[-1.0, 0.5, 2.0]) for cols in INFO.values()]
    return CombinedForest.random_generate(pop, INFO, ds), ds


def test_combined_forest_routes_named_columns():
    cf, _ = _combined()
    assert cf.output_names == ["A", "B"] and cf.input_names == ["x", "y", "z"]
    assert (cf.input_len, cf.output_len, len(cf)) == (3, 2, 300)
    cols = {k: torch.randn(17) for k in "xyz"}
    res = cf.batch_forward(cols)
    assert res["A"].shape == (300, 17, 1) and res["B"].shape == (300, 17, 1)
    # sub-forest B reads (y, z, x) in that order
    want = cf.forests[1].batch_forward(torch.stack([cols["y"], cols["z"], cols["x"]], dim=1))
    assert torch.equal(res["B"].nan_to_num(7.0), want.nan_to_num(7.0))
    # one row per individual
    rows = {k: torch.randn(300) for k in "xyz"}
    one = cf.forward(rows)
    assert one["A"].shape == (300, 1)
    want = cf.forests[0].forward(torch.stack([rows["x"], rows["y"]], dim=1))
    assert torch.equal(one["A"].nan_to_num(7.0), want.nan_to_num(7.0))


def test_combined_indexing_concatenation_and_tree_view():
    from evogp_amd.tree import CombinedForest, CombinedTree

    cf, _ = _combined()
    t = cf[5]
    assert isinstance(t, CombinedTree) and t.A is t.trees[0] and t.B is t.trees[1]
    cols = {k: torch.randn(9) for k in "xyz"}
    got = t.forward(cols)
    want = cf.batch_forward(cols)
    assert torch.equal(got["B"].nan_to_num(7.0), want["B"][5].nan_to_num(7.0))
    scalar = t.forward({k: v[0] for k, v in cols.items()})
    assert torch.allclose(scalar["A"].reshape(-1).nan_to_num(7.0), want["A"][5, 0].reshape(-1).nan_to_num(7.0))
    part = cf[torch.tensor([3, 1, 4])]
    assert isinstance(part, CombinedForest) and len(part) == 3
    assert torch.equal(part.forests[0].batch_node_value[1], cf.forests[0].batch_node_value[1])
    both = cf[:10] + part
    assert len(both) == 13 and torch.equal(both.forests[1].batch_subtree_size[10], cf.forests[1].batch_subtree_size[3])
    cf[0] = t
    assert torch.equal(cf.forests[0].batch_node_value[0], cf.forests[0].batch_node_value[5])
    cf[1:4] = part
    assert torch.equal(cf.forests[1].batch_node_type[2], part.forests[1].batch_node_type[1])
    assert sum(1 for _ in cf) == 300


def test_combined_default_operators_in_the_gp_loop():
    from evogp_amd.algorithm import CombinedDefaultCrossover, CombinedDefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.problem import CustomLoss

    cf, ds = _combined(pop=400)
    g = torch.Generator().manual_seed(0)
    cols = {k: torch.rand(32, generator=g) * 2 - 1 for k in "xyz"}
    target = cols["x"] * cols["y"] + cols["z"]

    def loss(x, y, z, target, A, B):  # fixed columns first, then the outputs of the forest
        return torch.mean((A + B - target) ** 2)

    problem = CustomLoss({**cols, "target": target}, loss)
    algo = GeneticProgramming(cf, CombinedDefaultCrossover(), CombinedDefaultMutation(0.4, [d.update(max_layer_cnt=3) for d in ds]),
                              DefaultSelection(0.3, elite_rate=0.02))
    best = []
    for _ in range(6):
        fit = problem.evaluate(algo.forest)
        assert fit.shape == (400,)
        fit = torch.nan_to_num(fit, nan=float("-inf"))
        best.append(float(fit.max()))
        algo.step(fit)
        for f in algo.forest.forests:
            assert f.pop_size == 400
            _check_well_formed(f)
    assert all(b >= a - 1e-6 for a, b in zip(best, best[1:])), best  # elites are kept
    # the loss sees exactly batch_forward's outputs
    out = algo.forest.batch_forward(cols)
    want = -torch.mean((out["A"].squeeze(-1) + out["B"].squeeze(-1) - target[None, :]) ** 2, dim=1)
    got = problem.evaluate(algo.forest)
    assert torch.allclose(got.nan_to_num(3.0), want.nan_to_num(3.0), rtol=1e-6, atol=1e-6)


def test_transformation_is_the_reference_formula():
    from evogp_amd.problem import Transformation
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(2)
    desc = GenerateDescriptor(max_tree_len=64, input_len=5, output_len=1, using_funcs=["+", "-", "*"], max_layer_cnt=4,
                              const_samples=[-1.0, 0.5, 2.0])
    forest = Forest.random_generate(500, desc)
    X = torch.randn(120, 5)
    y = X[:, 0] * X[:, 1] - X[:, 2] + 0.1 * torch.randn(120)
    prob = Transformation(X, y)
    fit = prob.evaluate(forest)
    out = forest.batch_forward(X).squeeze()
    od, ld = out - torch.mean(out), y - torch.mean(y)   # transformation.py:36-43, literally
    want = torch.abs(torch.sum(od * ld, dim=1) / torch.sqrt(torch.sum(od**2, dim=1) * torch.sum(ld**2)))
    assert torch.allclose(fit.nan_to_num(9.0), want.nan_to_num(9.0), rtol=1e-5, atol=1e-6)
    # the textbook variant equals numpy's corrcoef row by row and never returns NaN
    fit2 = Transformation(X, y, per_tree_mean=True).evaluate(forest)
    assert bool(torch.isfinite(fit2).all())
    o = out.numpy().astype(np.float64)
    for r in (0, 7, 123, 499):
        if np.std(o[r]) > 1e-6 and np.isfinite(o[r]).all():
            assert abs(abs(np.corrcoef(o[r], y.numpy())[0, 1]) - float(fit2[r])) < 1e-4
    feats = Transformation(X, y, per_tree_mean=True).new_feature(forest, 30, 6)
    assert feats.shape == (120, 6)
