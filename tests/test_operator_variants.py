"""Host-logic tests (CPU, TEST-ONLY oracle-backed ops) of the selection and crossover variants: statistical behaviour of
the selectors, index ranges, and a full GeneticProgramming loop that mixes the variants."""
import numpy as np
import pytest
import torch

import cpu_ops
from test_mutation_variants import _check_well_formed, _forest


@pytest.fixture(scope="module", autouse=True)
def _cpu_ops():
    cpu_ops.register()
    from evogp_amd.tree import set_default_device, default_device

    old = default_device()
    set_default_device("cpu")
    yield
    set_default_device(old)


def test_selectors_prefer_fit_individuals():
    from evogp_amd.algorithm import RankSelector, RouletteSelector, TournamentSelector, TruncationSelector

    torch.manual_seed(0)
    n = 2000
    fitness = torch.linspace(0.0, 1.0, n)          # individual i has fitness i / n
    fitness[7] = float("nan"); fitness[11] = float("-inf")
    for sel, expect_mean in ((RankSelector(1.0), 2 / 3), (RankSelector(0.0), 1 / 2), (RouletteSelector(), 2 / 3),
                             (TruncationSelector(0.25), 0.875), (TournamentSelector(2), 2 / 3),
                             (TournamentSelector(4, replace=False), 0.8), (TournamentSelector(3, best_probability=0.7), None)):
        idx = sel(fitness, 50_000)
        assert idx.dtype == torch.int32 and idx.shape == (50_000,)
        assert int(idx.min()) >= 0 and int(idx.max()) < n
        got = float(fitness[idx.long()].nan_to_num(nan=0.0, neginf=0.0).mean())
        if expect_mean is not None:
            assert abs(got - expect_mean) < 0.02, (type(sel).__name__, got, expect_mean)
        else:
            assert got > 0.6
        if not isinstance(sel, RankSelector) or sel.sp > 0:
            assert int(((idx == 7) | (idx == 11)).sum()) < 200   # the invalid individuals are (almost) never chosen
    # without replacement nobody enters two tournaments of one pass
    c = TournamentSelector(4, replace=False).contenders(n, n // 4, "cpu")
    assert c.numel() == torch.unique(c).numel()


def test_selections_return_elites_and_survivors():
    from evogp_amd.algorithm import RankSelection, RouletteSelection, TournamentSelection, TruncationSelection

    f, _ = _forest(pop=400)
    fitness = torch.randn(400)
    best = torch.argsort(fitness, descending=True)[:8]
    for sel in (RankSelection(0.5, survivor_rate=0.3, elite_cnt=8), RouletteSelection(survivor_rate=0.3, elite_cnt=8),
                TournamentSelection(3, survivor_rate=0.3, elite_rate=0.02), TruncationSelection(0.3, elite_cnt=8)):
        elites, surv = sel(f, fitness)
        assert surv.shape == (120,) and elites.shape == (8,)
        assert torch.equal(elites.long(), best)
        assert elites.dtype == torch.int32 and surv.dtype == torch.int32


def test_crossover_variants_and_a_mixed_gp_loop():
    from evogp_amd.algorithm import (CombinedMutation, DeleteMutation, DiversityCrossover, GeneticProgramming, HoistMutation,
                                     InsertMutation, LeafBiasedCrossover, RankSelector, SinglePointMutation,
                                     TournamentSelection, TournamentSelector)

    f, desc = _forest(pop=300, funcs=("+", "-", "*", "/"))
    fitness = torch.randn(300)
    surv = torch.argsort(fitness, descending=True)[:90].to(torch.int32)
    for cx in (DiversityCrossover(0.8), LeafBiasedCrossover(0.8, leaf_bias=1.0),
               DiversityCrossover(0.5, recipient_selector=RankSelector(0.8), donor_selector=TournamentSelector(3))):
        child = cx(forest=f, survivor_indices=surv, target_cnt=250, fitness=fitness)
        assert child.pop_size == 250
        _check_well_formed(child)
    # with leaf_bias = 1 a recombined child has exactly the recipient's size (a leaf replaced by a leaf)
    torch.manual_seed(1)
    child = LeafBiasedCrossover(1.0, leaf_bias=1.0)(forest=f, survivor_indices=surv, target_cnt=64, fitness=fitness)
    sizes = set(f.batch_subtree_size[surv.long(), 0].tolist())
    assert set(child.batch_subtree_size[:, 0].tolist()) <= sizes

    X = torch.rand(64, 4) * 4 - 2
    y = (X[:, 0] * X[:, 1] - X[:, 2]).unsqueeze(1)
    algo = GeneticProgramming(f, LeafBiasedCrossover(0.9, 0.3),
                              CombinedMutation([HoistMutation(0.1), InsertMutation(0.1, desc.update(max_layer_cnt=2)),
                                                DeleteMutation(0.1), SinglePointMutation(0.2, desc)]),
                              TournamentSelection(4, survivor_rate=0.4, elite_rate=0.02))
    best = []
    for _ in range(5):
        fit = -algo.forest.SR_fitness(X, y)
        fit[torch.isnan(fit)] = -torch.inf
        best.append(float(fit.max()))
        algo.step(fit)
        assert algo.forest.pop_size == 300
        _check_well_formed(algo.forest)
    assert best[-1] >= best[0]


def test_classification_problem_on_iris():
    """configs[3] in miniature: multi-output classifier trees on an offline sklearn set, blocked reduction == unblocked."""
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.problem import Classification
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(0)
    prob = Classification(dataset="iris")
    assert prob.problem_dim == 4 and prob.solution_dim == 3
    desc = GenerateDescriptor(max_tree_len=32, input_len=4, output_len=3, using_funcs=["+", "-", "*", "/"], max_layer_cnt=4,
                              const_samples=[-1.0, 0.0, 1.0], out_prob=0.5)
    forest = Forest.random_generate(200, desc, keys=torch.tensor([3, 4], dtype=torch.uint32))
    acc = prob.evaluate(forest)
    assert acc.shape == (200,) and float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 and float(acc.max()) >= 1 / 3 - 1e-6
    small = Classification(prob.datapoints, prob.labels, block_bytes=64 * 1024)
    assert torch.equal(small.evaluate(forest), acc)
    single = Classification(prob.datapoints, prob.labels, multi_output=False)
    f1 = Forest.random_generate(50, desc.update(output_len=1), keys=torch.tensor([3, 4], dtype=torch.uint32))
    assert single.evaluate(f1).shape == (50,)
    algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=2)),
                              DefaultSelection(0.3, elite_rate=0.02))
    best = []
    for _ in range(4):
        fit = prob.evaluate(algo.forest)
        best.append(float(fit.max()))
        algo.step(fit)
    assert best[-1] >= best[0]


def test_rollout_problem_runs_policy_trees_in_a_batched_environment():
    """configs[4] in miniature (no simulator in this image): the rollout loop on the CPU, eager."""
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.problem import LinearTrackingEnv, PendulumEnv, RolloutProblem
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(0)
    for env, steps in ((LinearTrackingEnv(obs_dim=5, act_dim=2), 30), (PendulumEnv(), 40)):
        prob = RolloutProblem(env, steps, use_graph=False)
        desc = GenerateDescriptor(max_tree_len=32, input_len=prob.problem_dim, output_len=prob.solution_dim,
                                  using_funcs=["+", "-", "*", "/"], max_layer_cnt=4, const_samples=[-1.0, 0.0, 1.0, 0.5], out_prob=0.5)
        forest = Forest.random_generate(120, desc, keys=torch.tensor([8, 9], dtype=torch.uint32))
        fit = prob.evaluate(forest)
        assert fit.shape == (120,) and bool(torch.isfinite(fit).all())
        assert torch.equal(prob.evaluate(forest), fit)      # deterministic
        algo = GeneticProgramming(forest, DefaultCrossover(), DefaultMutation(0.2, desc.update(max_layer_cnt=2)),
                                  DefaultSelection(0.3, elite_rate=0.02))
        best = []
        for _ in range(4):
            fit = prob.evaluate(algo.forest)
            best.append(float(fit.max()))
            algo.step(fit)
        assert best[-1] >= best[0]
