"""GeneticProgramming — one generation = selection -> crossover -> mutation -> elites + offspring
(src/evogp/algorithm/genetic_programming.py:30-124).  The optional Pareto front by tree size is
kept (``enable_pareto_front``); it is torch-level bookkeeping."""
from __future__ import annotations

import torch

from ..tree import Forest
from .crossover import BaseCrossover
from .mutation import BaseMutation
from .selection import BaseSelection


class ParetoFront:
    """Best fitness (and tree) seen for every tree length."""

    def __init__(self, max_tree_len: int, input_len: int, output_len: int, device):
        self.fitness = torch.full((max_tree_len,), float("-inf"), dtype=torch.float32, device=device)
        self.solution = Forest.zero_generate(max_tree_len, max_tree_len, input_len, output_len)

    def update(self, fitness: torch.Tensor, forest: Forest) -> None:
        L = forest.max_tree_len
        sizes = forest.batch_subtree_size[:, 0].to(torch.int64)
        by_size = torch.where(sizes[None, :] == torch.arange(L, device=fitness.device)[:, None], fitness[None, :],
                              float("-inf"))
        best, arg = torch.max(by_size, dim=1)
        better = best > self.fitness
        self.fitness = torch.where(better, best, self.fitness)
        for name in ("batch_node_value", "batch_node_type", "batch_subtree_size"):
            cur = getattr(self.solution, name)
            setattr(self.solution, name, torch.where(better[:, None], getattr(forest, name)[arg], cur))

    def __str__(self):
        rows = [f"size: {i}, fitness: {float(f):.2e}, solution: {self.solution[i]}" for i, f in enumerate(self.fitness)]
        return "\n".join(rows)


class GeneticProgramming:
    def __init__(self, initial_forest: Forest, crossover: BaseCrossover, mutation: BaseMutation,
                 selection: BaseSelection, enable_pareto_front: bool = False):
        self.forest = initial_forest
        self.pop_size = initial_forest.pop_size
        self.crossover = crossover
        self.mutation = mutation
        self.selection = selection
        self.enable_pareto_front = enable_pareto_front
        if enable_pareto_front:
            f = initial_forest
            self.pareto_front = ParetoFront(f.max_tree_len, f.input_len, f.output_len, f.batch_node_value.device)

    def step(self, fitness: torch.Tensor) -> Forest:
        assert self.forest is not None, "forest is not initialized"
        assert fitness.shape == (self.forest.pop_size,), (
            f"fitness shape should be ({self.forest.pop_size}, ), but got {fitness.shape}")
        if self.enable_pareto_front:
            self.pareto_front.update(fitness, self.forest)
        elite_indices, survivor_indices = self.selection(self.forest, fitness)
        offspring = self.crossover(forest=self.forest, survivor_indices=survivor_indices,
                                   target_cnt=self.pop_size - elite_indices.shape[0], fitness=fitness)
        offspring = self.mutation(offspring)
        self.forest = self.forest[elite_indices] + offspring  # elites first (genetic_programming.py:122)
        return self.forest
