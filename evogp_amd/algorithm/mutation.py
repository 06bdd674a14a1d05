"""Mutation operators.  ``DefaultMutation`` follows src/evogp/algorithm/mutation/default.py:10-75:
each tree mutates with probability ``mutation_rate``; a mutating tree gets a freshly generated
random subtree (``tree_generate`` with the mutation descriptor) spliced in at a random position
(``tree_mutate``).  The Bernoulli mask is drawn on the DEVICE here (the reference draws it on the
CPU and uploads it, default.py:43 — SURVEY.md §8f N2)."""
from __future__ import annotations

import torch

from ..tree import MAX_STACK, Forest, GenerateDescriptor


class BaseMutation:
    def __call__(self, forest: Forest):
        raise NotImplementedError


class DefaultMutation(BaseMutation):
    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor

    def __call__(self, forest: Forest) -> Forest:
        dev = forest.batch_node_value.device
        mask = torch.rand(forest.pop_size, device=dev) < self.mutation_rate
        n_mut = int(mask.sum())
        if n_mut == 0:
            return forest
        chosen = forest[mask]
        donors = Forest.random_generate(pop_size=n_mut, descriptor=self.descriptor)
        raw = torch.randint(0, MAX_STACK, (n_mut,), dtype=torch.int32, device=dev)
        positions = (raw % chosen.batch_subtree_size[:, 0]).to(torch.int32)
        forest[mask] = chosen.mutate(positions, donors)
        return forest
