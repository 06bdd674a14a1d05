"""Selection operators.  ``DefaultSelection`` follows src/evogp/algorithm/selection/default.py:8-71:
sort by fitness (descending), keep the top ``survival_rate`` fraction as parents and the top
``elite_cnt`` / ``elite_rate`` as elites that are copied unchanged.  Pure index arithmetic in torch."""
from __future__ import annotations

from typing import Optional

import torch

from ..tree import Forest


class BaseSelection:
    def __call__(self, forest: Forest, fitness: torch.Tensor):
        raise NotImplementedError


class DefaultSelection(BaseSelection):
    def __init__(self, survival_rate: float = 0.3, elite_cnt: Optional[int] = None,
                 elite_rate: Optional[float] = None):
        assert 0 <= survival_rate <= 1, "survival_rate should be in [0, 1]"
        assert elite_cnt is None or elite_rate is None, "elite_cnt and elite_rate should not be set at the same time"
        self.survival_rate = survival_rate
        self.elite_cnt = elite_cnt
        self.elite_rate = elite_rate

    def counts(self, pop_size: int):
        n_survive = int(pop_size * self.survival_rate)
        if self.elite_cnt is not None:
            n_elite = self.elite_cnt
        elif self.elite_rate is not None:
            n_elite = int(pop_size * self.elite_rate)
        else:
            n_elite = 0
        return n_elite, n_survive

    def __call__(self, forest: Forest, fitness: torch.Tensor):
        """-> (elite_indices int32, survivor_indices int32), both prefixes of the descending order.
        A stable sort is used so that replicated ranks of a sharded run agree on ties."""
        n_elite, n_survive = self.counts(forest.pop_size)
        order = torch.sort(fitness, descending=True, stable=True).indices
        return order[:n_elite].to(torch.int32), order[:n_survive].to(torch.int32)
