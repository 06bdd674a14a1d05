"""Crossover operators.  ``DefaultCrossover`` follows src/evogp/algorithm/crossover/default.py:8-66:
both parents and both subtree positions are drawn uniformly; the draws only build int32 index
tensors, the tree surgery is one ``tree_crossover`` kernel."""
from __future__ import annotations

import torch

from ..tree import Forest


class BaseCrossover:
    def __call__(self, forest: Forest, survivor_indices: torch.Tensor, target_cnt: int, fitness: torch.Tensor):
        raise NotImplementedError


class DefaultCrossover(BaseCrossover):
    def draw(self, tree_sizes: torch.Tensor, n_parents: int, target_cnt: int, device):
        """Index tensors of one crossover round (parents in [0, n_parents), positions = u % size)."""
        parents = torch.randint(0, n_parents, (2, target_cnt), dtype=torch.int32, device=device)
        raw = torch.randint(0, torch.iinfo(torch.int32).max, (2, target_cnt), dtype=torch.int32, device=device)
        left, right = parents[0], parents[1]
        left_pos = raw[0] % tree_sizes[left]
        right_pos = raw[1] % tree_sizes[right]
        return left, right, left_pos.to(torch.int32), right_pos.to(torch.int32)

    def __call__(self, forest: Forest, survivor_indices: torch.Tensor, target_cnt: int, fitness: torch.Tensor):
        parents = forest[survivor_indices]
        sizes = parents.batch_subtree_size[:, 0]
        left, right, left_pos, right_pos = self.draw(sizes, len(parents), target_cnt, sizes.device)
        return parents.crossover(left, right, left_pos, right_pos)
