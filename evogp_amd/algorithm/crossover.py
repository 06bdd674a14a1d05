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


# ---- the non-default crossovers ---------------------------------------------------------------------------------------
# Reference: crossover/diversity.py, crossover/leaf_biased.py.  ``crossover_rate`` of the offspring are recombined, the
# rest are verbatim copies of random survivors.  Here both kinds come out of ONE tree_crossover launch over the whole
# target count: a copy is a crossover whose left position is -1 (the kernel's copy-left rule, mutation.cu:256-266), so
# there is no second gather and no concatenation.

class DiversityCrossover(BaseCrossover):
    """Recipient and donor are drawn independently (uniformly from the survivors, or by ``recipient_selector`` /
    ``donor_selector`` from the whole population), positions uniformly (diversity.py)."""

    def __init__(self, crossover_rate: float = 0.9, recipient_selector=None, donor_selector=None):
        self.crossover_rate = crossover_rate
        self.recipient_selector = recipient_selector
        self.donor_selector = donor_selector

    def _parents(self, selector, fitness, survivor_indices, n):
        if selector is not None:
            return selector(fitness, n).to(torch.int64)
        pick = torch.randint(0, survivor_indices.shape[0], (n,), device=survivor_indices.device)
        return survivor_indices.to(torch.int64)[pick]

    def _positions(self, forest: Forest, recipients, donors):
        sizes = forest.batch_subtree_size[:, 0].to(torch.int64)
        raw = torch.randint(0, torch.iinfo(torch.int32).max, (2, recipients.shape[0]), device=recipients.device)
        return raw[0] % sizes[recipients], raw[1] % sizes[donors]

    def __call__(self, forest: Forest, survivor_indices: torch.Tensor, target_cnt: int, fitness: torch.Tensor):
        target_cnt = int(target_cnt)
        n_cross = int(target_cnt * self.crossover_rate)
        recipients = self._parents(self.recipient_selector, fitness, survivor_indices, target_cnt)
        donors = self._parents(self.donor_selector, fitness, survivor_indices, target_cnt)
        rpos, dpos = self._positions(forest, recipients, donors)
        copy = torch.arange(target_cnt, device=rpos.device) >= n_cross      # the tail of the offspring are plain copies
        rpos = torch.where(copy, torch.full_like(rpos, -1), rpos)
        i32 = lambda t: t.to(torch.int32).contiguous()  # noqa: E731
        return forest.crossover(i32(recipients), i32(donors), i32(rpos), i32(dpos))


class LeafBiasedCrossover(DiversityCrossover):
    """As DiversityCrossover, but with probability ``leaf_bias`` both positions are leaves (leaf_biased.py): swaps of
    single terminals instead of whole subtrees."""

    def __init__(self, crossover_rate: float = 0.9, leaf_bias: float = 0.3, recipient_selector=None, donor_selector=None):
        super().__init__(crossover_rate, recipient_selector, donor_selector)
        self.leaf_bias = leaf_bias

    @staticmethod
    def _leaf_pos(sizes_rows: torch.Tensor) -> torch.Tensor:
        n, L = sizes_rows.shape
        live = torch.arange(L, device=sizes_rows.device)[None, :] < sizes_rows[:, :1]
        return torch.argmax(torch.rand((n, L), device=sizes_rows.device) * (live & (sizes_rows == 1)), dim=1)

    def _positions(self, forest: Forest, recipients, donors):
        rpos, dpos = super()._positions(forest, recipients, donors)
        sizes = forest.batch_subtree_size
        leaf = torch.rand(recipients.shape[0], device=recipients.device) < self.leaf_bias
        return (torch.where(leaf, self._leaf_pos(sizes[recipients]), rpos),
                torch.where(leaf, self._leaf_pos(sizes[donors]), dpos))


class CombinedDefaultCrossover(BaseCrossover):
    """DefaultCrossover on every sub-forest of a CombinedForest: ONE draw of parent pairs shared by all sub-forests (an
    offspring takes all of its expressions from the same two parents), independent cut points per sub-forest
    (crossover/combined_dafault.py:8-54)."""

    def __call__(self, forest, survivor_indices: torch.Tensor, target_cnt: int, fitness: torch.Tensor):
        from ..tree import CombinedForest

        survivors = forest[survivor_indices]
        dev = survivors.forests[0].batch_node_value.device
        left, right = torch.randint(0, len(survivors), (2, target_cnt), dtype=torch.int32, device=dev)
        children = []
        for sub in survivors.forests:
            sizes = sub.batch_subtree_size[:, 0].to(torch.int64)
            raw = torch.randint(0, 2**31 - 1, (2, target_cnt), device=dev)
            lpos = (raw[0] % sizes[left.long()]).to(torch.int32)
            rpos = (raw[1] % sizes[right.long()]).to(torch.int32)
            children.append(sub.crossover(left, right, lpos, rpos))
        return CombinedForest(children, forest.data_info)
