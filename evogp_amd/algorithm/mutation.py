"""Mutation operators.  ``DefaultMutation`` follows src/evogp/algorithm/mutation/default.py:10-75:
each tree mutates with probability ``mutation_rate``; a mutating tree gets a freshly generated
random subtree (``tree_generate`` with the mutation descriptor) spliced in at a random position
(``tree_mutate``).  The Bernoulli mask is drawn on the DEVICE here (the reference draws it on the
CPU and uploads it, default.py:43 — SURVEY.md §8f N2)."""
from __future__ import annotations

from typing import Optional

import torch

from ..tree import MAX_STACK, Forest, GenerateDescriptor, NType


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


# ---- structural and point mutations -----------------------------------------------------------
# The reference builds these from boolean-mask gathers, a per-row "vmap_subtree" gather program and tree_mutate
# (mutation/mutation_utils.py:6-48, hoist.py:43-75, insert.py:45-85, delete.py:44-105, single_point.py:43-126,
# multi_point.py, single_const.py, multi_const.py).  Every structural one is a subtree replacement whose donor is a
# subtree of an existing tree — exactly what tree_crossover does — so here each is ONE (Insert: two) native launch over
# the whole population with no gather program and no host sync: trees that do not mutate get left position -1, which the
# kernel answers with a verbatim copy (mutation.cu:256-266).  The point mutations are elementwise torch programs over
# the (pop, L) grid.

def _rand_below(high: torch.Tensor) -> torch.Tensor:
    """uniform integer in [0, high) per element (high >= 1), int64, on high's device"""
    return torch.randint(0, 2**31 - 1, high.shape, device=high.device) % high.clamp(min=1).to(torch.int64)


def _mutate_mask(forest: Forest, rate: float) -> torch.Tensor:
    return torch.rand(forest.pop_size, device=forest.batch_node_value.device) < rate


class HoistMutation(BaseMutation):
    """Pick a subtree, pick a subtree inside it, and put the inner one in the outer one's place (hoist.py:43-75): trees
    shrink.  ``reference_indexing=True`` reproduces the reference's draw of the inner position, which is an ABSOLUTE node
    index in [0, size(outer)) rather than an offset inside the outer subtree (hoist.py:58-68)."""

    def __init__(self, mutation_rate: float, reference_indexing: bool = False):
        self.mutation_rate = mutation_rate
        self.reference_indexing = reference_indexing

    def __call__(self, forest: Forest) -> Forest:
        dev = forest.batch_node_value.device
        sizes = forest.batch_subtree_size
        p = _rand_below(sizes[:, 0])
        inner = _rand_below(sizes.gather(1, p[:, None]).squeeze(1))
        q = inner if self.reference_indexing else p + inner
        ar = torch.arange(forest.pop_size, dtype=torch.int32, device=dev)
        p = torch.where(_mutate_mask(forest, self.mutation_rate), p, -1)
        return forest.crossover(ar, ar, p.to(torch.int32), q.to(torch.int32))


class DeleteMutation(BaseMutation):
    """Pick a function node and replace it by one of its children (delete.py:44-105).  ``max_mutatable_size`` restricts
    the choice to nodes whose subtree is at most that large; as in the reference, a tree without an eligible node uses
    its root."""

    def __init__(self, mutation_rate: float, max_mutatable_size: Optional[int] = None):
        self.mutation_rate = mutation_rate
        self.max_mutatable_size = max_mutatable_size

    def __call__(self, forest: Forest) -> Forest:
        dev = forest.batch_node_value.device
        sizes = forest.batch_subtree_size.to(torch.int64)
        n, L = sizes.shape
        live = torch.arange(L, device=dev)[None, :] < sizes[:, :1]
        eligible = live & (sizes > 1)
        if self.max_mutatable_size:
            eligible &= sizes <= self.max_mutatable_size
        p = torch.argmax(torch.rand((n, L), device=dev) * eligible, dim=1)          # a random eligible node (0 if none)
        arity = ((forest.batch_node_type.gather(1, p[:, None]).squeeze(1).to(torch.int64) & NType.TYPE_MASK)
                 - NType.UFUNC + 1).clamp(min=1)
        nth = 1 + _rand_below(arity)
        c1 = (p + 1).clamp(max=L - 1)
        c2 = (c1 + sizes.gather(1, c1[:, None]).squeeze(1)).clamp(max=L - 1)
        c3 = (c2 + sizes.gather(1, c2[:, None]).squeeze(1)).clamp(max=L - 1)
        q = torch.where(nth == 3, c3, torch.where(nth == 2, c2, c1))
        mask = _mutate_mask(forest, self.mutation_rate) & (sizes[:, 0] > 1)
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        return forest.crossover(ar, ar, torch.where(mask, p, -1).to(torch.int32), q.to(torch.int32))


class InsertMutation(BaseMutation):
    """Pick a subtree, generate a small random tree, hang the subtree into a random position (>= 1) of the new tree and
    put the result where the subtree was (insert.py:45-85): trees grow by one random operator layer."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor

    def __call__(self, forest: Forest) -> Forest:
        dev = forest.batch_node_value.device
        n = forest.pop_size
        mask = _mutate_mask(forest, self.mutation_rate)
        p = _rand_below(forest.batch_subtree_size[:, 0])
        fresh = Forest.random_generate(pop_size=n, descriptor=self.descriptor)
        fsize = fresh.batch_subtree_size[:, 0].to(torch.int64)
        r = 1 + _rand_below(fsize - 1)                                  # a position below the new root
        mask = mask & (fsize > 1)                                       # a single-node tree has no such position
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        both = fresh + forest                                           # rows [0, n): new trees, [n, 2n): the population
        grafted = both.crossover(ar, ar + n, r.to(torch.int32), p.to(torch.int32))
        return forest.mutate(torch.where(mask, p, -1).to(torch.int32), grafted)


def _same_kind_values(ntype: torch.Tensor, value: torch.Tensor, d: GenerateDescriptor, input_len: int, output_len: int,
                      modify_output: bool) -> torch.Tensor:
    """For every node a fresh payload of the node's own kind: a function of the same arity drawn from the descriptor's
    per-arity roulettes (an output node keeps, or with modify_output redraws, its output index in the high half-word),
    a variable index, or a constant sample (single_point.py:70-124)."""
    dev = value.device
    kind = ntype.to(torch.int64) & NType.TYPE_MASK
    is_out = (ntype.to(torch.int64) & NType.OUT_NODE) != 0
    shape = value.shape
    # the per-arity roulettes are cumulative sums of the UNnormalised class probabilities (descriptor.py:113-139), so the
    # uniform draw is scaled by the class total; the reference draws in [0, 1) and can land on the invalid id 29
    old_func = torch.where(is_out, value.contiguous().view(torch.int32) & 0xFFFF, value.to(torch.int32))
    draws = []
    for rl in (d.roulette_ufuncs, d.roulette_bfuncs, d.roulette_tfuncs):
        total = rl[-1]
        u = torch.rand(shape, device=dev).reshape(-1) * total
        idx = torch.searchsorted(rl, u, right=True, out_int32=True).reshape(shape).clamp(max=rl.shape[0] - 1)
        draws.append(torch.where(total > 0, idx, old_func))
    func = torch.where(kind == NType.TFUNC, draws[2], torch.where(kind == NType.BFUNC, draws[1], draws[0]))
    if modify_output:
        out_idx = torch.randint(0, output_len, shape, dtype=torch.int32, device=dev)
    else:
        out_idx = torch.where(is_out, value.contiguous().view(torch.int32) >> 16, 0)
    packed = (func + (out_idx << 16)).to(torch.int32).view(torch.float32)
    func_val = torch.where(is_out, packed, func.to(torch.float32))
    var_val = torch.randint(0, input_len, shape, device=dev).to(torch.float32)
    const_val = d.const_samples[torch.randint(0, d.const_samples.shape[0], shape, device=dev)]
    return torch.where(kind == NType.CONST, const_val, torch.where(kind == NType.VAR, var_val, func_val))


class MultiPointMutation(BaseMutation):
    """Every node of a mutating tree is replaced, with probability ``mutation_intensity``, by a random node of its own
    kind (multi_point.py); the tree structure is unchanged."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, mutation_intensity: float = 0.3,
                 modify_output: bool = False):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor
        self.mutation_intensity = mutation_intensity
        self.modify_output = modify_output

    def _targets(self, forest: Forest) -> torch.Tensor:
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        return live & _mutate_mask(forest, self.mutation_rate)[:, None] & (torch.rand((n, L), device=dev) < self.mutation_intensity)

    def __call__(self, forest: Forest) -> Forest:
        fresh = _same_kind_values(forest.batch_node_type, forest.batch_node_value, self.descriptor, forest.input_len,
                                  forest.output_len, self.modify_output)
        value = torch.where(self._targets(forest), fresh, forest.batch_node_value)
        return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size)


class SinglePointMutation(MultiPointMutation):
    """One random node of a mutating tree is replaced by a random node of its own kind (single_point.py:43-126)."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, modify_output: bool = False):
        super().__init__(mutation_rate, descriptor, 1.0, modify_output)

    def _targets(self, forest: Forest) -> torch.Tensor:
        dev = forest.batch_node_value.device
        L = forest.max_tree_len
        p = _rand_below(forest.batch_subtree_size[:, 0])
        return (torch.arange(L, device=dev)[None, :] == p[:, None]) & _mutate_mask(forest, self.mutation_rate)[:, None]


class MultiConstMutation(BaseMutation):
    """Every constant of a mutating tree is redrawn from the descriptor's samples with probability
    ``mutation_intensity`` (multi_const.py)."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, mutation_intensity: float = 0.3):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor
        self.mutation_intensity = mutation_intensity

    def _targets(self, forest: Forest) -> torch.Tensor:
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        is_const = (forest.batch_node_type.to(torch.int64) & NType.TYPE_MASK) == NType.CONST
        return (live & is_const & _mutate_mask(forest, self.mutation_rate)[:, None]
                & (torch.rand((n, L), device=dev) < self.mutation_intensity))

    def __call__(self, forest: Forest) -> Forest:
        d = self.descriptor
        dev = forest.batch_node_value.device
        fresh = d.const_samples[torch.randint(0, d.const_samples.shape[0], forest.batch_node_value.shape, device=dev)]
        value = torch.where(self._targets(forest), fresh, forest.batch_node_value)
        return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size)


class SingleConstMutation(MultiConstMutation):
    """One random constant of a mutating tree is redrawn (single_const.py); trees without constants are unchanged."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        super().__init__(mutation_rate, descriptor, 1.0)

    def _targets(self, forest: Forest) -> torch.Tensor:
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        is_const = live & ((forest.batch_node_type.to(torch.int64) & NType.TYPE_MASK) == NType.CONST)
        score = torch.rand((n, L), device=dev) * is_const
        p = torch.argmax(score, dim=1)
        chosen = (torch.arange(L, device=dev)[None, :] == p[:, None]) & is_const
        return chosen & _mutate_mask(forest, self.mutation_rate)[:, None]


class CombinedMutation(BaseMutation):
    """Apply a list of mutation operators one after the other (combined.py)."""

    def __init__(self, mutation_operator):
        self.mutation_operator = list(mutation_operator)

    def __call__(self, forest: Forest) -> Forest:
        for op in self.mutation_operator:
            forest = op(forest)
        return forest


class CombinedDefaultMutation(BaseMutation):
    """DefaultMutation on every sub-forest of a CombinedForest, each with ``mutation_rate / number of sub-forests`` so that
    an individual is touched about as often as with a single forest (mutation/combined_default.py:9-51)."""

    def __init__(self, mutation_rate: float, descriptors):
        self.mutation_rate = mutation_rate
        self.descriptors = descriptors
        self._per_forest = None

    def __call__(self, combined_forest):
        from ..tree import CombinedForest

        n = len(combined_forest.forests)
        if self._per_forest is None:
            ds = [self.descriptors] * n if isinstance(self.descriptors, GenerateDescriptor) else list(self.descriptors)
            assert len(ds) == n, f"the length of descriptors should be {n}, but got {len(ds)}"
            self._per_forest = [DefaultMutation(self.mutation_rate / n, d) for d in ds]
        assert len(self._per_forest) == n, f"the pattern_num should be {len(self._per_forest)}, but got {n}"
        return CombinedForest([mut(f) for mut, f in zip(self._per_forest, combined_forest.forests)],
                              combined_forest.data_info)
