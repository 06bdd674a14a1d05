"""Mutation operators.  ``DefaultMutation`` follows src/evogp/algorithm/mutation/default.py:10-75:
each tree mutates with probability ``mutation_rate``; a mutating tree gets a freshly generated
random subtree (``tree_generate`` with the mutation descriptor) spliced in at a random position
(``tree_mutate``).  The Bernoulli mask is drawn on the DEVICE here (the reference draws it on the
CPU and uploads it, default.py:43 — SURVEY.md §8f N2)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from ..tree import MAX_STACK, Forest, GenerateDescriptor, NType


class BaseMutation:
    def __call__(self, forest: Forest):
        raise NotImplementedError


# ---- one launch per operator (csrc/mutate_ops.hip, round 5) -------------------------------------------------------------------
# On a device forest the reference's operators below DRAW and APPLY in one native launch: the random numbers are counter-based words of
# (the operator's seed, its call count), the distributions are those of the reference's draws.  `skip_rows`: the first rows of the forest
# are copied whatever is drawn -- GeneticProgramming.step hands over the whole next generation, elites first, instead of the offspring
# alone.  The torch programs (`draw` + `apply`) stay: they are what reproduces the reference bit for bit from ITS draws
# (tests/test_gpu_mutation_parity.py), the CPU path, and EVOGP_NATIVE_MUTATION=0.
def _native(forest: Forest, descriptor: GenerateDescriptor = None, same_shape: bool = False) -> bool:
    """the native launch takes the call: a device forest and, where the operator has a descriptor, its tables on the forest's device
    (a CPU descriptor goes through the torch program, which moves what it needs); `same_shape`: the descriptor generates
    rows of the forest's own width, inputs and outputs (InsertMutation's fresh trees are grafted row by row)"""
    if not forest.batch_node_value.is_cuda or os.environ.get("EVOGP_NATIVE_MUTATION", "1") == "0":
        return False
    if descriptor is not None:
        dev = forest.batch_node_value.device
        tables = (descriptor.const_samples, descriptor.roulette_funcs, descriptor.depth2leaf_probs, descriptor.roulette_ufuncs,
                  descriptor.roulette_bfuncs, descriptor.roulette_tfuncs)
        if any(t is not None and t.device != dev for t in tables) or descriptor.roulette_ufuncs is None:
            return False   # (a descriptor given a ready-made roulette has no per-arity tables: the torch programs)
        if same_shape and (descriptor.max_tree_len != forest.max_tree_len or descriptor.input_len != forest.input_len
                           or descriptor.output_len != forest.output_len):
            return False
    return True


def _next_call(op):
    """(seed, call): the seed once per operator object from torch's CPU generator (reproducible under torch.manual_seed, no device sync)"""
    if not hasattr(op, "_word_seed"):
        op._word_seed = int(torch.randint(0, 2**40, (1,)).item())
        op._calls = 0
    op._calls += 1
    return op._word_seed, op._calls


def _keep_rows(mask: torch.Tensor, skip_rows: int) -> torch.Tensor:
    if skip_rows > 0:
        mask = mask.clone()
        mask[:skip_rows] = False
    return mask


class DefaultMutation(BaseMutation):
    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor

    takes_skip_rows = True

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        dev = forest.batch_node_value.device
        mask = _keep_rows(torch.rand(forest.pop_size, device=dev) < self.mutation_rate, skip_rows)
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
# multi_point.py:46-143, single_const.py:39-98, multi_const.py:43-95).  Every structural one is a subtree replacement whose
# donor is a subtree of an existing tree — exactly what tree_crossover does — so Hoist and Delete are ONE native launch over
# the whole population with no gather program and no host sync: trees that do not mutate get left position -1, which the
# kernel answers with a verbatim copy (mutation.cu:256-266).  The point mutations are elementwise programs over the
# (pop, L) grid.
#
# Each operator is split into `draw` (the random numbers, on the device) and `apply` (what the operator does with them).
# `apply` takes the draws the REFERENCE operator makes, in the reference's meaning, as population-sized tensors (entries of
# trees that do not mutate are ignored): tests/golden/make_mutation_golden.py records the reference's own draws and results
# on the CPU, tests/test_gpu_mutation_parity.py feeds the draws to `apply` on the GPU and compares bit for bit.
# Where the reference's behaviour is a quirk rather than a choice, it is still the default and the alternative is an option:
#   * HoistMutation draws the inner subtree as an ABSOLUTE node index below size(outer) (hoist.py:58-68), not as an offset
#     inside the outer subtree (`inner_is_offset=True` for that);
#   * Multi{Point,Const}Mutation compare ONE uniform number per tree with the intensity (multi_point.py:70-75): either every
#     (constant) node of a mutating tree is redrawn or none (`per_node=True` draws one number per node);
#   * the per-arity roulettes are cumulative sums of probabilities normalised over ALL functions (descriptor.py:106-139), so a
#     uniform draw above the class total lands on the invalid id 29 (single_point.py:70-90).  `fix_roulette=True` scales
#     the draw by the class total instead.

def _uniform_int(u: torch.Tensor, low, high) -> torch.Tensor:
    """the reference's randint (tree/utils.py:306-310): trunc(low + u * (high - low)), computed in float32"""
    return (low + u * (high - low)).to(torch.int64)


def _rand(shape, dev):
    return torch.rand(shape, device=dev)


class HoistMutation(BaseMutation):
    """Pick a subtree, pick a subtree inside it, and put the inner one in the outer one's place (hoist.py:43-75)."""

    def __init__(self, mutation_rate: float, inner_is_offset: bool = False):
        self.mutation_rate = mutation_rate
        self.inner_is_offset = inner_is_offset

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        sizes = forest.batch_subtree_size.to(torch.int64)
        mask = _rand(forest.pop_size, dev) < self.mutation_rate
        p = _uniform_int(_rand(forest.pop_size, dev), 0, sizes[:, 0].to(torch.float32))
        inner = _uniform_int(_rand(forest.pop_size, dev), 0, sizes.gather(1, p[:, None]).squeeze(1).to(torch.float32))
        return mask, p, inner

    def apply(self, forest: Forest, mask: torch.Tensor, positions: torch.Tensor, subtree_positions: torch.Tensor) -> Forest:
        """positions: hoist.py:53-58 `mutate_positions`; subtree_positions: :61-68 (an absolute node index in the reference)"""
        dev = forest.batch_node_value.device
        p = positions.to(torch.int64)
        q = subtree_positions.to(torch.int64) + (p if self.inner_is_offset else 0)
        ar = torch.arange(forest.pop_size, dtype=torch.int32, device=dev)
        return forest.crossover(ar, ar, torch.where(mask, p, -1).to(torch.int32), q.to(torch.int32))

    takes_skip_rows = True

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        if _native(forest):
            seed, call = _next_call(self)
            v, t, s, _ = torch.ops.evogp_hip.structural_mutate(1, float(self.mutation_rate), 0, bool(self.inner_is_offset), int(skip_rows), seed, call,
                                                               *forest._tensors(), False)
            return Forest(forest.input_len, forest.output_len, v, t, s, func_mask=forest.func_mask)
        mask, p, inner = self.draw(forest)
        return self.apply(forest, _keep_rows(mask, skip_rows), p, inner)


class DeleteMutation(BaseMutation):
    """Pick a function node and replace it by one of its children (delete.py:44-105).  ``max_mutatable_size`` restricts
    the choice to nodes whose subtree is at most that large; as in the reference, a tree without an eligible node uses
    its root."""

    def __init__(self, mutation_rate: float, max_mutatable_size: Optional[int] = None):
        self.mutation_rate = mutation_rate
        self.max_mutatable_size = max_mutatable_size

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n, L = forest.batch_subtree_size.shape
        mask = (_rand(n, dev) < self.mutation_rate) & (forest.batch_subtree_size[:, 0] > 1)
        return mask, _rand((n, L), dev), _rand(n, dev)

    def apply(self, forest: Forest, mask: torch.Tensor, node_scores: torch.Tensor, child_u: torch.Tensor = None,
              nth_childs: torch.Tensor = None) -> Forest:
        """node_scores: the (pop, L) uniform numbers of `choose_nonleaf_pos` (delete.py:66-85); the child is either drawn from
        child_u as the reference does (:96-101) or given directly as nth_childs"""
        dev = forest.batch_node_value.device
        sizes = forest.batch_subtree_size.to(torch.int64)
        n, L = sizes.shape
        live = torch.arange(L, device=dev)[None, :] < sizes[:, :1]
        score = node_scores * live
        score = torch.where(sizes == 1, torch.zeros_like(score), score)
        if self.max_mutatable_size:
            score = torch.where(sizes > self.max_mutatable_size, torch.zeros_like(score), score)
        p = torch.argmax(score, dim=1)                                   # a random eligible node (0 if none)
        if nth_childs is None:
            arity = (forest.batch_node_type.gather(1, p[:, None]).squeeze(1).to(torch.int64) & NType.TYPE_MASK) - NType.UFUNC + 1
            nth_childs = _uniform_int(child_u, 1, arity.to(torch.float32))
        c1 = (p + 1).clamp(max=L - 1)
        c2 = (c1 + sizes.gather(1, c1[:, None]).squeeze(1)).clamp(max=L - 1)
        c3 = (c2 + sizes.gather(1, c2[:, None]).squeeze(1)).clamp(max=L - 1)
        q = torch.where(nth_childs == 3, c3, torch.where(nth_childs == 2, c2, c1))
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        return forest.crossover(ar, ar, torch.where(mask, p, -1).to(torch.int32), q.to(torch.int32))

    takes_skip_rows = True

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        if _native(forest):
            seed, call = _next_call(self)
            v, t, s, _ = torch.ops.evogp_hip.structural_mutate(0, float(self.mutation_rate), int(self.max_mutatable_size or 0), False, int(skip_rows), seed,
                                                               call, *forest._tensors(), False)
            return Forest(forest.input_len, forest.output_len, v, t, s, func_mask=forest.func_mask)
        mask, scores, child_u = self.draw(forest)
        return self.apply(forest, _keep_rows(mask, skip_rows), scores, child_u)


class InsertMutation(BaseMutation):
    """Pick a subtree, generate a small random tree, hang the subtree into a random position (>= 1) of the new tree and
    put the result where the subtree was (insert.py:45-85): trees grow by one random operator layer.  As in the reference
    the new trees are generated for the mutating trees only (tree index = rank among them), which costs one host sync."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n = forest.pop_size
        mask = _rand(n, dev) < self.mutation_rate
        p = _uniform_int(_rand(n, dev), 0, forest.batch_subtree_size[:, 0].to(torch.float32))
        keys = torch.randint(0, 1000000, (2,), device=dev).to(torch.uint32)
        return mask, p, keys, _rand(n, dev)

    def apply(self, forest: Forest, mask: torch.Tensor, positions: torch.Tensor, keys: torch.Tensor, new_u: torch.Tensor = None,
              new_positions: torch.Tensor = None) -> Forest:
        """positions: insert.py:57-62; keys: the two generation keys of :68-71; the position inside the new tree is drawn from
        new_u as the reference does (:74-79: low 1, high = its size) or given as new_positions — both indexed by tree"""
        idx = torch.nonzero(mask).squeeze(1)
        m = int(idx.shape[0])
        if m == 0:
            return forest
        sub = forest[idx]
        p = positions[idx].to(torch.int64)
        fresh = Forest.random_generate(pop_size=m, descriptor=self.descriptor, keys=keys)
        if new_positions is None:
            r = _uniform_int(new_u[idx], 1, fresh.batch_subtree_size[:, 0].to(torch.float32))
        else:
            r = new_positions[idx].to(torch.int64)
        ar = torch.arange(m, dtype=torch.int32, device=idx.device)
        both = fresh + sub                                               # rows [0, m): new trees, [m, 2m): the mutating trees
        grafted = both.crossover(ar, ar + m, r.to(torch.int32), p.to(torch.int32))   # subtree p of the old tree into position r of the new
        out = sub.mutate(p.to(torch.int32), grafted)
        res = Forest(forest.input_len, forest.output_len, forest.batch_node_value.clone(), forest.batch_node_type.clone(),
                     forest.batch_subtree_size.clone())
        res[idx] = out
        return res

    takes_skip_rows = True

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        d = self.descriptor
        if _native(forest, d, same_shape=True):
            # two launches: fresh trees for the rows whose word lies under the rate (the donor kernel of the fused default step), then
            # csrc/mutate_ops.hip insert_mutate_kernel under the same words -- no list of mutating trees, no host sync
            seed, call = _next_call(self)
            below = int(min(max(float(self.mutation_rate), 0.0), 1.0) * (2**31 - 1))
            fresh = torch.ops.evogp_hip.tree_generate_masked_hashed(forest.pop_size, forest.max_tree_len, d.input_len, d.output_len, int(d.const_samples.shape[0]),
                                                                    float(d.out_prob), float(d.const_prob), d.depth2leaf_probs, d.roulette_funcs,
                                                                    d.const_samples, 0, seed, call, below)
            v, t, s, _ = torch.ops.evogp_hip.insert_mutate(below, int(skip_rows), seed, call, *forest._tensors(), *fresh, False)
            return Forest(forest.input_len, forest.output_len, v, t, s, func_mask=Forest.join_masks(forest.func_mask, d.func_mask))
        mask, p, keys, u = self.draw(forest)
        return self.apply(forest, _keep_rows(mask, skip_rows), p, keys, u)


_warned_roulette = False


def _warn_unnormalised_roulette(d: GenerateDescriptor) -> None:
    """The reference searches the per-arity roulettes with u in [0, 1) although a class's entries only sum to the class's share
    of the function probabilities (single_point.py:70-90): every draw above that share writes the invalid function id 29 (it
    evaluates as 0 and has no name).  That behaviour is the default here because it is the reference's; say so once."""
    global _warned_roulette
    if _warned_roulette:
        return
    totals = [float(r[-1]) for r in (d.roulette_ufuncs, d.roulette_bfuncs, d.roulette_tfuncs)]
    if any(0.0 < t < 1.0 - 1e-6 for t in totals):
        import warnings

        _warned_roulette = True
        warnings.warn("point mutation with the reference's unnormalised per-arity roulettes (class totals "
                      f"{[round(t, 3) for t in totals]}): draws above a class total write the invalid function id 29, as upstream does "
                      "(single_point.py:70-90); pass fix_roulette=True to scale the draw by the class total instead", stacklevel=3)


def _roulette_pick(roulette: torch.Tensor, u: torch.Tensor, fix: bool, fallback: torch.Tensor) -> torch.Tensor:
    """function id for a uniform draw: the reference's searchsorted on the class roulette (single_point.py:70-84), or with
    `fix` the draw scaled by the class total (then never the invalid id 29; a class without functions keeps the old id)"""
    if not fix:
        return torch.searchsorted(roulette, u.contiguous(), out_int32=True)
    total = roulette[-1]
    idx = torch.searchsorted(roulette, (u * total).contiguous(), right=True, out_int32=True).clamp(max=roulette.shape[0] - 1)
    return torch.where(total > 0, idx, fallback)


def _same_kind_values(ntype: torch.Tensor, value: torch.Tensor, d: GenerateDescriptor, input_len: int, u_uf, u_bf, u_tf, var_idx, const_idx,
                      out_idx=None, fix_roulette: bool = False) -> torch.Tensor:
    """For every node a fresh payload of the node's own kind: a function of the same arity drawn from the descriptor's
    per-arity roulettes (an output node keeps, or with out_idx takes, its output index in the high half-word), a variable
    index, or a constant sample (single_point.py:64-124).  All arguments have the shape of `value`."""
    kind = ntype.to(torch.int64) & NType.TYPE_MASK
    is_out = (ntype.to(torch.int64) & NType.OUT_NODE) != 0
    bits = value.contiguous().view(torch.int32)
    old_func = torch.where(is_out, bits & 0xFFFF, value.to(torch.int32))
    uf = _roulette_pick(d.roulette_ufuncs, u_uf, fix_roulette, old_func)
    bf = _roulette_pick(d.roulette_bfuncs, u_bf, fix_roulette, old_func)
    tf = _roulette_pick(d.roulette_tfuncs, u_tf, fix_roulette, old_func)
    sel = (kind - NType.UFUNC).clamp(0, 2)                               # single_point.py:86-89: leaves read the unary draw (unused)
    func = torch.where(sel == 2, tf, torch.where(sel == 1, bf, uf))
    if out_idx is None:
        out_idx = torch.where(is_out, bits >> 16, torch.zeros_like(bits))
    packed = (func + (out_idx.to(torch.int32) << 16)).to(torch.int32).view(torch.float32)
    func_val = torch.where(is_out, packed, func.to(torch.float32))
    var_val = var_idx.to(torch.float32)
    const_val = d.const_samples[const_idx.to(torch.int64)]
    return torch.where(kind == NType.CONST, const_val, torch.where(kind == NType.VAR, var_val, func_val))


class MultiPointMutation(BaseMutation):
    """Nodes of a mutating tree are replaced by random nodes of their own kind (multi_point.py:46-143); the tree structure is
    unchanged.  Reference behaviour (default): one uniform number per TREE is compared with ``mutation_intensity``, so
    all nodes of a mutating tree are redrawn or none; ``per_node=True`` compares one number per node."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, mutation_intensity: float = 0.3,
                 modify_output: bool = False, per_node: bool = False, fix_roulette: bool = False):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor
        self.mutation_intensity = mutation_intensity
        self.modify_output = modify_output
        self.per_node = per_node
        self.fix_roulette = fix_roulette
        if not fix_roulette:
            _warn_unnormalised_roulette(descriptor)

    def _node_draws(self, forest: Forest):
        dev = forest.batch_node_value.device
        shape = forest.batch_node_value.shape
        d = self.descriptor
        out_idx = torch.randint(0, forest.output_len, shape, dtype=torch.int32, device=dev) if self.modify_output else None
        return dict(u_uf=_rand(shape, dev), u_bf=_rand(shape, dev), u_tf=_rand(shape, dev),
                    var_idx=_uniform_int(_rand(shape, dev), 0, forest.input_len),
                    const_idx=_uniform_int(_rand(shape, dev), 0, d.const_samples.shape[0]), out_idx=out_idx)

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        mask = _rand(n, dev) < self.mutation_rate
        intensity_u = _rand((n, L) if self.per_node else (n, 1), dev)
        return mask, intensity_u, self._node_draws(forest)

    def targets(self, forest: Forest, mask: torch.Tensor, intensity_u: torch.Tensor) -> torch.Tensor:
        dev = forest.batch_node_value.device
        L = forest.max_tree_len
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        return live & mask[:, None] & (intensity_u < self.mutation_intensity)

    def apply(self, forest: Forest, targets: torch.Tensor, node_draws: dict) -> Forest:
        """targets: (pop, L) bool; node_draws: per-node tensors u_uf, u_bf, u_tf (uniform), var_idx, const_idx and optionally out_idx"""
        fresh = _same_kind_values(forest.batch_node_type, forest.batch_node_value, self.descriptor, forest.input_len,
                                  fix_roulette=self.fix_roulette, **node_draws)
        value = torch.where(targets, fresh, forest.batch_node_value)
        return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size)

    takes_skip_rows = True
    _native_mode = 0   # csrc/mutate_ops.hip point_mutate_kernel: 0 multi-point, 1 single-point

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        d = self.descriptor
        if _native(forest, d):
            seed, call = _next_call(self)
            value = torch.ops.evogp_hip.point_mutate(self._native_mode, float(self.mutation_rate), float(self.mutation_intensity), bool(self.per_node),
                                                     bool(self.modify_output), bool(self.fix_roulette), int(skip_rows), forest.input_len, forest.output_len,
                                                     seed, call, *forest._tensors(), d.roulette_ufuncs, d.roulette_bfuncs, d.roulette_tfuncs, d.const_samples)
            # (without fix_roulette a draw can write the invalid function id 29, single_point.py:70-90: the function set is no longer known)
            mask = Forest.join_masks(forest.func_mask, d.func_mask) if self.fix_roulette else 0
            return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size, func_mask=mask)
        mask, second, node_draws = self.draw(forest)
        return self.apply(forest, self.targets(forest, _keep_rows(mask, skip_rows), second), node_draws)


class SinglePointMutation(MultiPointMutation):
    """One random node of a mutating tree is replaced by a random node of its own kind (single_point.py:43-126)."""

    _native_mode = 1

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, modify_output: bool = False, fix_roulette: bool = False):
        super().__init__(mutation_rate, descriptor, 1.0, modify_output, fix_roulette=fix_roulette)

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n = forest.pop_size
        mask = _rand(n, dev) < self.mutation_rate
        p = _uniform_int(_rand(n, dev), 0, forest.batch_subtree_size[:, 0].to(torch.float32))
        return mask, p, self._node_draws(forest)

    def targets(self, forest: Forest, mask: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        dev = forest.batch_node_value.device
        return (torch.arange(forest.max_tree_len, device=dev)[None, :] == positions.to(torch.int64)[:, None]) & mask[:, None]


class MultiConstMutation(BaseMutation):
    """Constants of a mutating tree are redrawn from the descriptor's samples (multi_const.py:43-95).  Reference behaviour
    (default): one uniform number per TREE against ``mutation_intensity``; ``per_node=True``: one per node."""

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor, mutation_intensity: float = 0.3, per_node: bool = False):
        self.mutation_rate = mutation_rate
        self.descriptor = descriptor
        self.mutation_intensity = mutation_intensity
        self.per_node = per_node

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        mask = _rand(n, dev) < self.mutation_rate
        const_idx = torch.randint(0, self.descriptor.const_samples.shape[0], (n, L), device=dev)
        return mask, _rand((n, L) if self.per_node else (n, 1), dev), const_idx

    def targets(self, forest: Forest, mask: torch.Tensor, intensity_u: torch.Tensor) -> torch.Tensor:
        dev = forest.batch_node_value.device
        L = forest.max_tree_len
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        is_const = forest.batch_node_type == NType.CONST                  # multi_const.py:73: the raw type
        return live & is_const & mask[:, None] & (intensity_u < self.mutation_intensity)

    def apply(self, forest: Forest, targets: torch.Tensor, const_idx: torch.Tensor) -> Forest:
        fresh = self.descriptor.const_samples[const_idx.to(torch.int64)]
        value = torch.where(targets, fresh, forest.batch_node_value)
        return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size)

    takes_skip_rows = True
    _native_mode = 2   # csrc/mutate_ops.hip point_mutate_kernel: 2 multi-const, 3 single-const

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        d = self.descriptor
        if _native(forest, d):   # (the kernel's argument list wants the roulettes; the constant modes do not read them)
            seed, call = _next_call(self)
            value = torch.ops.evogp_hip.point_mutate(self._native_mode, float(self.mutation_rate), float(self.mutation_intensity), bool(self.per_node),
                                                     False, False, int(skip_rows), forest.input_len, forest.output_len, seed, call,
                                                     *forest._tensors(), d.roulette_ufuncs, d.roulette_bfuncs, d.roulette_tfuncs, d.const_samples)
            return Forest(forest.input_len, forest.output_len, value, forest.batch_node_type, forest.batch_subtree_size, func_mask=forest.func_mask)
        mask, u, const_idx = self.draw(forest)
        return self.apply(forest, self.targets(forest, _keep_rows(mask, skip_rows), u), const_idx)


class SingleConstMutation(MultiConstMutation):
    """One random constant of a mutating tree is redrawn (single_const.py:39-98); trees without constants are unchanged."""

    _native_mode = 3

    def __init__(self, mutation_rate: float, descriptor: GenerateDescriptor):
        super().__init__(mutation_rate, descriptor, 1.0)

    def draw(self, forest: Forest):
        dev = forest.batch_node_value.device
        n, L = forest.batch_node_value.shape
        mask = _rand(n, dev) < self.mutation_rate
        const_idx = torch.randint(0, self.descriptor.const_samples.shape[0], (n, 1), device=dev).expand(n, L)
        return mask, _rand((n, L), dev), const_idx

    def targets(self, forest: Forest, mask: torch.Tensor, node_scores: torch.Tensor) -> torch.Tensor:
        """node_scores: the (pop, L) uniform numbers of `choose_constant_pos` (single_const.py:55-72)"""
        dev = forest.batch_node_value.device
        L = forest.max_tree_len
        live = torch.arange(L, device=dev)[None, :] < forest.batch_subtree_size[:, :1]
        is_const = forest.batch_node_type == NType.CONST
        score = torch.where(is_const, node_scores * live, torch.zeros_like(node_scores))
        p = torch.argmax(score, dim=1)
        return (torch.arange(L, device=dev)[None, :] == p[:, None]) & is_const & mask[:, None]   # :92-97: only if it IS a constant


class CombinedMutation(BaseMutation):
    """Apply a list of mutation operators one after the other (combined.py)."""

    def __init__(self, mutation_operator):
        self.mutation_operator = list(mutation_operator)

    @property
    def takes_skip_rows(self) -> bool:
        return all(getattr(op, "takes_skip_rows", False) for op in self.mutation_operator)

    def __call__(self, forest: Forest, skip_rows: int = 0) -> Forest:
        for op in self.mutation_operator:
            forest = op(forest, skip_rows=skip_rows) if skip_rows else op(forest)
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
