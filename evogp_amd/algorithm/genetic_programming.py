"""GeneticProgramming — one generation = selection -> crossover -> mutation -> elites + offspring
(src/evogp/algorithm/genetic_programming.py:30-124).  The optional Pareto front by tree size is
kept (``enable_pareto_front``); it is torch-level bookkeeping."""
from __future__ import annotations

import os

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
        plan = self._native_plan()
        if plan is not None:
            first, rest = plan
            nxt = self._native_default_step(fitness, first)
            if nxt is not None:
                # the other mutation operators of the list, one launch each over the offspring rows (csrc/mutate_ops.hip): the elites
                # -- the first rows -- are copied (genetic_programming.py:118-122: only the offspring mutate)
                for op in rest:
                    self.forest = op(self.forest, skip_rows=self._last_n_elite)
                return self.forest
            # (a selection whose lists the fused pass cannot take -- nothing but elites, no parents: the composed operators below)
        lists = self.__dict__.pop("_replay_lists", None)   # (the selection was already drawn by the fused pass that handed back)
        elite_indices, survivor_indices = lists if lists is not None else self.selection(self.forest, fitness)
        offspring = self.crossover(forest=self.forest, survivor_indices=survivor_indices,
                                   target_cnt=self.pop_size - elite_indices.shape[0], fitness=fitness)
        offspring = self.mutation(offspring)
        self.forest = self.forest[elite_indices] + offspring  # elites first (genetic_programming.py:122)
        return self.forest

    # ---- fused default step ------------------------------------------------------------
    def _native_plan(self):
        """DefaultCrossover on a device forest, under ANY selection operator, with DefaultMutation and / or the reference's other
        mutation operators (alone or as a CombinedMutation list, e.g. example/brax_task.py:38-45): one selection (a single launch for
        DefaultSelection; the operator's own torch program otherwise -- its survivor list may repeat trees, selection/tournament.py:
        59-133), masked donor generation and ONE breeding pass instead of ~90 small launches and two host syncs, then one launch per
        further mutation operator (same distributions; the random words are counter-based).  Returns (the DefaultMutation that leads the
        list or None, the operators behind it), or None where the composed torch programs have to run.  EVOGP_NATIVE_STEP=0 disables it."""
        from .crossover import DefaultCrossover
        from .mutation import CombinedMutation, DefaultMutation
        from .selection import DefaultSelection

        if os.environ.get("EVOGP_NATIVE_STEP", "1") == "0" or type(self.crossover) is not DefaultCrossover:
            return None
        f = self.forest
        if not f.batch_node_value.is_cuda:
            return None
        ops = list(self.mutation.mutation_operator) if type(self.mutation) is CombinedMutation else [self.mutation]
        first = ops[0] if ops and type(ops[0]) is DefaultMutation else None
        rest = ops[1:] if first is not None else ops
        if first is not None and first.descriptor.max_tree_len != f.max_tree_len:
            return None
        from . import mutation as _m

        built_in = (_m.DefaultMutation, _m.HoistMutation, _m.DeleteMutation, _m.InsertMutation, _m.MultiPointMutation, _m.SinglePointMutation,
                    _m.MultiConstMutation, _m.SingleConstMutation)
        if any(type(op) not in built_in for op in rest):
            # an operator of the caller's own -- also a SUBCLASS of a built-in one, whose __call__ may have the reference's signature
            # (forest) and would not take skip_rows: it sees the offspring forest alone, as the reference hands it over
            return None
        if rest and os.environ.get("EVOGP_NATIVE_MUTATION", "1") == "0":
            return None
        if type(self.selection) is DefaultSelection:
            n_elite, n_surv = self.selection.counts(f.pop_size)
            if not (0 <= n_elite < f.pop_size and 0 < n_surv <= f.pop_size):
                return None
        return first, rest

    def _native_default_step(self, fitness: torch.Tensor, mutation=None) -> Forest:
        """selection + DefaultCrossover + `mutation` (a DefaultMutation, or None: no offspring mutates here) in one breeding pass"""
        f = self.forest
        dev = f.batch_node_value.device
        pop, L = f.pop_size, f.max_tree_len
        # DefaultSelection: elites first, then the other survivors (each group by tree index) from ONE launch instead of a sort of
        # the whole vector -- nothing downstream uses the order inside the two sets (csrc/select.hip; parallel.default_lists also
        # covers more elites than parents).  Any other operator: its own two lists.  (Drawing the words and generating the donors
        # on a second stream meanwhile was measured twice: the stream hand-over costs more than the 25 us it hides.)
        from ..parallel import default_lists
        from .selection import DefaultSelection

        if not fitness.is_cuda or fitness.device != dev:
            return None   # (a fitness vector on another device: the operators' own torch programs deal with it)
        mark = getattr(self, "stage_marker", None) or (lambda name: None)   # (bench.py: an event behind every stage of the step)
        if type(self.selection) is DefaultSelection:
            if fitness.dtype != torch.float32:
                return None   # (a cast could merge ties of a float64 fitness: the selection operator ranks what it was given)
            elites, parents = default_lists(fitness, *self.selection.counts(pop))
        else:
            lists = None
            counter_based = getattr(self.selection, "counter_based", None)
            if counter_based is not None and os.environ.get("EVOGP_NATIVE_TOURNAMENT", "1") != "0":
                # (TournamentSelection with its default arguments: two launches; contenders from the counter-based words of this step)
                if not hasattr(self, "_word_seed"):
                    self._word_seed = int(torch.randint(0, 2**40, (1,)).item())
                lists = counter_based(fitness, self._word_seed, getattr(self, "_steps", 0) + 1)
            elites, parents = lists if lists is not None else self.selection(f, fitness)
            # the same guards DefaultSelection's counts get in _native_plan: at least one offspring row, at least one parent, lists
            # on the forest's device
            if elites.numel() >= pop or parents.numel() == 0:
                self._replay_lists = (elites, parents)
                return None
            elites, parents = elites.to(device=dev, dtype=torch.int32).contiguous(), parents.to(device=dev, dtype=torch.int32).contiguous()
        mark("select")
        n_elite = elites.numel()
        n_new = pop - n_elite
        # no draw at all: the donor kernel and the breeding pass compute the six words of offspring i (and the two generation keys) as
        # hash(seed, step, word, i) themselves (csrc/evogp_defs.hpp counter_word).  The seed is drawn ONCE per object from torch's
        # CPU generator (reproducible under torch.manual_seed, independent between objects, no device sync); the reference draws
        # seven tensors per generation (crossover/default.py:40-58, mutation/default.py:43-66, tree/forest.py:51-57)
        if not hasattr(self, "_word_seed"):
            self._word_seed = int(torch.randint(0, 2**40, (1,)).item())
        self._steps = getattr(self, "_steps", 0) + 1
        self._last_n_elite = n_elite
        value, ntype, size = f._tensors()
        if mutation is not None:
            below = int(min(max(mutation.mutation_rate, 0.0), 1.0) * (2**31 - 1))
            d = mutation.descriptor
            donors = torch.ops.evogp_hip.tree_generate_masked_hashed(
                n_new, L, d.input_len, d.output_len, d.const_samples.shape[0], d.out_prob, d.const_prob,
                d.depth2leaf_probs, d.roulette_funcs, d.const_samples, 0, self._word_seed, self._steps, below)
            mask = Forest.join_masks(f.func_mask, d.func_mask)
        else:   # no offspring mutates in the pass: the donor rows are never read
            below = 0
            donors = (torch.empty((n_new, L), dtype=torch.float32, device=dev), torch.empty((n_new, L), dtype=torch.int16, device=dev),
                      torch.empty((n_new, L), dtype=torch.int16, device=dev))
            mask = f.func_mask
        mark("donors")
        nv, nt, ns = torch.ops.evogp_hip.breed_rows_hashed(pop, L, value, ntype, size, elites, parents, self._word_seed, self._steps,
                                                           below, *donors, 0, pop)
        mark("breeding")
        self.forest = Forest(f.input_len, f.output_len, nv, nt, ns, func_mask=mask)
        return self.forest
