"""Population sharding over the GPUs of one node — one process per GPU, RCCL over xGMI.

The reference is single-GPU (no collective anywhere); this module adds the
data-parallel axis the workload has for free:

* fitness evaluation, tree generation and mutation are independent per tree  -> every rank works
  on its own contiguous block of ``pop / G`` trees, no communication;
* selection ranks the WHOLE population and crossover draws parents from the global survivor set
  -> two collectives per generation: an all-gather of the local FITNESS values (4 B per tree), after which every rank
  knows the global ranking, and an all-gather of the rows of the trees that can still be read — the survivors and elites,
  ``max(survival_rate, elite_rate) * pop`` of them — packed as ``uint8[cap x 8 L]`` = {value, type, size} per rank
  (``cap`` = the largest number of kept trees on any rank).  At pop = 1 M, L = 64 and 30 % survivors that is ~20 MB per
  rank instead of the 64.5 MB of the whole shard.  On the xGMI mesh every rank sends its block to its 7 peers over 7
  direct links, so one large collective per array kind is the right granularity.

Nothing in a rank's step is O(global population) except streaming passes over the gathered fitness vector: the elite and
survivor SETS come from an exact radix select in one launch (``select_order`` / csrc/select.hip; nothing is sorted), a kept tree's
row in the gathered table is index arithmetic on the kept mask, and the six random words of an offspring are a counter-based hash of
(seed, generation, word, GLOBAL offspring index) (``random_words`` / ``evogp_hip_random_words``): a rank computes exactly the
words of its own rows, and all ranks agree on them whatever the world size.

After the exchange every rank holds the identical table of parents and the identical ranking, then materialises ONLY its
own slice ``[r*pop/G, (r+1)*pop/G)`` of the next generation: with the fused breeding pass (``evogp_hip_generate_masked`` +
``evogp_hip_breed_default_rows``) on a GPU, or with ``tree_crossover`` / ``tree_generate`` (tree-index offset = global
mutation rank) / ``tree_mutate`` and a generator seeded identically on all ranks otherwise.  By construction the union of
the shards is bit-identical for every world size, G = 1 included (tests/test_sharded_gloo.py checks G = 2 against G = 1 on
the gloo backend; tests/test_gpu_breed.py builds the shards of G = 1, 2, 3, 8 on one GPU; tests/test_gpu_rccl.py runs two
ranks over RCCL where two GPUs exist).

Operator semantics are those of DefaultSelection / DefaultCrossover / DefaultMutation
(src/evogp/algorithm/{selection,crossover,mutation}/default.py): same distributions, drawn from an
explicit generator so that ranks agree.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from .algorithm.selection import DefaultSelection
from .tree import MAX_STACK, Forest, GenerateDescriptor


def _pack(forest: Forest, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8[n x 8 L]: value | type | size of the given rows (all rows when ``rows`` is None)"""
    v, t, s = forest.batch_node_value, forest.batch_node_type, forest.batch_subtree_size
    if rows is not None:
        v, t, s = v[rows], t[rows], s[rows]
    n = v.shape[0]
    parts = [v.contiguous().view(torch.uint8).view(n, -1), t.contiguous().view(torch.uint8).view(n, -1),
             s.contiguous().view(torch.uint8).view(n, -1)]
    return torch.cat(parts, dim=1).contiguous()


def _unpack(buf: torch.Tensor, L: int, input_len: int, output_len: int) -> Forest:
    n = buf.shape[0]
    value = buf[:, : 4 * L].contiguous().view(torch.float32).view(n, L)
    ntype = buf[:, 4 * L: 6 * L].contiguous().view(torch.int16).view(n, L)
    size = buf[:, 6 * L: 8 * L].contiguous().view(torch.int16).view(n, L)
    return Forest(input_len, output_len, value, ntype, size)


def select_order(fit: torch.Tensor, n_elite: int, n_keep: int) -> torch.Tensor:
    """int32[n_keep]: the n_elite best trees, then the other survivors, each group in ascending tree index; ties at a threshold go
    to the lower index, i.e. the two SETS are those of a stable descending sort.  Nothing downstream uses an order inside the sets
    (elites are copied, parents are drawn uniformly), so no sort of the population is needed: on a GPU this is ONE launch of an
    exact radix select (csrc/select.hip: 25 us at 100 k values, 40 us at 1 M; torch.sort takes 61 / 219 us, torch.kthvalue 0.39 /
    3.8 ms); elsewhere the same result from torch ops.  Every rank runs it on the identical gathered vector."""
    n_elite = min(n_elite, n_keep)
    if fit.is_cuda and fit.dtype == torch.float32 and os.environ.get("EVOGP_NATIVE_SELECT", "1") != "0":
        return torch.ops.evogp_hip.select_survivors(fit.contiguous(), n_elite, n_keep)
    best = torch.sort(fit, descending=True, stable=True).indices[:n_keep]
    return torch.cat([torch.sort(best[:n_elite]).values, torch.sort(best[n_elite:]).values]).to(torch.int32)


def plan_exchange(fit_all: torch.Tensor, n_elite: int, n_keep: int, world: int):
    """From the gathered fitness of the whole population (rank-major): which trees are kept, how many rows every rank contributes,
    and where each kept tree lands in the gathered table.  -> (per_rank bool[world][n_local], cap, order int32[n_keep] of TABLE rows,
    elites first).  ``cap`` -- the largest number of kept trees on any rank, the row count every rank pads its block to -- is the
    step's one host sync (one integer); a bound that needs none would have to be min(n_local, n_keep), i.e. gather the whole
    population at world sizes where n_keep > n_local."""
    pop = fit_all.shape[0]
    chosen = select_order(fit_all, n_elite, n_keep).long()
    kept = torch.zeros(pop, dtype=torch.bool, device=fit_all.device)
    kept[chosen] = True
    per_rank = kept.view(world, pop // world)
    cap = int(per_rank.sum(1).max())
    # a rank sends its kept trees in ascending local index (kept_rows): tree g of rank r is row r * cap + (kept trees of r below g)
    below = torch.cumsum(per_rank.to(torch.int64), dim=1) - 1
    table_row = (below + torch.arange(world, device=fit_all.device)[:, None] * cap).view(-1)
    return per_rank, cap, table_row[chosen].to(torch.int32).contiguous()


def kept_rows(mine: torch.Tensor, cap: int) -> torch.Tensor:
    """local indices of this rank's kept trees in ascending order, padded with other local trees up to ``cap`` rows"""
    return torch.argsort((~mine).to(torch.int8), stable=True)[:cap]


# ---- counter-based random words ------------------------------------------------------------------------------------------
# Every rank needs the six random words of ITS offspring only, and all ranks must agree on the words of offspring i whatever the
# world size.  A generator with a running state forces every rank to draw the words of the whole generation (24 MB per rank and
# generation at 1 M trees); here word k of offspring i of generation g is a hash of (seed, g, k, i) — the splitmix64 finaliser
# in wrap-around int64 arithmetic — so a rank computes exactly its slice.
_M1, _M2 = -4658895280553007687, -7723592293110705685    # 0xBF58476D1CE4E5B9, 0x94D049BB133111EB as signed 64-bit


def _mix64(x: torch.Tensor) -> torch.Tensor:
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * _M1
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * _M2
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def random_words(seed: int, generation: int, rows: int, lo: int, hi: int, device) -> torch.Tensor:
    """int32[rows][hi - lo]: word k of offspring i in [lo, hi), uniform in [0, 2^31 - 1) like torch.randint(0, 2^31 - 1)"""
    i = torch.arange(lo, hi, dtype=torch.int64, device=device)[None, :]
    k = torch.arange(rows, dtype=torch.int64, device=device)[:, None]
    base = _mix64(torch.tensor([seed * 1000003 + generation], dtype=torch.int64, device=device))
    x = _mix64(base + (k << 40) + i)
    return (((x >> 33) & 0x7FFFFFFF) % (2**31 - 1)).to(torch.int32)


class ShardedGeneticProgramming:
    """GeneticProgramming over a population sharded across the ranks of ``group``.

    ``local_forest`` is this rank's block of the population (equal sizes on all ranks).  ``step``
    takes the fitness of the LOCAL trees and returns the local block of the next generation.
    """

    def __init__(self, local_forest: Forest, mutation_rate: float, mutation_descriptor: GenerateDescriptor,
                 selection: Optional[DefaultSelection] = None, seed: int = 0, group=None):
        self.forest = local_forest
        self.mutation_rate = mutation_rate
        self.descriptor = mutation_descriptor
        self.selection = selection or DefaultSelection(survival_rate=0.3, elite_rate=0.01)
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_local = local_forest.pop_size
        self.pop_size = self.n_local * self.world
        dev = local_forest.batch_node_value.device
        self.seed = seed
        self.generation = 0
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)  # identical on every rank: the draws of the torch composition (slice_torch) agree

    # -- the exchange step -------------------------------------------------------------------------
    def exchange(self, local_fitness: torch.Tensor):
        """-> (table Forest, order int32[n_keep] of table rows: the elites, then the other survivors, global pop)"""
        n_elite, n_surv = self.selection.counts(self.pop_size)
        n_keep = max(n_elite, n_surv, 1)
        f = self.forest
        fit = local_fitness.to(torch.float32).contiguous()
        if self.world == 1:
            return f, select_order(fit, n_elite, n_keep), self.pop_size
        fit_all = torch.empty(self.pop_size, dtype=torch.float32, device=fit.device)
        dist.all_gather_into_tensor(fit_all, fit, group=self.group)
        per_rank, cap, order = plan_exchange(fit_all, n_elite, n_keep, self.world)
        send = _pack(f, kept_rows(per_rank[self.rank], cap))
        table = torch.empty((self.world * cap, send.shape[1]), dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(table, send, group=self.group)
        return _unpack(table, f.max_tree_len, f.input_len, f.output_len), order, self.pop_size

    def step(self, local_fitness: torch.Tensor) -> Forest:
        assert local_fitness.shape == (self.n_local,)
        table, order, pop = self.exchange(local_fitness)
        lo, hi = self.rank * self.n_local, (self.rank + 1) * self.n_local
        if local_fitness.is_cuda and self.descriptor.max_tree_len == table.max_tree_len \
                and os.environ.get("EVOGP_NATIVE_STEP", "1") != "0":
            self.forest = self.slice_native(table, order, pop, lo, hi)
        else:
            self.forest = self.slice_torch(table, order, pop, lo, hi)
        self.generation += 1
        return self.forest

    def _order_of(self, full: Forest, fitness: torch.Tensor) -> torch.Tensor:
        n_elite, n_surv = self.selection.counts(full.pop_size)
        return select_order(fitness.to(torch.float32), n_elite, max(n_elite, n_surv, 1))

    def next_slice_native(self, full: Forest, fitness: torch.Tensor, lo: int, hi: int) -> Forest:
        """rows [lo, hi) of the next generation from the WHOLE population and its fitness (tests, single-table use)"""
        return self.slice_native(full, self._order_of(full, fitness), full.pop_size, lo, hi)

    def next_slice_torch(self, full: Forest, fitness: torch.Tensor, lo: int, hi: int) -> Forest:
        return self.slice_torch(full, self._order_of(full, fitness), full.pop_size, lo, hi)

    def slice_native(self, table: Forest, order: torch.Tensor, pop: int, lo: int, hi: int) -> Forest:
        """Rows [lo, hi) of the next generation with the fused breeding pass (csrc/breed.hip).  ``table`` holds the trees
        that can be parents or elites, ``order`` ranks them (table rows, best first), ``pop`` is the size of the WHOLE
        population.  Every rank draws the SAME six random words per offspring and the same generation keys from its
        generator, generates donors only for its own mutating offspring (tree index = global offspring index) and builds
        only its own rows.  The union over the ranks is the single-device result for the same generator state."""
        full = table
        dev = order.device
        L = full.max_tree_len
        n_elite, n_surv = self.selection.counts(pop)
        n_new = pop - n_elite
        below = int(min(max(self.mutation_rate, 0.0), 1.0) * (2**31 - 1))
        d = self.descriptor
        rows = hi - lo
        o_lo, o_hi = max(lo, n_elite) - n_elite, max(hi, n_elite) - n_elite   # my offspring indices
        # the words of MY offspring only (the breeding pass indexes the array by offspring number: the other columns are
        # never read and stay uninitialised), and this generation's two generation keys
        rnd = torch.ops.evogp_hip.random_words(self.seed, self.generation, 6, n_new, o_lo, o_hi, dev)   # one launch, this rank's columns
        keys = (torch.ops.evogp_hip.random_words(self.seed, self.generation, 8, 2, 0, 2, dev)[7] % 1000000).to(torch.uint32)   # row 7: not an offspring word
        if o_hi > o_lo:
            donors = torch.ops.evogp_hip.tree_generate_masked(
                o_hi - o_lo, L, d.input_len, d.output_len, d.const_samples.shape[0], d.out_prob, d.const_prob, keys,
                d.depth2leaf_probs, d.roulette_funcs, d.const_samples, o_lo, rnd[4, o_lo:o_hi].contiguous(), below)
            # donors cover my offspring rows only; breed_default_rows skips the elite rows at the head of the range
        else:  # a slice of elites only: donors are never read
            donors = (torch.empty((rows, L), dtype=torch.float32, device=dev), torch.empty((rows, L), dtype=torch.int16, device=dev),
                      torch.empty((rows, L), dtype=torch.int16, device=dev))
        value, ntype, size = full._tensors()
        nv, nt, ns = torch.ops.evogp_hip.breed_default_rows(pop, L, n_elite, n_surv, value, ntype, size, order, rnd, below,
                                                            *donors, lo, rows)
        return Forest(full.input_len, full.output_len, nv, nt, ns)

    def slice_torch(self, table: Forest, order: torch.Tensor, pop: int, lo: int, hi: int) -> Forest:
        """The same slice composed from the reference's operators (any device)."""
        full = table
        dev = order.device
        n_elite, n_surv = self.selection.counts(pop)
        elite_idx, surv_idx = order[:n_elite], order[:n_surv]     # identical on every rank
        target = pop - n_elite
        parents = full[surv_idx.to(torch.int64)]
        sizes = parents.batch_subtree_size[:, 0].to(torch.int64)
        n_par = len(parents)

        # draws for the WHOLE next generation, identical on every rank (generator seeded identically)
        g = self.gen
        pair = torch.randint(0, n_par, (2, target), generator=g, device=dev)
        raw = torch.randint(0, torch.iinfo(torch.int32).max, (2, target), generator=g, device=dev)
        left, right = pair[0], pair[1]
        left_pos = raw[0] % sizes[left]
        right_pos = raw[1] % sizes[right]
        mut_mask = torch.rand(target, generator=g, device=dev) < self.mutation_rate
        mut_raw = torch.randint(0, MAX_STACK, (target,), generator=g, device=dev)
        keys = torch.randint(0, 1000000, (2,), generator=g, device=dev).to(torch.uint32)

        # this rank's slots of the next generation: [lo, hi) of [elites | offspring]
        e_lo, e_hi = min(lo, n_elite), min(hi, n_elite)
        o_lo, o_hi = max(lo, n_elite) - n_elite, max(hi, n_elite) - n_elite
        pieces = []
        if e_hi > e_lo:
            pieces.append(full[elite_idx[e_lo:e_hi].to(torch.int64)])
        if o_hi > o_lo:
            sl = slice(o_lo, o_hi)
            i32 = lambda t: t.to(torch.int32).contiguous()  # noqa: E731
            child = parents.crossover(i32(left[sl]), i32(right[sl]), i32(left_pos[sl]), i32(right_pos[sl]))
            m = mut_mask[sl]
            n_mut = int(m.sum())
            if n_mut > 0:
                before = int(mut_mask[:o_lo].sum())  # global rank of my first mutated offspring
                donors = Forest.random_generate(n_mut, self.descriptor, keys=keys, tree_index_offset=before)
                chosen = child[m]
                pos = (mut_raw[sl][m] % chosen.batch_subtree_size[:, 0].to(torch.int64)).to(torch.int32)
                child[m] = chosen.mutate(pos, donors)
            pieces.append(child)
        nxt = pieces[0]
        for p in pieces[1:]:
            nxt = nxt + p
        return nxt
