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

from .algorithm.selection import BaseSelection, DefaultSelection
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


def _select_key(fit: torch.Tensor) -> torch.Tensor:
    """the order-preserving integer key of csrc/select.hip (select_key) as int64: larger fitness <=> larger key, NaN -> 0 (worse
    than -inf), -0 and +0 share a key"""
    f = fit.to(torch.float32)
    u = f.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = torch.where(f == 0, torch.zeros_like(u), u)
    key = torch.where(u >= 0x80000000, 0xFFFFFFFF - u, u + 0x80000000)
    return torch.where(f != f, torch.zeros_like(key), key)


def select_order(fit: torch.Tensor, n_elite: int, n_keep: int) -> torch.Tensor:
    """int32[n_keep]: the n_elite best trees, then the other survivors, each group in ascending tree index; ties at a threshold go
    to the lower index, i.e. the two SETS are those of a stable descending sort (NaN ranks worst, -0 = +0).  Nothing downstream uses
    an order inside the sets (elites are copied, parents are drawn uniformly), so no sort of the population is needed: on a GPU this
    is ONE launch of an exact radix select (csrc/select.hip: 25 us at 100 k values, 40 us at 1 M; torch.sort takes 61 / 219 us,
    torch.kthvalue 0.39 / 3.8 ms); elsewhere the same result from torch ops.  Every rank runs it on the identical gathered vector."""
    n_elite = min(n_elite, n_keep)
    if fit.is_cuda and fit.dtype == torch.float32 and os.environ.get("EVOGP_NATIVE_SELECT", "1") != "0":
        return torch.ops.evogp_hip.select_survivors(fit.contiguous(), n_elite, n_keep)
    best = torch.sort(_select_key(fit), descending=True, stable=True).indices[:n_keep]
    return torch.cat([torch.sort(best[:n_elite]).values, torch.sort(best[n_elite:]).values]).to(torch.int32)


def default_lists(fit: torch.Tensor, n_elite: int, n_surv: int):
    """DefaultSelection (selection/default.py:42-71) without a sort: -> (elites int32[n_elite], parents int32[n_surv]), both
    prefixes of ONE select_order call.  The smaller set is selected first, so the prefixes are right whichever is larger
    (DefaultSelection(survival_rate=0.3, elite_cnt=large) is legal: the elites are then the n_elite best, the parents the n_surv
    best of them)."""
    order = select_order(fit, min(n_elite, n_surv), max(n_elite, n_surv, 1))
    return order[:n_elite], order[:n_surv]


def plan_exchange(elites: torch.Tensor, parents: torch.Tensor, pop: int, world: int, cap: Optional[int] = None):
    """From the selection's two lists of GLOBAL tree indices (rank-major population): which trees are kept (read by the next
    generation), how many rows every rank contributes, and where each kept tree lands in the gathered table.
    -> (per_rank bool[world][n_local], cap, elite_rows int32, parent_rows int32 — the lists as TABLE rows).
    ``cap`` — the row count every rank pads its block to — is, when not given, the largest number of kept trees on any rank:
    the step's one host sync (one integer).  A caller that passes a bound (``min(n_local, len(elites) + len(parents))`` never
    overflows) avoids the sync and gathers more rows."""
    dev = parents.device
    kept = torch.zeros(pop, dtype=torch.bool, device=dev)
    kept[parents.long()] = True
    if elites.numel():
        kept[elites.long()] = True
    per_rank = kept.view(world, pop // world)
    if cap is None:
        cap = int(per_rank.sum(1).max())
    # a rank sends its kept trees in ascending local index (kept_rows): tree g of rank r is row r * cap + (kept trees of r below g)
    below = torch.cumsum(per_rank.to(torch.int64), dim=1) - 1
    table_row = (below + torch.arange(world, device=dev)[:, None] * cap).view(-1)
    return per_rank, cap, table_row[elites.long()].to(torch.int32).contiguous(), table_row[parents.long()].to(torch.int32).contiguous()


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


def random_words(seed: int, generation: int, rows: int, lo: int, hi: int, device, first_row: int = 0) -> torch.Tensor:
    """int32[rows][hi - lo]: word k (first_row <= k < first_row + rows) of offspring i in [lo, hi), uniform in [0, 2^31 - 1) like
    torch.randint(0, 2^31 - 1)"""
    i = torch.arange(lo, hi, dtype=torch.int64, device=device)[None, :]
    k = torch.arange(first_row, first_row + rows, dtype=torch.int64, device=device)[:, None]
    base = _mix64(torch.tensor([seed * 1000003 + generation], dtype=torch.int64, device=device))
    x = _mix64(base + (k << 40) + i)
    return (((x >> 33) & 0x7FFFFFFF) % (2**31 - 1)).to(torch.int32)


class _Population:
    """what a selection operator may read of the (sharded) population: its size"""

    def __init__(self, pop_size: int):
        self.pop_size = pop_size

    def __len__(self):
        return self.pop_size

    def __getattr__(self, name):
        raise AttributeError(f"a selection operator of a sharded run sees the gathered fitness vector and pop_size only, not "
                             f"Forest.{name}: the trees live on other ranks")


class ShardedGeneticProgramming:
    """GeneticProgramming over a population sharded across the ranks of ``group``.

    ``local_forest`` is this rank's block of the population (equal sizes on all ranks).  ``step`` takes the fitness of the
    LOCAL trees and returns the local block of the next generation.

    ``selection``: any ``BaseSelection`` (genetic_programming.py:110: ``elite_indices, next_indices = self.selection(forest,
    fitness)``).  Every rank holds the identical gathered fitness vector and runs the operator on it under a generator seeded
    with (seed, generation), so all ranks — and every world size — get the same two index lists; the
    survivor list may repeat trees (TournamentSelection, selection/tournament.py:59-133) and is used exactly as
    crossover/default.py:37-58 uses it: parents are drawn uniformly from the LIST.  ``DefaultSelection`` takes the
    one-launch radix select instead of a sort (same sets).

    ``exchange``: "rows" — all-gather the fitness (4 B per tree), select, all-gather only the rows the lists name;
    "packed" — ONE all-gather of {fitness | value | type | size} per tree (4 + 8 L bytes, SURVEY.md §8e's exact-semantics
    variant, the north star's "single all-gather"), then select on the gathered table.
    ``cap`` (rows mode): "exact" — every rank pads its block to the largest kept count of any rank (one host sync per step);
    "bound" — to ``min(n_local, n_elite + n_surv)`` (``max`` for DefaultSelection, whose lists nest): no host sync, more rows.
    ``last_exchange`` records what the step sent."""

    def __init__(self, local_forest: Forest, mutation_rate: float, mutation_descriptor: GenerateDescriptor,
                 selection: Optional[BaseSelection] = None, seed: int = 0, group=None, exchange: str = "rows", cap: str = "exact"):
        if selection is None:
            selection = DefaultSelection(survival_rate=0.3, elite_rate=0.01)
        if not isinstance(selection, BaseSelection):
            raise TypeError(f"selection must be a BaseSelection, got {type(selection).__name__}: the sharded step has no "
                            "deterministic way to run it on every rank")
        assert exchange in ("rows", "packed"), f"exchange should be 'rows' or 'packed', but got {exchange}"
        assert cap in ("exact", "bound"), f"cap should be 'exact' or 'bound', but got {cap}"
        self.forest = local_forest
        self.mutation_rate = mutation_rate
        self.descriptor = mutation_descriptor
        self.selection = selection
        self.exchange_mode = exchange
        self.cap_mode = cap
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
        self.last_exchange = {}

    # -- selection, identical on every rank --------------------------------------------------------
    def select(self, fit_all: torch.Tensor):
        """-> (elites int32[n_elite], parents int32[n_surv]): global tree indices; parents may repeat"""
        pop = fit_all.shape[0]
        if type(self.selection) is DefaultSelection:
            n_elite, n_surv = self.selection.counts(pop)
            assert n_surv >= 1, "the selection leaves no parent (crossover/default.py:40 draws from an empty range)"
            return default_lists(fit_all, n_elite, n_surv)
        dev = fit_all.device
        counter_based = getattr(self.selection, "counter_based", None)
        if counter_based is not None and os.environ.get("EVOGP_NATIVE_TOURNAMENT", "1") != "0":
            # operators that can draw from the counter-based words (TournamentSelection with its default arguments): one launch on a
            # GPU, the same numbers from torch ops elsewhere, no generator state
            lists = counter_based(fit_all, self.seed, self.generation)
            if lists is not None:
                return lists
        seed = int(_mix64(torch.tensor([self.seed * 1000003 + self.generation], dtype=torch.int64))[0]) & 0x7FFFFFFFFFFF
        with torch.random.fork_rng(devices=[dev] if dev.type == "cuda" else [], enabled=True):
            # seed exactly the generators that were forked: the CPU's and this device's (torch.manual_seed would also reseed the
            # other devices of a process that holds several, and those are not restored)
            torch.default_generator.manual_seed(seed)
            if dev.type == "cuda":
                torch.cuda.manual_seed(seed) if dev.index is None else torch.cuda.default_generators[dev.index].manual_seed(seed)
            try:
                elites, parents = self.selection(_Population(pop), fit_all)
            except AttributeError as e:
                raise TypeError(f"{type(self.selection).__name__} cannot run in a sharded step: {e}") from None
        assert parents.numel() >= 1, "the selection leaves no parent (crossover/default.py:40 draws from an empty range)"
        return elites.to(torch.int32).contiguous(), parents.to(torch.int32).contiguous()

    def _cap_bound(self, n_elite: int, n_surv: int) -> int:
        both = max(n_elite, n_surv) if type(self.selection) is DefaultSelection else n_elite + n_surv
        return max(1, min(self.n_local, both))

    # -- the exchange step -------------------------------------------------------------------------
    def exchange(self, local_fitness: torch.Tensor):
        """-> (table Forest, elite_rows int32[n_elite], parent_rows int32[n_surv] — rows of the table, global pop)"""
        f = self.forest
        fit = local_fitness.to(torch.float32).contiguous()
        L = f.max_tree_len
        if self.world == 1:
            elites, parents = self.select(fit)
            self.last_exchange = dict(mode="none", collectives=0, bytes_sent=0)
            return f, elites, parents, self.pop_size
        if self.exchange_mode == "packed":
            # ONE collective: {fitness | value | type | size} of every local tree
            send = torch.cat([fit.view(-1, 1).view(torch.uint8), _pack(f)], dim=1).contiguous()
            table = torch.empty((self.pop_size, send.shape[1]), dtype=torch.uint8, device=send.device)
            dist.all_gather_into_tensor(table, send, group=self.group)
            fit_all = table[:, :4].contiguous().view(torch.float32).view(-1)
            elites, parents = self.select(fit_all)
            self.last_exchange = dict(mode="packed", collectives=1, bytes_sent=send.numel(), rows_sent=self.n_local)
            return _unpack(table[:, 4:], L, f.input_len, f.output_len), elites, parents, self.pop_size
        fit_all = torch.empty(self.pop_size, dtype=torch.float32, device=fit.device)
        dist.all_gather_into_tensor(fit_all, fit, group=self.group)
        elites, parents = self.select(fit_all)
        cap = None if self.cap_mode == "exact" else self._cap_bound(elites.numel(), parents.numel())
        per_rank, cap, elite_rows, parent_rows = plan_exchange(elites, parents, self.pop_size, self.world, cap)
        send = _pack(f, kept_rows(per_rank[self.rank], cap))
        table = torch.empty((self.world * cap, send.shape[1]), dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(table, send, group=self.group)
        self.last_exchange = dict(mode="rows", cap=self.cap_mode, collectives=2, bytes_sent=fit.numel() * 4 + send.numel(), rows_sent=cap)
        return _unpack(table, L, f.input_len, f.output_len), elite_rows, parent_rows, self.pop_size

    def step(self, local_fitness: torch.Tensor) -> Forest:
        assert local_fitness.shape == (self.n_local,)
        table, elite_rows, parent_rows, pop = self.exchange(local_fitness)
        lo, hi = self.rank * self.n_local, (self.rank + 1) * self.n_local
        if local_fitness.is_cuda and self.descriptor.max_tree_len == table.max_tree_len \
                and os.environ.get("EVOGP_NATIVE_STEP", "1") != "0":
            self.forest = self.slice_native(table, elite_rows, parent_rows, pop, lo, hi)
        else:
            self.forest = self.slice_torch(table, elite_rows, parent_rows, pop, lo, hi)
        self.generation += 1
        return self.forest

    def next_slice_native(self, full: Forest, fitness: torch.Tensor, lo: int, hi: int) -> Forest:
        """rows [lo, hi) of the next generation from the WHOLE population and its fitness (tests, single-table use)"""
        elites, parents = self.select(fitness.to(torch.float32))
        return self.slice_native(full, elites, parents, full.pop_size, lo, hi)

    def next_slice_torch(self, full: Forest, fitness: torch.Tensor, lo: int, hi: int) -> Forest:
        elites, parents = self.select(fitness.to(torch.float32))
        return self.slice_torch(full, elites, parents, full.pop_size, lo, hi)

    def slice_native(self, table: Forest, elite_rows: torch.Tensor, parent_rows: torch.Tensor, pop: int, lo: int, hi: int) -> Forest:
        """Rows [lo, hi) of the next generation with the fused breeding pass (csrc/breed.hip).  ``table`` holds the trees
        that can be parents or elites, ``elite_rows`` / ``parent_rows`` name them (table rows; parents may repeat), ``pop``
        is the size of the WHOLE population.  Every rank computes the SAME six random words per offspring and the same
        generation keys (counter-based), generates donors only for its own mutating offspring (tree index = global
        offspring index) and builds only its own rows.  The union over the ranks is the single-device result."""
        full = table
        dev = parent_rows.device
        L = full.max_tree_len
        n_elite = elite_rows.numel()
        n_new = pop - n_elite
        below = int(min(max(self.mutation_rate, 0.0), 1.0) * (2**31 - 1))
        d = self.descriptor
        rows = hi - lo
        o_lo, o_hi = max(lo, n_elite) - n_elite, max(hi, n_elite) - n_elite   # my offspring indices
        # no array of random words: both kernels compute word k of offspring i = hash(seed, generation, k, i) themselves (the
        # numbers evogp_hip_random_words / random_words give), and the generation keys are words (7, 0 / 1): every rank builds
        # its rows from the same words whatever the world size, with two launches (donors, breeding pass)
        if o_hi > o_lo:
            donors = torch.ops.evogp_hip.tree_generate_masked_hashed(
                o_hi - o_lo, L, d.input_len, d.output_len, d.const_samples.shape[0], d.out_prob, d.const_prob,
                d.depth2leaf_probs, d.roulette_funcs, d.const_samples, o_lo, self.seed, self.generation, below)
            # donors cover my offspring rows only; the breeding pass skips the elite rows at the head of the range
        else:  # a slice of elites only: donors are never read
            donors = (torch.empty((rows, L), dtype=torch.float32, device=dev), torch.empty((rows, L), dtype=torch.int16, device=dev),
                      torch.empty((rows, L), dtype=torch.int16, device=dev))
        value, ntype, size = full._tensors()
        nv, nt, ns = torch.ops.evogp_hip.breed_rows_hashed(pop, L, value, ntype, size, elite_rows, parent_rows, self.seed, self.generation,
                                                                  below, *donors, lo, rows)
        return Forest(full.input_len, full.output_len, nv, nt, ns, func_mask=Forest.join_masks(full.func_mask, d.func_mask))

    def slice_torch(self, table: Forest, elite_rows: torch.Tensor, parent_rows: torch.Tensor, pop: int, lo: int, hi: int) -> Forest:
        """The same slice composed from the reference's operators (any device): genetic_programming.py:110-122."""
        full = table
        dev = parent_rows.device
        n_elite = elite_rows.numel()
        elite_idx, surv_idx = elite_rows, parent_rows             # identical on every rank
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
