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


# ---- selectors and the non-default selections ---------------------------------------------------------------------
# Reference: selection/selection_utils.py:6-130, rank.py, roulette.py, tournament.py:59-133, truncation.py.  All of them
# are index programs over the fitness vector; they are written here for the documented behaviour (the reference's
# Rank/Truncation code indexes its probability vectors with tree indices where ranks are meant) and run without host
# syncs.  ``fitness`` is "higher is better"; -inf / NaN entries are never preferred.

def _clean(fitness: torch.Tensor) -> torch.Tensor:
    return torch.nan_to_num(fitness.to(torch.float32), nan=float("-inf"))


class BaseSelector:
    """fitness, n -> int32[n] indices of chosen individuals (with replacement)"""

    def __call__(self, fitness: torch.Tensor, choosed_num: int) -> torch.Tensor:
        raise NotImplementedError


class RankSelector(BaseSelector):
    """Linear ranking: P(rank r) = (1/n) (1 + sp (1 - 2 r / (n - 1))), r = 0 for the best."""

    def __init__(self, selection_pressure: float = 0.5):
        assert 0 <= selection_pressure <= 1, "selection_pressure should be in [0, 1]"
        self.sp = selection_pressure

    def __call__(self, fitness, choosed_num):
        n = fitness.shape[0]
        order = torch.sort(_clean(fitness), descending=True, stable=True).indices
        r = torch.arange(n, dtype=torch.float32, device=fitness.device)
        prob = (1.0 / n) * (1.0 + self.sp * (1.0 - 2.0 * r / max(n - 1, 1)))
        return order[torch.multinomial(prob, choosed_num, replacement=True)].to(torch.int32)


class RouletteSelector(BaseSelector):
    """P(i) proportional to fitness_i (negative / non-finite fitness counts as 0; all-zero falls back to uniform)."""

    def __call__(self, fitness, choosed_num):
        w = _clean(fitness).clamp(min=0.0)
        w = torch.where(torch.isfinite(w), w, torch.zeros_like(w))
        w = w + (w.sum() <= 0).to(w.dtype)  # uniform when nothing is positive
        return torch.multinomial(w, choosed_num, replacement=True).to(torch.int32)


class TruncationSelector(BaseSelector):
    """Uniform among the best ``survivor_rate`` fraction."""

    def __init__(self, survivor_rate: float = 0.5):
        assert 0 <= survivor_rate <= 1, "survivor_rate should be in [0, 1]"
        self.survivor_rate = survivor_rate

    def __call__(self, fitness, choosed_num):
        n = fitness.shape[0]
        top = max(1, int(n * self.survivor_rate))
        order = torch.sort(_clean(fitness), descending=True, stable=True).indices
        return order[torch.randint(0, top, (choosed_num,), device=fitness.device)].to(torch.int32)


class TournamentSelector(BaseSelector):
    """``choosed_num`` tournaments of ``tournament_size`` contenders (selection/tournament.py:59-133): the k-th best contender
    wins with probability p (1 - p)^k, ranks past the tournament fall back to the best (:98-104).  The contenders come in
    PASSES of ``n // tournament_size`` tournaments, as in the reference (:117-121: one ``torch.multinomial`` row of
    ``n_tournament * t_size`` uniform draws per pass): with ``replace=False`` nobody enters twice in a pass.

    Split into ``draw`` (the random numbers) and ``apply`` (deterministic), like the mutation operators:
    tests/golden/make_tournament_golden.py records the reference's own draws, ``apply`` must return its survivors."""

    def __init__(self, tournament_size: int, best_probability: float = 1, replace: bool = True):
        assert tournament_size >= 1
        self.t_size = tournament_size
        self.best_p = best_probability
        self.replace = replace

    def passes(self, n: int, count: int):
        """-> (tournaments per pass, number of passes) -- tournament.py:117-119"""
        per_pass = max(n // self.t_size, 1)
        return per_pass, (count - 1) // per_pass + 1

    def contenders(self, n: int, count: int, device) -> torch.Tensor:
        """int64[count][t_size]: the tournaments' members"""
        t = self.t_size
        per_pass, passes = self.passes(n, count)
        if self.replace:
            return torch.randint(0, n, (passes * per_pass, t), device=device)[:count]
        perm = torch.rand((passes, n), device=device).argsort(dim=1)[:, : per_pass * t]
        if perm.shape[1] < per_pass * t:  # population smaller than one tournament
            perm = perm.repeat(1, (per_pass * t) // perm.shape[1] + 1)[:, : per_pass * t]
        return perm.reshape(-1, t)[:count]

    def draw(self, n: int, count: int, device):
        """-> (contenders int64[count][t], u float32[count] or None when the best always wins)"""
        c = self.contenders(n, count, device)
        u = None if (self.best_p >= 1 and self.t_size > 1000) else torch.rand(count, device=device)
        return c, u

    def apply(self, fitness: torch.Tensor, contenders: torch.Tensor, u: Optional[torch.Tensor]) -> torch.Tensor:
        f = _clean(fitness)
        c = contenders.to(torch.int64)
        cf = f[c]
        if u is None or self.best_p >= 1:
            # p = 1: log(u) / log(0) = -0 -> rank 0, the best (tournament.py:98-104); :123-124 takes the arg-max directly
            # for tournaments above 1000 contenders
            pick = cf.argmax(dim=1, keepdim=True)
        else:
            rank = cf.argsort(dim=1, descending=True, stable=True)
            one_minus_p = 1 - torch.tensor(self.best_p, dtype=torch.float32, device=f.device)   # in float32, as :100-101
            nth = (torch.log(u.to(torch.float32)) / torch.log(one_minus_p)).to(torch.int64)
            nth = torch.where((nth >= self.t_size) | (nth < 0), torch.zeros_like(nth), nth)
            pick = rank.gather(1, nth[:, None])
        return c.gather(1, pick).squeeze(1).to(torch.int32)

    def __call__(self, fitness, choosed_num):
        c, u = self.draw(fitness.shape[0], choosed_num, fitness.device)
        return self.apply(fitness, c, u)


class _SelectorSelection(BaseSelection):
    """survivors = selector(fitness, survivor count), elites = the best elite count (rank.py, roulette.py, ...)"""

    def __init__(self, selector: BaseSelector, survivor_rate: float = 0.5, elite_rate: float = 0,
                 survivor_cnt: Optional[int] = None, elite_cnt: Optional[int] = None):
        assert 0 <= survivor_rate <= 1, "survivor_rate should be in [0, 1]"
        assert 0 <= elite_rate <= 1, "elite_rate should be in [0, 1]"
        self.selector = selector
        self.survivor_rate = survivor_rate
        self.survivor_cnt = survivor_cnt
        self.elite_rate = elite_rate
        self.elite_cnt = elite_cnt

    def counts(self, pop_size: int):
        n_surv = self.survivor_cnt if self.survivor_cnt is not None else int(pop_size * self.survivor_rate)
        n_elite = self.elite_cnt if self.elite_cnt is not None else int(pop_size * self.elite_rate)
        return n_elite, n_surv

    def __call__(self, forest: Forest, fitness: torch.Tensor):
        n_elite, n_surv = self.counts(forest.pop_size)
        survivors = self.selector(fitness, n_surv)
        if n_elite > 0:
            elites = torch.topk(_clean(fitness), n_elite, sorted=True).indices.to(torch.int32)
        else:
            elites = torch.empty(0, dtype=torch.int32, device=fitness.device)
        return elites, survivors


class RankSelection(_SelectorSelection):
    def __init__(self, selection_pressure: float = 0.5, survivor_rate: float = 0.5, elite_rate: float = 0,
                 survivor_cnt: Optional[int] = None, elite_cnt: Optional[int] = None):
        super().__init__(RankSelector(selection_pressure), survivor_rate, elite_rate, survivor_cnt, elite_cnt)


class RouletteSelection(_SelectorSelection):
    def __init__(self, survivor_rate: float = 0.5, elite_rate: float = 0, survivor_cnt: Optional[int] = None,
                 elite_cnt: Optional[int] = None):
        super().__init__(RouletteSelector(), survivor_rate, elite_rate, survivor_cnt, elite_cnt)


class TruncationSelection(_SelectorSelection):
    def __init__(self, survivor_rate: float = 0.5, elite_rate: float = 0, survivor_cnt: Optional[int] = None,
                 elite_cnt: Optional[int] = None):
        super().__init__(TruncationSelector(survivor_rate), survivor_rate, elite_rate, survivor_cnt, elite_cnt)


class TournamentSelection(_SelectorSelection):
    def __init__(self, tournament_size: int, best_probability: float = 1, replace: bool = True,
                 survivor_rate: float = 0.5, elite_rate: float = 0, survivor_cnt: Optional[int] = None,
                 elite_cnt: Optional[int] = None):
        super().__init__(TournamentSelector(tournament_size, best_probability, replace), survivor_rate, elite_rate,
                         survivor_cnt, elite_cnt)

    def counter_based(self, fitness: torch.Tensor, seed: int, generation: int):
        """The operator with its contenders taken from the counter-based words of evogp_amd/parallel.py (contender k of tournament
        i = word(seed, generation, 16 + k, i) % n) instead of torch's generator: every rank of a sharded run names the same
        contenders without sharing generator state, and on a GPU the whole selection is two launches (csrc/select.hip
        tournament_kernel + the radix select for the elites).  Same distribution as ``__call__`` (tournament.py:59-133).  Only for
        the reference's default arguments — contenders with replacement, the best one wins; None otherwise (the caller then runs
        ``__call__`` under a seeded generator)."""
        sel = self.selector
        if not sel.replace or sel.best_p < 1:
            return None
        from ..parallel import _select_key, default_lists, random_words

        n = fitness.shape[0]
        n_elite, n_surv = self.counts(n)
        if n_surv < 1:
            return None
        fit = fitness.to(torch.float32).contiguous()
        if fit.is_cuda:
            parents = torch.ops.evogp_hip.tournament_select(fit, n_surv, sel.t_size, seed, generation)
        else:
            c = random_words(seed, generation, sel.t_size, 0, n_surv, fit.device, first_row=16).to(torch.int64) % n     # (t, n_surv)
            best = _select_key(fit)[c].argmax(dim=0, keepdim=True)                                                        # first of equal keys
            parents = c.gather(0, best).squeeze(0).to(torch.int32)
        elites = default_lists(fit, n_elite, max(n_elite, 1))[0] if n_elite > 0 else torch.empty(0, dtype=torch.int32, device=fit.device)
        return elites.contiguous(), parents.contiguous()
