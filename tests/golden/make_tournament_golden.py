#!/usr/bin/env python3
"""Golden vectors for TournamentSelection (BASELINE configs[2] names it): the REFERENCE's own operator
(/root/reference/src/evogp/algorithm/selection/tournament.py:59-133) executed in this container on the CPU with every
random number it draws recorded.

The reference hard-codes ``.cuda()`` and runs its per-tournament functions under ``torch.vmap(randomness="different")``;
here "cuda" means the CPU (as in make_mutation_golden.py) and ``torch.vmap`` is replaced by a plain loop over the batch —
one independent call per row, which is what randomness="different" means — so that ``torch.multinomial`` (the contenders
of one pass) and ``torch.rand`` (one number per tournament) can be logged in call order.  Each case stores the fitness
vector, the logged draws and the reference's (elite, survivor) indices in tests/golden/tournament_<case>.npz;
tests/test_host_logic.py feeds the draws to ``TournamentSelector.apply`` and compares.

    python tests/golden/make_tournament_golden.py          (needs /root/reference; never runs on the GPU box)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src"


def _is_cuda(d):
    return d is not None and str(d).startswith("cuda")


def _decuda(fn):
    def wrapped(*a, **k):
        if _is_cuda(k.get("device")):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for _name in ("arange", "rand", "randint", "zeros", "ones", "empty", "tensor", "full"):
    setattr(torch, _name, _decuda(getattr(torch, _name)))
torch.Tensor.cuda = lambda self, *a, **k: self


def _loop_vmap(fn=None, **kw):
    if fn is None:
        return lambda f: _loop_vmap(f, **kw)
    return lambda batch: torch.stack([fn(row) for row in batch])


torch.vmap = _loop_vmap

import cpu_ops  # noqa: E402  (the reference package imports its ops at import time)

cpu_ops.register()
for _m in [m for m in sys.modules if m == "evogp" or m.startswith("evogp.")]:
    del sys.modules[_m]
sys.modules["evogp.evogp_cuda"] = types.ModuleType("evogp.evogp_cuda")
sys.path.insert(0, REF)
import evogp  # noqa: E402  (the reference)

assert evogp.__file__.startswith(REF), evogp.__file__
from evogp.algorithm.selection.tournament import TournamentSelection  # noqa: E402

LOG = []
_multinomial0, _rand0 = torch.multinomial, torch.rand


def _multinomial(*a, **k):
    r = _multinomial0(*a, **k)
    LOG.append(("multinomial", r.clone()))
    return r


def _rand(*a, **k):
    r = _rand0(*a, **k)
    LOG.append(("rand", r.clone()))
    return r


torch.multinomial, torch.rand = _multinomial, _rand


class _Pop:  # the operator only reads pop_size
    def __init__(self, n):
        self.pop_size = n


CASES = {
    # name: (population, constructor kwargs, seed)
    "t3_p1": (300, dict(tournament_size=3, best_probability=1, replace=True, survivor_rate=0.5, elite_rate=0.02), 1),
    "t7_p08_replace": (500, dict(tournament_size=7, best_probability=0.8, replace=True, survivor_rate=0.6, elite_cnt=5), 2),
    "t4_p09_noreplace": (403, dict(tournament_size=4, best_probability=0.9, replace=False, survivor_rate=0.9, elite_rate=0.0), 3),
    "t5_p05_many_passes": (200, dict(tournament_size=5, best_probability=0.5, replace=False, survivor_cnt=190, elite_cnt=1), 4),
}


def main():
    for name, (n, kw, seed) in CASES.items():
        torch.manual_seed(seed)
        fitness = torch.randn(n, dtype=torch.float32)          # distinct values: no tie rule is exercised
        fitness[::17] = float("-inf")                           # what the pipeline makes of NaN (pipeline/standard.py:43)
        del LOG[:]
        elites, survivors = TournamentSelection(**kw)(_Pop(n), fitness)
        multi = [t for tag, t in LOG if tag == "multinomial"]
        rands = [t for tag, t in LOG if tag == "rand"]
        arrays = dict(fitness=fitness.numpy(), elites=elites.numpy().astype(np.int64), survivors=survivors.numpy().astype(np.int64),
                      contenders=torch.stack(multi).numpy().astype(np.int64), u=torch.cat(rands).numpy(),
                      meta=np.frombuffer(json.dumps(dict(case=name, pop=n, kwargs=kw, seed=seed)).encode(), dtype=np.uint8))
        np.savez_compressed(os.path.join(HERE, f"tournament_{name}.npz"), **arrays)
        print(f"{name}: {len(multi)} passes of {multi[0].numel()} contenders, {len(rands)} tournaments, {elites.numel()} elites, "
              f"{survivors.numel()} survivors ({len(set(survivors.tolist()))} distinct)")


if __name__ == "__main__":
    main()
