"""TEST-ONLY: CPU implementations of torch.ops.evogp_cuda.* / evogp_hip.* backed by the CPU oracle,
so that host-side logic (Forest API, genetic operators, population sharding over gloo) can be
exercised in the GPU-less container.  The product registers no CPU implementation; importing this
module is what a test does when it needs the ops on CPU tensors."""
import numpy as np
import torch

import evogp_amd  # noqa: F401  (defines the schemas)
from oracle.pyoracle import Oracle

_O = Oracle("port")
_done = False


def _np(t):
    return t.detach().cpu().numpy()


def _t3(a):
    return tuple(torch.from_numpy(x) for x in a)


def register():
    global _done
    if _done:
        return
    _done = True

    def gen(pop, L, var_len, out_len, n_const, out_prob, const_prob, keys, d2l, rou, cs, offset=0):
        k = _np(keys.to(torch.int64)).astype(np.uint32)
        return _t3(_O.generate(pop, L, var_len, out_len, out_prob, const_prob, k, _np(d2l), _np(rou), _np(cs), offset))

    torch.library.impl("evogp_cuda::tree_generate", "CPU")(gen)
    torch.library.impl("evogp_hip::tree_generate_offset", "CPU")(gen)
    torch.library.impl("evogp_cuda::tree_mutate", "CPU")(
        lambda pop, L, v, t, s, idx, nv, nt, ns: _t3(_O.mutate(_np(v), _np(t), _np(s), _np(idx), _np(nv), _np(nt), _np(ns))))
    torch.library.impl("evogp_cuda::tree_crossover", "CPU")(
        lambda po, pn, L, v, t, s, li, ri, ln, rn: _t3(_O.crossover(_np(v), _np(t), _np(s), _np(li), _np(ri), _np(ln), _np(rn))))
    torch.library.impl("evogp_cuda::tree_evaluate", "CPU")(
        lambda pop, L, vl, ol, v, t, s, x: torch.from_numpy(_O.evaluate(_np(v), _np(t), _np(s), _np(x), ol)))
    torch.library.impl("evogp_cuda::tree_SR_fitness", "CPU")(
        lambda pop, D, L, vl, ol, mse, v, t, s, X, y, kt: torch.from_numpy(_O.sr_fitness(_np(v), _np(t), _np(s), _np(X), _np(y), mse, 1)))
    torch.library.impl("evogp_hip::tree_batch_evaluate", "CPU")(
        lambda pop, D, L, vl, ol, v, t, s, X: torch.from_numpy(_O.batch_evaluate(_np(v), _np(t), _np(s), _np(X), ol)))
