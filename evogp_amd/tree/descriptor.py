"""GenerateDescriptor — turns a user-level tree-shape configuration into the three small tensors
``tree_generate`` consumes.  API and numerics follow src/evogp/tree/descriptor.py:8-188:

* ``depth2leaf_probs`` f32[10]  = [layer_leaf_prob]*(max_layer_cnt-1) + [1.0]*rest          (:33-38)
* ``roulette_funcs``   f32[29]  = cumsum of the normalised function weights                 (:106-111)
  (plus the per-arity roulettes used by point mutations, :113-139)
* ``const_samples``    f32[n]   = explicit list/tensor or ``sample_cnt`` uniform draws in ``const_range``
"""
from __future__ import annotations

import warnings
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from .utils import FUNCS_NAMES, MAX_FULL_DEPTH, MAX_STACK, Func, check_tensor, default_device, dict2prob, func_arity


def _need(ok: bool, message: str) -> None:
    """argument errors are AssertionErrors, as in the reference (its constructor is a chain of asserts)"""
    if not ok:
        raise AssertionError(message)


def check_tree_length(max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob) -> Tensor:
    """``depth2leaf_probs`` for trees of at most ``max_layer_cnt`` layers — after checking that the LARGEST such tree (every
    node a function of the widest arity in use) fits a row of ``max_tree_len`` nodes (descriptor.py:8-39)."""
    unknown = [name for name in using_funcs if name not in FUNCS_NAMES]
    _need(not unknown, f"function name(s) {unknown} not among {FUNCS_NAMES}")
    widest = max(func_arity(FUNCS_NAMES.index(name)) for name in using_funcs)
    # 1 + a + a^2 + ... + a^(layers - 1) nodes
    largest = max_layer_cnt if widest <= 1 else (widest**max_layer_cnt - 1) // (widest - 1)
    _need(max_tree_len >= largest,
          f"a full tree of {max_layer_cnt} layers of arity-{widest} functions has {largest} nodes: max_tree_len={max_tree_len} cannot hold it")
    function_layers = max_layer_cnt - 1
    return torch.tensor([layer_leaf_prob] * function_layers + [1.0] * (MAX_FULL_DEPTH - function_layers), device=default_device())


class GenerateDescriptor:
    def __init__(
        self,
        max_tree_len: int,
        input_len: int,
        output_len: int,
        const_prob: float = 0.5,
        out_prob: float = 0.5,
        depth2leaf_probs: Optional[Tensor] = None,
        roulette_funcs: Optional[Tensor] = None,
        const_samples: Optional[Union[list, Tensor]] = None,
        using_funcs: Optional[Union[dict, list]] = None,
        max_layer_cnt: Optional[int] = None,
        layer_leaf_prob: Optional[float] = 0.2,
        const_range: Optional[Tuple[float, float]] = None,
        sample_cnt: Optional[int] = None,
    ):
        self._ctor_kwargs = {k: v for k, v in locals().items() if k not in ("self", "__class__")}

        _need(max_tree_len <= MAX_STACK, f"rows of {max_tree_len} nodes exceed the operand-stack bound of the kernels ({MAX_STACK})")
        for name, n in (("input_len", input_len), ("output_len", output_len)):
            _need(isinstance(n, int) and n > 0, f"{name} must be a positive int, got {n!r}")
        for name, q in (("const_prob", const_prob), ("out_prob", out_prob)):
            _need(0.0 <= q <= 1.0, f"{name} is a probability, got {q}")
        if output_len > 1 and out_prob == 0.0:
            warnings.warn(f"{output_len} outputs but out_prob = 0: no generated node will ever write one")

        self.max_tree_len, self.input_len, self.output_len = max_tree_len, input_len, output_len
        self.const_prob, self.out_prob = const_prob, out_prob
        self.depth2leaf_probs = self._leaf_probabilities(depth2leaf_probs, max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob)
        self.roulette_funcs, self.roulette_ufuncs, self.roulette_bfuncs, self.roulette_tfuncs = self._roulettes(roulette_funcs, using_funcs)
        self.const_samples = self._constants(const_samples, const_range, sample_cnt)
        self.func_mask = self._function_mask(roulette_funcs, using_funcs)

    @staticmethod
    def _function_mask(roulette_funcs, using_funcs) -> int:
        """bit f = function id f can be generated (its weight is positive); 0 = unknown.  Forests remember the masks of the
        descriptors their trees came from, and tree_SR_fitness skips launches such a forest cannot need (include/evogp_hip.h
        evogp_hip_sr_fitness_hinted).  A descriptor given a ready-made roulette tensor says "unknown" (reading it back would
        synchronise with the device)."""
        if roulette_funcs is not None or not isinstance(using_funcs, (dict, list)):
            return 0
        weights = dict.fromkeys(using_funcs, 1.0) if isinstance(using_funcs, list) else using_funcs
        mask = 0
        for name, w in weights.items():
            if w > 0 and name in FUNCS_NAMES:
                mask |= 1 << FUNCS_NAMES.index(name)
        return mask

    @staticmethod
    def _as_f32(t, what, shape=None) -> Tensor:
        t = check_tensor(t).to(torch.float32).contiguous()
        if shape is not None:
            _need(tuple(t.shape) == shape, f"{what} must have shape {shape}, got {tuple(t.shape)}")
        return t

    def _leaf_probabilities(self, given, max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob) -> Tensor:
        """f32[10]: probability that a node generated at depth d is a leaf (descriptor.py:33-38, 93-100)"""
        if given is None:
            _need(max_layer_cnt is not None and layer_leaf_prob is not None,
                  "without depth2leaf_probs, max_layer_cnt and layer_leaf_prob define the tree shape: both are required")
            given = check_tree_length(max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob)
        return self._as_f32(given, "depth2leaf_probs", (MAX_FULL_DEPTH,))

    def _roulettes(self, given, using_funcs):
        """f32[29] cumulative function weights (descriptor.py:106-111) and, when built from ``using_funcs``, one cumulative
        table per arity class for the point mutations (:113-139: the class's weights in place, zeros elsewhere)"""
        per_arity = (None, None, None)
        if given is None:
            _need(isinstance(using_funcs, (dict, list)), "without roulette_funcs, using_funcs (a list of names or a name -> weight dict) is required")
            weights = dict.fromkeys(using_funcs, 1.0) if isinstance(using_funcs, list) else using_funcs
            prob = dict2prob(weights)
            dev = default_device()
            given = torch.cumsum(prob, dim=0, dtype=torch.float32).to(dev)

            def of_class(lo, hi):
                only = torch.zeros_like(prob)
                only[lo:hi] = prob[lo:hi]
                return torch.cumsum(only, dim=0, dtype=torch.float32).to(dev)

            per_arity = (of_class(Func.UF_START, Func.END), of_class(Func.BF_START, Func.UF_START), of_class(Func.TF_START, Func.BF_START))
        return (self._as_f32(given, "roulette_funcs", (Func.END,)),) + per_arity

    def _constants(self, given, const_range, sample_cnt) -> Tensor:
        """f32[n]: the pool constants are drawn from — given, or ``sample_cnt`` uniform draws in ``const_range``"""
        if given is None:
            _need(const_range is not None and sample_cnt is not None, "without const_samples, const_range and sample_cnt are required")
            lo, hi = const_range
            given = torch.rand(sample_cnt, device=default_device()) * (hi - lo) + lo
        elif isinstance(given, list):
            given = torch.tensor(given, dtype=torch.float32, device=default_device())
        given = self._as_f32(given, "const_samples")
        _need(given.dim() == 1, f"const_samples is a flat pool of values, got {given.dim()} dimensions")
        return given

    def update(self, **kwargs) -> "GenerateDescriptor":
        """A new descriptor built from the stored constructor arguments overridden by ``kwargs``
        (descriptor.py:186-188; random ``const_samples`` are re-drawn when ``const_range`` is used)."""
        merged = dict(self._ctor_kwargs)
        merged.update(kwargs)
        return self.__class__(**merged)

    def __str__(self):
        return (
            f"max_tree_len: {self.max_tree_len}\ninput_len: {self.input_len}\noutput_len: {self.output_len}\n"
            f"const_prob: {self.const_prob}\nout_prob: {self.out_prob}\ndepth2leaf_probs: {self.depth2leaf_probs}\n"
            f"roulette_funcs: {self.roulette_funcs}\nconst_samples: {self.const_samples}\n"
        )
