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


def check_tree_length(max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob) -> Tensor:
    """Assert that a full tree of ``max_layer_cnt`` layers with the largest arity in use fits in
    ``max_tree_len`` nodes, and build ``depth2leaf_probs`` (descriptor.py:8-39)."""
    for name in using_funcs:
        assert name in FUNCS_NAMES, f"Unknown function name: {name}, total functions are {FUNCS_NAMES}"
    max_arity = max(func_arity(FUNCS_NAMES.index(name)) for name in using_funcs)
    if max_arity > 1:
        full_len = int((max_arity**max_layer_cnt - 1) / (max_arity - 1))
    else:
        full_len = max_layer_cnt
    assert max_tree_len >= full_len, (
        f"max_tree_len={max_tree_len} is too small\n"
        f"max_tree_len should >={full_len}\n"
        f"as the max arity of funcs is {max_arity} and the max layer is {max_layer_cnt}."
    )
    inner = max_layer_cnt - 1
    return torch.tensor([layer_leaf_prob] * inner + [1.0] * (MAX_FULL_DEPTH - inner), device=default_device())


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

        assert max_tree_len <= MAX_STACK, f"max_tree_len={max_tree_len} is too large, MAX_STACK={MAX_STACK}"
        assert isinstance(input_len, int) and input_len > 0, "input_len should be a positive integer"
        assert isinstance(output_len, int) and output_len > 0, "output_len should be a positive integer"
        assert 0.0 <= const_prob <= 1.0, "const_prob should be in [0.0, 1.0]"
        assert 0.0 <= out_prob <= 1.0, "out_prob should be in [0.0, 1.0]"
        if output_len > 1 and out_prob == 0.0:
            warnings.warn(f"output_len={output_len} > 1, but out_prob={out_prob} is 0.0.")

        dev = default_device()
        if depth2leaf_probs is None:
            assert max_layer_cnt is not None, "max_layer_cnt should not be None when depth2leaf_probs is None"
            assert layer_leaf_prob is not None, "layer_leaf_prob should not be None when depth2leaf_probs is None"
            depth2leaf_probs = check_tree_length(max_tree_len, using_funcs, max_layer_cnt, layer_leaf_prob)

        roulette_ufuncs = roulette_bfuncs = roulette_tfuncs = None
        if roulette_funcs is None:
            assert using_funcs is not None, "func_prob should not be None when roulette_funcs is None"
            assert isinstance(using_funcs, (dict, list)), "func_prob should be a dictionary or a list"
            weights = {f: 1.0 for f in using_funcs} if isinstance(using_funcs, list) else using_funcs
            prob = dict2prob(weights)
            roulette_funcs = torch.cumsum(prob, dim=0, dtype=torch.float32).to(dev)

            def masked_roulette(lo, hi):
                part = torch.zeros_like(prob)
                part[lo:hi] = prob[lo:hi]
                return torch.cumsum(part, dim=0, dtype=torch.float32).to(dev)

            roulette_tfuncs = masked_roulette(Func.TF_START, Func.BF_START)
            roulette_bfuncs = masked_roulette(Func.BF_START, Func.UF_START)
            roulette_ufuncs = masked_roulette(Func.UF_START, Func.END)

        if const_samples is None:
            assert const_range is not None, "const_range should not be None when const_samples is None"
            assert sample_cnt is not None, "sample_cnt should not be None when const_samples is None"
            const_samples = torch.rand(sample_cnt, device=dev) * (const_range[1] - const_range[0]) + const_range[0]
        if isinstance(const_samples, list):
            const_samples = torch.tensor(const_samples, dtype=torch.float32, device=dev)

        depth2leaf_probs = check_tensor(depth2leaf_probs).to(torch.float32).contiguous()
        roulette_funcs = check_tensor(roulette_funcs).to(torch.float32).contiguous()
        const_samples = check_tensor(const_samples).to(torch.float32).contiguous()

        assert depth2leaf_probs.shape == (MAX_FULL_DEPTH,), (
            f"depth2leaf_probs shape should be ({MAX_FULL_DEPTH}), but got {depth2leaf_probs.shape}")
        assert roulette_funcs.shape == (Func.END,), (
            f"roulette_funcs shape should be ({Func.END}), but got {roulette_funcs.shape}")
        assert const_samples.dim() == 1, f"const_samples dim should be 1, but got {const_samples.dim()}"

        self.max_tree_len = max_tree_len
        self.input_len = input_len
        self.output_len = output_len
        self.const_prob = const_prob
        self.out_prob = out_prob
        self.depth2leaf_probs = depth2leaf_probs
        self.roulette_funcs = roulette_funcs
        self.roulette_ufuncs = roulette_ufuncs
        self.roulette_bfuncs = roulette_bfuncs
        self.roulette_tfuncs = roulette_tfuncs
        self.const_samples = const_samples

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
