"""Forest — a population of expression trees as three dense row-major device tensors.

    batch_node_value   float32 (pop, max_tree_len)   variable index / constant / function id / OUT bits
    batch_node_type    int16   (pop, max_tree_len)   NType, bit 7 = OUT_NODE
    batch_subtree_size int16   (pop, max_tree_len)   subtree sizes, [:, 0] = live tree length

API surface follows src/evogp/tree/forest.py:11-499 (constructor, ``random_generate``,
``zero_generate``, ``forward``, ``batch_forward``, ``mutate``, ``crossover``, ``SR_fitness``,
indexing, concatenation, iteration, pickling).  Every heavy method is one call into
``torch.ops.evogp_cuda.*`` (evogp_amd/ops.py), i.e. one HIP kernel.  Differences from the
reference, all deliberate:

* ``batch_forward`` does NOT replicate the forest ``batch`` times (forest.py:151-161: three
  ``repeat_interleave`` copies + ``x.repeat``): it calls the non-replicating
  ``evogp_hip::tree_batch_evaluate`` and returns the same ``(pop, batch, out)`` tensor;
* ``random_generate`` accepts ``tree_index_offset`` / ``keys`` so a sharded population can be made
  bit-identical to the single-device one.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from torch import Tensor

from . import utils as _utils
from .descriptor import GenerateDescriptor
from .tree import Tree
from .utils import NType, check_tensor

_PREPARED_FORWARD = __import__("os").environ.get("EVOGP_PREPARED_FORWARD", "1") != "0"
_SR_MODES = {"hybrid parallel": 0, "data parallel": 1, "tree parallel": 2, "auto": 4}  # forest.py:340-347


class Forest:
    def __init__(self, input_len, output_len, batch_node_value: Tensor, batch_node_type: Tensor,
                 batch_subtree_size: Tensor, func_mask: int = 0):
        """``func_mask`` (no counterpart in the reference): bit f = function id f may occur in the trees, 0 = unknown.  Set by
        ``random_generate`` from the descriptor and handed on by the operators that combine forests; a forest built from raw
        tensors is "unknown".  ``SR_fitness`` passes it on as long as the tensors are untouched."""
        self.input_len = input_len
        self.output_len = output_len
        self.pop_size, self.max_tree_len = batch_node_value.shape
        shape = (self.pop_size, self.max_tree_len)
        assert batch_node_type.shape == shape, f"node_type shape should be {shape}, but got {batch_node_type.shape}"
        assert batch_subtree_size.shape == shape, (
            f"subtree_size shape should be {shape}, but got {batch_subtree_size.shape}")
        self.batch_node_value = batch_node_value
        self.batch_node_type = batch_node_type
        self.batch_subtree_size = batch_subtree_size
        self._func_mask = (int(func_mask), self._forest_key()) if func_mask else None

    @property
    def func_mask(self) -> int:
        """the functions that may occur in this forest (bit f = function id f), 0 when unknown or when a tensor was replaced or
        written in place since the mask was set"""
        m = getattr(self, "_func_mask", None)
        if m is None or m[1] != self._forest_key():
            return 0
        return m[0]

    @staticmethod
    def join_masks(*masks: int) -> int:
        """the mask of a forest whose trees come from forests / descriptors with these masks: unknown if any is"""
        out = 0
        for m in masks:
            if not m:
                return 0
            out |= m
        return out

    # ---- construction -------------------------------------------------------------------------
    @staticmethod
    def random_generate(pop_size: int, descriptor: GenerateDescriptor, keys: Optional[Tensor] = None,
                        tree_index_offset: int = 0) -> "Forest":
        assert isinstance(pop_size, int) and pop_size > 0, "pop_size should be a positive integer"
        if keys is None:
            # two 32-bit keys from torch's generator of the device (forest.py:51-58)
            try:
                keys = torch.randint(low=0, high=1000000, size=(2,), dtype=torch.uint32, device=_utils.default_device())
            except RuntimeError:  # back ends without a uint32 randint
                keys = torch.randint(0, 1000000, (2,), device=_utils.default_device()).to(torch.uint32)
        args = (pop_size, descriptor.max_tree_len, descriptor.input_len, descriptor.output_len,
                descriptor.const_samples.shape[0], descriptor.out_prob, descriptor.const_prob, keys,
                descriptor.depth2leaf_probs, descriptor.roulette_funcs, descriptor.const_samples)
        if tree_index_offset:
            value, ntype, size = torch.ops.evogp_hip.tree_generate_offset(*args, tree_index_offset)
        else:
            value, ntype, size = torch.ops.evogp_cuda.tree_generate(*args)
        return Forest(descriptor.input_len, descriptor.output_len, value, ntype, size, func_mask=descriptor.func_mask)

    @staticmethod
    def zero_generate(pop_size: int, max_tree_len: int, input_len: int, output_len: int) -> "Forest":
        """pop_size copies of the constant-0 tree (forest.py:86-110)."""
        dev = _utils.default_device()
        value = torch.zeros((pop_size, max_tree_len), dtype=torch.float32, device=dev)
        ntype = torch.zeros((pop_size, max_tree_len), dtype=torch.int16, device=dev)
        size = torch.zeros((pop_size, max_tree_len), dtype=torch.int16, device=dev)
        ntype[:, 0] = NType.CONST
        size[:, 0] = 1
        return Forest(input_len, output_len, value, ntype, size)

    def _tensors(self):
        return (self.batch_node_value.contiguous(), self.batch_node_type.contiguous(),
                self.batch_subtree_size.contiguous())

    # ---- evaluation ---------------------------------------------------------------------------
    def _forest_key(self):
        v, t, s = self.batch_node_value, self.batch_node_type, self.batch_subtree_size
        return (v.data_ptr(), t.data_ptr(), s.data_ptr(), v._version, t._version, s._version, self.pop_size, self.max_tree_len)

    def prepare_forward(self):
        """Decode a multi-output forest once for many ``forward`` calls (a rollout evaluates the same forest once per
        environment step, src/evogp/problem/brax_problem.py:54-93): the operation lists of csrc/evaluate_prepared.hip.  Returns
        the cached (workspace, with_fallback) or None when the forest is not eligible.  The cache is keyed on the tensors'
        identity and version counters, so an in-place edit of the forest invalidates it; ``forward`` calls this itself from
        its SECOND call on an unchanged forest on (one host sync per preparation: the count of trees left to the stack
        interpreter)."""
        v, t, s = self.batch_node_value, self.batch_node_type, self.batch_subtree_size
        if not (v.is_cuda and 2 <= self.output_len <= 32 and self.input_len <= 255 and v.is_contiguous() and t.is_contiguous() and s.is_contiguous()):
            return None
        key = self._forest_key()
        cached = getattr(self, "_prepared", None)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        ws, info = torch.ops.evogp_hip.tree_evaluate_prepare(self.pop_size, self.max_tree_len, self.input_len, self.output_len, v, t, s)
        with_fallback = bool(int(info[0]) != 0)
        self._prepared = (key, ws, with_fallback)
        return ws, with_fallback

    def forward(self, x: Tensor) -> Tensor:
        """One input row per tree: x (pop, input_len) -> (pop, output_len)."""
        x = check_tensor(x, self.batch_node_value.device)
        assert x.shape == (self.pop_size, self.input_len), (
            f"x shape should be ({self.pop_size}, {self.input_len}), but got {x.shape}")
        # a forest that is evaluated AGAIN unchanged is a policy population inside a rollout: from the second call on it runs
        # from its operation lists (multi-output forests only; EVOGP_PREPARED_FORWARD=0 switches this off).  Preparing costs a
        # host sync, so it never happens while the stream is being captured: RolloutProblem prepares before it captures.
        if self.output_len > 1 and x.is_cuda and _PREPARED_FORWARD:
            key = self._forest_key()
            prepared = None
            cached = getattr(self, "_prepared", None)
            if cached is not None and cached[0] == key:
                prepared = cached[1:]
            elif getattr(self, "_forward_seen", None) == key and not torch.cuda.is_current_stream_capturing():
                prepared = self.prepare_forward()
            self._forward_seen = key
            if prepared is not None:
                return torch.ops.evogp_hip.tree_evaluate_prepared(self.pop_size, self.max_tree_len, self.input_len, self.output_len,
                                                                  *self._tensors(), prepared[0], prepared[1], x.contiguous().to(torch.float32))
        return torch.ops.evogp_cuda.tree_evaluate(self.pop_size, self.max_tree_len, self.input_len, self.output_len,
                                                  *self._tensors(), x.contiguous().to(torch.float32))

    def batch_forward(self, x: Tensor) -> Tensor:
        """Shared input rows: x (batch, input_len) -> (pop, batch, output_len)."""
        x = check_tensor(x, self.batch_node_value.device)
        assert x.dim() == 2 and x.shape[1] == self.input_len, (
            f"x shape[1] should be {self.input_len}, but got {tuple(x.shape)}")
        return torch.ops.evogp_hip.tree_batch_evaluate(self.pop_size, x.shape[0], self.max_tree_len, self.input_len,
                                                       self.output_len, *self._tensors(),
                                                       x.contiguous().to(torch.float32))

    def SR_fitness(self, inputs: Tensor, labels: Tensor, use_MSE: bool = True, execute_mode: str = "auto") -> Tensor:
        """Mean squared / absolute error of every tree over the dataset: (pop,), positive."""
        inputs, labels = check_tensor(inputs, self.batch_node_value.device), check_tensor(labels, self.batch_node_value.device)
        n = inputs.shape[0]
        assert inputs.shape == (n, self.input_len), (
            f"inputs shape should be ({n}, {self.input_len}), but got {inputs.shape}")
        assert labels.shape == (n, self.output_len), (
            f"outputs shape should be ({n}, {self.output_len}), but got {labels.shape}")
        assert execute_mode in _SR_MODES, f"execute_mode should be one of {list(_SR_MODES)}, but got {execute_mode}"
        mask = self.func_mask
        if inputs.is_cuda and mask:
            # what this object knows beyond the tensors: the function set of the trees (the descriptors they came from)
            return torch.ops.evogp_hip.tree_SR_fitness_masked(self.pop_size, n, self.max_tree_len, self.input_len, self.output_len, use_MSE,
                                                              *self._tensors(), inputs.contiguous().to(torch.float32),
                                                              labels.contiguous().to(torch.float32), _SR_MODES[execute_mode], mask)
        return torch.ops.evogp_cuda.tree_SR_fitness(self.pop_size, n, self.max_tree_len, self.input_len,
                                                    self.output_len, use_MSE, *self._tensors(),
                                                    inputs.contiguous().to(torch.float32),
                                                    labels.contiguous().to(torch.float32), _SR_MODES[execute_mode])

    # ---- genetic operators --------------------------------------------------------------------
    def mutate(self, replace_pos: Tensor, new_sub_forest: "Forest") -> "Forest":
        """Replace the subtree at replace_pos[n] of tree n by the whole tree new_sub_forest[n]."""
        replace_pos = check_tensor(replace_pos, self.batch_node_value.device)
        assert replace_pos.shape == (self.pop_size,), (
            f"replace_pos shape should be ({self.pop_size}, ), but got {replace_pos.shape}")
        for attr in ("pop_size", "input_len", "output_len", "max_tree_len"):
            assert getattr(self, attr) == getattr(new_sub_forest, attr), (
                f"{attr} should be {getattr(self, attr)}, but got {getattr(new_sub_forest, attr)}")
        value, ntype, size = torch.ops.evogp_cuda.tree_mutate(
            self.pop_size, self.max_tree_len, *self._tensors(), replace_pos.contiguous().to(torch.int32),
            *new_sub_forest._tensors())
        return Forest(self.input_len, self.output_len, value, ntype, size, func_mask=Forest.join_masks(self.func_mask, new_sub_forest.func_mask))

    def crossover(self, left_indices: Tensor, right_indices: Tensor, left_pos: Tensor, right_pos: Tensor) -> "Forest":
        """out[n] = self[left_indices[n]] with subtree left_pos[n] replaced by subtree right_pos[n] of
        self[right_indices[n]]."""
        idx = [check_tensor(t, self.batch_node_value.device).contiguous().to(torch.int32) for t in (left_indices, right_indices, left_pos, right_pos)]
        n = idx[0].shape[0]
        for name, t in zip(("left_indices", "right_indices", "left_pos", "right_pos"), idx):
            assert t.shape == (n,), f"{name} shape should be ({n}, ), but got {t.shape}"
        value, ntype, size = torch.ops.evogp_cuda.tree_crossover(self.pop_size, n, self.max_tree_len,
                                                                 *self._tensors(), *idx)
        return Forest(self.input_len, self.output_len, value, ntype, size, func_mask=self.func_mask)

    # ---- container protocol -------------------------------------------------------------------
    def __getitem__(self, index):
        if isinstance(index, int) or (hasattr(index, "shape") and tuple(index.shape) == ()):
            return Tree(self.input_len, self.output_len, self.batch_node_value[index], self.batch_node_type[index],
                        self.batch_subtree_size[index])
        if isinstance(index, (slice, Tensor, np.ndarray)):
            return Forest(self.input_len, self.output_len, self.batch_node_value[index],
                          self.batch_node_type[index], self.batch_subtree_size[index], func_mask=self.func_mask)
        raise Exception(f"Do not support index type {type(index)}")

    def __setitem__(self, index, value):
        if isinstance(index, int):
            assert isinstance(value, Tree), f"value should be Tree when index is int, but got {type(value)}"
            self.batch_node_value[index] = value.node_value
            self.batch_node_type[index] = value.node_type
            self.batch_subtree_size[index] = value.subtree_size
        elif isinstance(index, (slice, Tensor, np.ndarray)):
            assert isinstance(value, Forest), f"value should be Forest when index is slice, but got {type(value)}"
            joined = Forest.join_masks(self.func_mask, value.func_mask)
            self.batch_node_value[index] = value.batch_node_value
            self.batch_node_type[index] = value.batch_node_type
            self.batch_subtree_size[index] = value.batch_subtree_size
            self._func_mask = (joined, self._forest_key()) if joined else None
        else:
            raise NotImplementedError

    def __iter__(self):
        for i in range(self.pop_size):
            yield self[i]

    def __len__(self):
        return self.pop_size

    def __add__(self, other):
        assert other.input_len == self.input_len and other.output_len == self.output_len
        if isinstance(other, Forest):
            parts = (other.batch_node_value, other.batch_node_type, other.batch_subtree_size)
        elif isinstance(other, Tree):
            parts = (other.node_value.unsqueeze(0), other.node_type.unsqueeze(0), other.subtree_size.unsqueeze(0))
        else:
            raise NotImplementedError
        return Forest(self.input_len, self.output_len,
                      torch.cat([self.batch_node_value, parts[0]], dim=0),
                      torch.cat([self.batch_node_type, parts[1]], dim=0),
                      torch.cat([self.batch_subtree_size, parts[2]], dim=0),
                      func_mask=Forest.join_masks(self.func_mask, other.func_mask) if isinstance(other, Forest) else 0)

    def __radd__(self, other):
        return self.__add__(other)

    def __str__(self):
        lines = [f"Forest(pop size: {self.pop_size})", "["]
        lines += [f"  {tree}, " for tree in self]
        lines.append("]")
        return "\n".join(lines)

    __repr__ = __str__

    # ---- pickling (numpy round trip, forest.py:476-499) ----------------------------------------
    def __getstate__(self):
        return {
            "input_len": self.input_len,
            "output_len": self.output_len,
            "batch_node_value": self.batch_node_value.cpu().numpy(),
            "batch_node_type": self.batch_node_type.cpu().numpy(),
            "batch_subtree_size": self.batch_subtree_size.cpu().numpy(),
        }

    def __setstate__(self, state):
        dev = _utils.default_device()
        self.input_len = state["input_len"]
        self.output_len = state["output_len"]
        self.pop_size, self.max_tree_len = state["batch_node_value"].shape
        self.batch_node_value = torch.from_numpy(state["batch_node_value"]).to(dev)
        self.batch_node_type = torch.from_numpy(state["batch_node_type"]).to(dev)
        self.batch_subtree_size = torch.from_numpy(state["batch_subtree_size"]).to(dev)
