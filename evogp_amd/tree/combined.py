"""CombinedForest / CombinedTree — several single-output forests evolved side by side, one per named output, each
reading its own subset of named input columns (reference: src/evogp/tree/combined_forest.py:13-157,
combined_tree.py:7-52).  ``data_info`` maps an output name to the list of input names its forest consumes, in the order
of that forest's variables.  Every sub-forest is an ordinary :class:`Forest`, so evaluation runs on the same native
kernels; this layer only routes columns and indices."""
from __future__ import annotations

from typing import Dict, List, Union

import numpy as np
import torch
from torch import Tensor

from .descriptor import GenerateDescriptor
from .forest import Forest


def _names(data_info: Dict[str, List[str]]):
    outputs = list(data_info.keys())
    inputs: List[str] = []
    for cols in data_info.values():
        for c in cols:
            if c not in inputs:
                inputs.append(c)
    return outputs, inputs


def _columns(x: Dict[str, Tensor], cols: List[str]) -> Tensor:
    """stack the named columns: tensors of shape (n,) -> (n, len(cols))"""
    return torch.stack([x[c] for c in cols], dim=1).to(torch.float32)


class CombinedTree:
    def __init__(self, trees, data_info: Dict[str, List[str]]):
        assert len(trees) == len(data_info), f"{len(data_info)} outputs but {len(trees)} trees"
        self.trees = list(trees)
        self.data_info = data_info
        self.output_names, self.input_names = _names(data_info)
        self.input_len, self.output_len = len(self.input_names), len(self.output_names)
        for name, tree in zip(self.output_names, self.trees):
            setattr(self, name, tree)  # best.A, best.B ... as in the reference (combined_tree.py:21-22)

    @staticmethod
    def random_generate(descriptors: Union[List[GenerateDescriptor], GenerateDescriptor], data_info: Dict[str, List[str]]):
        return CombinedForest.random_generate(pop_size=1, data_info=data_info, descriptors=descriptors)[0]

    def to_combined_forest(self) -> "CombinedForest":
        return CombinedForest([t.to_forest() for t in self.trees], self.data_info)

    def forward(self, x: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """x[name]: scalar tensors (one input row) or (n,) columns (a batch of rows)"""
        first = next(iter(x.values()))
        if first.dim() == 0:
            res = self.to_combined_forest().forward({k: v.reshape(1) for k, v in x.items()})
        else:
            res = self.to_combined_forest().batch_forward(x)
        return {k: v[0] for k, v in res.items()}  # drop the population axis

    def __str__(self):
        return "\n".join(f"{n} = {t}" for n, t in zip(self.output_names, self.trees))


class CombinedForest:
    def __init__(self, forests: List[Forest], data_info: Dict[str, List[str]]):
        assert len(forests) == len(data_info), f"{len(data_info)} outputs but {len(forests)} forests"
        assert len({f.pop_size for f in forests}) == 1, "all sub-forests must have the same population size"
        self.forests = list(forests)
        self.data_info = data_info
        self.output_names, self.input_names = _names(data_info)
        self.input_len, self.output_len = len(self.input_names), len(self.output_names)
        self.pop_size = forests[0].pop_size

    @staticmethod
    def random_generate(pop_size: int, data_info: Dict[str, List[str]],
                        descriptors: Union[List[GenerateDescriptor], GenerateDescriptor]) -> "CombinedForest":
        if isinstance(descriptors, GenerateDescriptor):
            descriptors = [descriptors] * len(data_info)
        assert isinstance(descriptors, list) and len(descriptors) == len(data_info), (
            f"there are {len(data_info)} sub_forests, but got {len(descriptors)} descriptors")
        for d, cols in zip(descriptors, data_info.values()):
            assert d.input_len == len(cols), "input size not match"
            assert d.output_len == 1, "output size must be 1"
        return CombinedForest([Forest.random_generate(pop_size=pop_size, descriptor=d) for d in descriptors], data_info)

    # x[name]: (pop,) -> {output: (pop, 1)}      one input row per individual (combined_forest.py:60-72)
    def forward(self, x: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {name: f.forward(_columns(x, self.data_info[name])) for name, f in zip(self.output_names, self.forests)}

    # x[name]: (n,) -> {output: (pop, n, 1)}     a shared batch of rows (combined_forest.py:75-87)
    def batch_forward(self, x: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {name: f.batch_forward(_columns(x, self.data_info[name])) for name, f in zip(self.output_names, self.forests)}

    def __getitem__(self, index):
        if isinstance(index, (int, np.integer)):
            return CombinedTree([f[int(index)] for f in self.forests], self.data_info)
        if isinstance(index, (slice, Tensor, np.ndarray)):
            return CombinedForest([f[index] for f in self.forests], self.data_info)
        raise TypeError(f"unsupported index type {type(index)}")

    def __setitem__(self, index, value):
        if isinstance(index, (int, np.integer)):
            assert isinstance(value, CombinedTree), f"value should be CombinedTree when index is int, but got {type(value)}"
            for f, t in zip(self.forests, value.trees):
                f[int(index)] = t
        elif isinstance(index, (slice, Tensor, np.ndarray)):
            assert isinstance(value, CombinedForest), f"value should be CombinedForest, but got {type(value)}"
            for f, g in zip(self.forests, value.forests):
                f[index] = g
        else:
            raise TypeError(f"unsupported index type {type(index)}")

    def __iter__(self):
        return (self[i] for i in range(self.pop_size))

    def __len__(self):
        return self.pop_size

    def __add__(self, other):
        assert self.data_info == other.data_info, "cannot concatenate combined forests of different layouts"
        if isinstance(other, CombinedForest):
            return CombinedForest([a + b for a, b in zip(self.forests, other.forests)], self.data_info)
        if isinstance(other, CombinedTree):
            return CombinedForest([a + t for a, t in zip(self.forests, other.trees)], self.data_info)
        return NotImplemented

    __radd__ = __add__
