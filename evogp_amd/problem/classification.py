"""Classification — fitness = accuracy of every tree over a labelled dataset
(src/evogp/problem/classification.py:9-83).  ``multi_output=True``: the tree has one output per class and predicts the
arg-max of the soft-maxed outputs; ``multi_output=False``: the single output is rounded onto the label range.  The
outputs come from ``Forest.batch_forward`` — here the non-replicating batch op, so a population of
200 k trees on the 1797-point digits set does not have to be replicated 1797 times as in the reference — and are
reduced tree-block by tree-block so that the (pop, D, classes) tensor never exists in full."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor

from ..tree import Forest, default_device
from .base import BaseProblem

_SKLEARN = ("iris", "wine", "breast_cancer", "digits")


class Classification(BaseProblem):
    def __init__(self, datapoints: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                 dataset: Optional[str] = None, multi_output: bool = True, block_bytes: int = 1 << 30):
        self.multi_output = multi_output
        self.block_bytes = block_bytes
        if datapoints is not None and labels is not None:
            self.datapoints, self.labels = datapoints, labels
        else:
            assert dataset is not None, "dataset must be provided when datapoints and labels are not provided"
            self.datapoints, self.labels = self.generate_data(dataset)
        self.maximum = int(torch.max(self.labels))

    @staticmethod
    def generate_data(dataset: str):
        assert dataset in _SKLEARN, "Invalid dataset"
        from sklearn import datasets  # offline toy sets only (classification.py:35-48)

        X, y = getattr(datasets, f"load_{dataset}")(return_X_y=True)
        dev = default_device()
        return (torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(y, dtype=torch.float32, device=dev))

    def transform(self, x: Tensor) -> Tensor:
        return torch.clamp(torch.round(x + self.maximum / 2), 0, self.maximum).squeeze(-1)

    def _accuracy(self, outputs: Tensor) -> Tensor:
        if not self.multi_output:
            pred = self.transform(outputs)
        else:
            eps = 1e-15
            pred = torch.argmax(torch.clip(torch.softmax(outputs, dim=2), eps, 1 - eps), dim=2)
        return torch.sum(pred == self.labels, dim=1, dtype=torch.float32) / self.labels.shape[0]

    def _int_labels(self):
        """int32 copy of the labels when every label is integral, else None (checked once: one host sync per problem)"""
        if not hasattr(self, "_labels_i32"):
            lab = self.labels
            ok = bool(torch.all(lab == torch.round(lab))) if lab.is_floating_point() else True
            self._labels_i32 = lab.to(torch.int32).contiguous() if ok else None
        return self._labels_i32

    def evaluate(self, forest: Forest) -> Tensor:
        D = self.datapoints.shape[0]
        if (self.multi_output and self.datapoints.is_cuda and 2 <= forest.output_len <= 16
                and forest.input_len * 256 <= 150 * 1024 and os.environ.get("EVOGP_FUSED_ACCURACY", "1") != "0"
                and self._int_labels() is not None):
            # fused epilogue: only the per-tree count of correct rows leaves the kernel (csrc/sr_tc.hip END_CLS, csrc/sr_wide.hip).
            # The kernel compares the arg-max with an int32 label: only taken for integral labels (a label like 1.5 never equals
            # a prediction in the reference's `pred == labels`, classification.py:62-75; a truncating cast would make it class 1).
            # The count EQUALS the reference's: argmax(clip(softmax(.))) is the raw arg-max except where an output in front of the
            # maximum is within ~1e-7 of it (their fp32 soft-max values may round to the same float, and torch returns the first);
            # trees with such a row are recounted with aten's soft-max arithmetic (csrc/interp.hpp argmax_as_torch).
            counts = torch.ops.evogp_hip.tree_batch_argmax_count(
                forest.pop_size, D, forest.max_tree_len, forest.input_len, forest.output_len, *forest._tensors(),
                self.datapoints.contiguous().to(torch.float32), self._int_labels())
            return counts.to(torch.float32) / D
        per_tree = 4 * D * max(forest.output_len, 1) * 3        # outputs + soft-max + clip temporaries
        step = max(1, min(forest.pop_size, self.block_bytes // per_tree))
        if step >= forest.pop_size:
            return self._accuracy(forest.batch_forward(self.datapoints))
        parts = [self._accuracy(forest[i:i + step].batch_forward(self.datapoints)) for i in range(0, forest.pop_size, step)]
        return torch.cat(parts)

    @property
    def problem_dim(self):
        return self.datapoints.shape[1]

    @property
    def solution_dim(self):
        return self.maximum + 1 if self.multi_output else 1
