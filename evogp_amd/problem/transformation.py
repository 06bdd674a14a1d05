"""Transformation — feature construction: fitness = |Pearson correlation| between a tree's output over a dataset and the
regression target (reference: src/evogp/problem/transformation.py:11-69).  The outputs come from ``Forest.batch_forward``
(the non-replicating batch op); the reduction is the reference's formula, literally — including that
the outputs are centred with the mean over the WHOLE population (transformation.py:38), so one tree with a NaN output
turns every fitness into NaN exactly as it does there.  ``per_tree_mean=True`` centres every tree with its own mean
(the textbook correlation) and scores non-finite trees 0 instead."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ..tree import Forest, default_device
from .base import BaseProblem


class Transformation(BaseProblem):
    def __init__(self, datapoints: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                 dataset: Optional[str] = None, per_tree_mean: bool = False):
        self.per_tree_mean = per_tree_mean
        if datapoints is not None and labels is not None:
            self.datapoints, self.labels = datapoints, labels
        else:
            assert dataset is not None, "dataset must be provided when datapoints and labels are not provided"
            self.datapoints, self.labels = self.generate_data(dataset)

    @staticmethod
    def generate_data(dataset: str):
        if dataset != "diabetes":
            raise ValueError("Invalid dataset")
        from sklearn.datasets import load_diabetes  # offline toy set (transformation.py:27-33)

        X, y = load_diabetes(return_X_y=True)
        dev = default_device()
        return torch.tensor(X, dtype=torch.float32, device=dev), torch.tensor(y, dtype=torch.float32, device=dev)

    def _outputs(self, forest: Forest) -> Tensor:
        return forest.batch_forward(self.datapoints)[:, :, 0]  # (pop, D)

    def evaluate(self, forest: Forest) -> Tensor:
        out = self._outputs(forest)
        labels = self.labels.reshape(-1).to(out.dtype)
        lab = labels - torch.mean(labels)
        if self.per_tree_mean:
            o = out - torch.mean(out, dim=1, keepdim=True)
        else:
            o = out - torch.mean(out)
        corr = torch.sum(o * lab, dim=1) / torch.sqrt(torch.sum(o**2, dim=1) * torch.sum(lab**2))
        corr = torch.abs(corr)
        if self.per_tree_mean:
            corr = torch.where(torch.isfinite(corr), corr, torch.zeros_like(corr))
        return corr

    def new_feature(self, forest: Forest, n_best: int, n_features: int) -> Tensor:
        """Outputs of ``n_features`` of the ``n_best`` trees as new columns (D, n_features): starting from the n_best
        fittest, the tree taking part in the currently most correlated pair is dropped until n_features remain
        (transformation.py:45-69; of the pair, the one ranked worse goes)."""
        fitness = self.evaluate(forest)
        best = torch.argsort(torch.nan_to_num(fitness, nan=-1.0), descending=True)[:n_best]
        feats = self._outputs(forest[best])  # (n_best, D)
        corr = torch.abs(torch.corrcoef(feats))
        corr = torch.nan_to_num(corr, nan=0.0)
        corr.fill_diagonal_(0.0)
        keep = torch.ones(best.shape[0], dtype=torch.bool, device=feats.device)
        while int(keep.sum()) > n_features:
            flat = int(torch.argmax(corr))
            worst = max(flat // corr.shape[1], flat % corr.shape[1])
            keep[worst] = False
            corr[worst, :] = 0.0
            corr[:, worst] = 0.0
        return feats[keep].T
