"""CustomLoss — fitness = minus a user-supplied loss of the outputs of a CombinedForest (reference:
src/evogp/problem/custom_loss.py:8-33).  ``loss_func``'s parameter names are matched against ``existing_data`` (fixed
named columns, passed as they are) and the output names of the combined forest (one (D,) column per individual); the
loss is vmapped over the population axis of the latter."""
from __future__ import annotations

import inspect
from typing import Callable, Dict

import torch
from torch import Tensor

from .base import BaseProblem


def inspect_function(func: Callable):
    """names of the positional parameters of ``func`` (tree/utils.py:313-323)"""
    assert callable(func), "formula should be Callable"
    params = inspect.signature(func).parameters
    assert len(params) > 0, "formula should have at least one parameter"
    for name, p in params.items():
        assert p.default is inspect.Parameter.empty, f"formula should not have default parameters, but got {name}={p.default}"
    return list(params.keys())


class CustomLoss(BaseProblem):
    def __init__(self, existing_data: Dict[str, Tensor], loss_func: Callable):
        self.existing_data = existing_data
        self.loss_func = loss_func
        self.loss_parameters = inspect_function(loss_func)
        self.in_dims = tuple(None if n in existing_data else 0 for n in self.loss_parameters)
        assert any(d == 0 for d in self.in_dims), "the loss must take at least one output of the forest"
        self.vmap_loss_func = torch.vmap(loss_func, in_dims=self.in_dims, out_dims=0)

    def evaluate(self, forest) -> Tensor:
        outputs = forest.batch_forward(self.existing_data)  # {output name: (pop, D, 1)}
        args = [self.existing_data[n] if n in self.existing_data else outputs[n].squeeze(-1) for n in self.loss_parameters]
        return -self.vmap_loss_func(*args)  # negative loss as fitness
