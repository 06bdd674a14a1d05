"""SymbolicRegression — fitness = minus the mean squared/absolute error of every tree over a
dataset (src/evogp/problem/symbolic_regression.py:9-96).  ``execute_mode`` keeps the reference's
strings; every kernel mode maps onto the one fused HIP kernel, ``"torch"`` evaluates with
``Forest.batch_forward`` (the non-replicating batch op) and reduces in torch."""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

from ..tree import Forest, default_device
from .base import BaseProblem

_MODES = ["torch", "hybrid parallel", "data parallel", "tree parallel", "auto"]


class SymbolicRegression(BaseProblem):
    def __init__(self, datapoints: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                 func: Optional[Callable] = None, num_inputs: Optional[int] = None, num_data: Optional[int] = 100,
                 lower_bounds=-1, upper_bounds=1, execute_mode: str = "auto"):
        assert execute_mode in _MODES, f"execute_mode should be one of {_MODES}, but got {execute_mode}"
        self.execute_mode = execute_mode
        if datapoints is not None and labels is not None:
            self.datapoints, self.labels = datapoints, labels
            return
        assert func is not None and num_inputs is not None, (
            "func and num_inputs, must be provided when datapoints and labels are not provided")
        self.datapoints, self.labels = self.generate_data(func, num_inputs, num_data, lower_bounds, upper_bounds)

    @staticmethod
    def generate_data(func, num_inputs, num_data, lower_bounds, upper_bounds):
        dev = default_device()

        def as_bound(b):
            if isinstance(b, (int, float)):
                return torch.full((num_inputs,), float(b), device=dev)
            return torch.as_tensor(b, dtype=torch.float32, device=dev)

        lo, hi = as_bound(lower_bounds)[None, :], as_bound(upper_bounds)[None, :]
        inputs = torch.rand(num_data, num_inputs, device=dev) * (hi - lo) + lo
        outputs = torch.vmap(func)(inputs)
        if outputs.dim() == 1:
            outputs = outputs[:, None]
        return inputs, outputs

    def evaluate(self, forest: Forest, use_MSE: bool = True) -> Tensor:
        if self.execute_mode == "torch":
            pred = forest.batch_forward(self.datapoints)  # (pop, D, out)
            err = pred - self.labels[None, :, :]
            # mean over datapoints AND outputs, as the reference's torch mode (symbolic_regression.py:76-80)
            return -torch.mean(err**2 if use_MSE else err.abs(), dim=(1, 2))
        return -forest.SR_fitness(self.datapoints, self.labels, use_MSE, self.execute_mode)

    def scores(self, forest: Forest, use_MSE: bool = True) -> Tensor:
        """``evaluate`` with the NaN entries already at -inf (what StandardPipeline.step makes of them, pipeline/standard.py:41-43):
        on the device the sign and the scrub are ONE launch behind the fitness pass instead of four torch launches"""
        if self.execute_mode != "torch" and forest.batch_node_value.is_cuda:
            err = forest.SR_fitness(self.datapoints, self.labels, use_MSE, self.execute_mode)
            return torch.ops.evogp_hip.fitness_scores(err, True)
        f = self.evaluate(forest, use_MSE)
        return torch.where(torch.isnan(f), torch.full_like(f, float("-inf")), f)

    @property
    def problem_dim(self):
        return self.datapoints.shape[1]

    @property
    def solution_dim(self):
        return self.labels.shape[1]
