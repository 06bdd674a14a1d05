"""evogp_amd.problem — fitness evaluation front ends (reference: src/evogp/problem/)."""
from .base import BaseProblem
from .symbolic_regression import SymbolicRegression
from .classification import Classification
from .transformation import Transformation
from .custom_loss import CustomLoss
from .rollout import LinearTrackingEnv, PendulumEnv, RolloutProblem

__all__ = ["BaseProblem", "SymbolicRegression", "Classification", "Transformation", "CustomLoss", "RolloutProblem", "LinearTrackingEnv", "PendulumEnv"]
