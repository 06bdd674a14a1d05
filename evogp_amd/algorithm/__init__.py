"""evogp_amd.algorithm — genetic operators (reference: src/evogp/algorithm/): the default set and the structural /
point mutations."""
from .selection import BaseSelection, DefaultSelection
from .crossover import BaseCrossover, DefaultCrossover
from .mutation import (BaseMutation, CombinedMutation, DefaultMutation, DeleteMutation, HoistMutation, InsertMutation,
                       MultiConstMutation, MultiPointMutation, SingleConstMutation, SinglePointMutation)
from .genetic_programming import GeneticProgramming, ParetoFront

__all__ = ["BaseSelection", "DefaultSelection", "BaseCrossover", "DefaultCrossover", "BaseMutation",
           "DefaultMutation", "HoistMutation", "InsertMutation", "DeleteMutation", "SinglePointMutation",
           "MultiPointMutation", "SingleConstMutation", "MultiConstMutation", "CombinedMutation", "GeneticProgramming",
           "ParetoFront"]
