"""evogp_amd.algorithm — the default genetic operator set (reference: src/evogp/algorithm/)."""
from .selection import BaseSelection, DefaultSelection
from .crossover import BaseCrossover, DefaultCrossover
from .mutation import BaseMutation, DefaultMutation
from .genetic_programming import GeneticProgramming, ParetoFront

__all__ = ["BaseSelection", "DefaultSelection", "BaseCrossover", "DefaultCrossover", "BaseMutation",
           "DefaultMutation", "GeneticProgramming", "ParetoFront"]
