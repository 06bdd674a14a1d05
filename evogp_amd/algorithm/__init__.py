"""evogp_amd.algorithm — genetic operators (reference: src/evogp/algorithm/): the default set, the rank / roulette /
tournament / truncation selections, the diversity and leaf-biased crossovers, the structural and point mutations."""
from .selection import (BaseSelection, BaseSelector, DefaultSelection, RankSelection, RankSelector, RouletteSelection,
                        RouletteSelector, TournamentSelection, TournamentSelector, TruncationSelection, TruncationSelector)
from .crossover import BaseCrossover, CombinedDefaultCrossover, DefaultCrossover, DiversityCrossover, LeafBiasedCrossover
from .mutation import (BaseMutation, CombinedDefaultMutation, CombinedMutation, DefaultMutation, DeleteMutation, HoistMutation, InsertMutation,
                       MultiConstMutation, MultiPointMutation, SingleConstMutation, SinglePointMutation)
from .genetic_programming import GeneticProgramming, ParetoFront

__all__ = ["BaseSelection", "DefaultSelection", "RankSelection", "RouletteSelection", "TournamentSelection",
           "TruncationSelection", "BaseSelector", "RankSelector", "RouletteSelector", "TournamentSelector",
           "TruncationSelector", "BaseCrossover", "DefaultCrossover", "DiversityCrossover", "LeafBiasedCrossover", "BaseMutation",
           "DefaultMutation", "HoistMutation", "InsertMutation", "DeleteMutation", "SinglePointMutation",
           "MultiPointMutation", "SingleConstMutation", "MultiConstMutation", "CombinedMutation", "CombinedDefaultCrossover",
           "CombinedDefaultMutation", "GeneticProgramming",
           "ParetoFront"]
