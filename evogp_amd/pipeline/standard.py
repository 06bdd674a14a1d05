"""StandardPipeline — evaluate, track the best tree, evolve; stop on a fitness target, a time limit
or a generation limit (src/evogp/pipeline/standard.py:11-106)."""
from __future__ import annotations

import time

import torch

from ..algorithm import GeneticProgramming
from ..problem import BaseProblem


class BasePipeline:
    def step(self):
        raise NotImplementedError

    def run(self):
        raise NotImplementedError


class StandardPipeline(BasePipeline):
    def __init__(self, algorithm: GeneticProgramming, problem: BaseProblem, fitness_target: float = None,
                 generation_limit: int = 100, time_limit: int = None, is_show_details: bool = True,
                 valid_fitness_boundry: float = 1e8):
        self.algorithm = algorithm
        self.problem = problem
        self.fitness_target = fitness_target
        self.generation_limit = generation_limit
        self.time_limit = time_limit
        self.is_show_details = is_show_details
        self.valid_fitness_boundry = valid_fitness_boundry
        self.best_tree = None
        self.best_fitness = float("-inf")
        self.fitness = None

    def step(self):
        """One generation.  Same result as the reference's loop (standard.py:41-52), but nothing waits on the device
        before the whole generation is enqueued: the NaN scrub is a select instead of a boolean-mask assignment (which
        hides a host sync), and the fitness vector is brought to the host AFTER ``algorithm.step`` has queued the
        selection / breeding kernels, so the copy overlaps them instead of idling the GPU."""
        forest = self.algorithm.forest  # step() builds a new forest; the best tree comes from this one
        scores = getattr(self.problem, "scores", None)
        if scores is not None:   # (a problem that hands over the scrubbed fitness itself: SymbolicRegression, one launch)
            fitness = scores(forest)
        else:
            fitness = self.problem.evaluate(forest)
            fitness = torch.where(torch.isnan(fitness), torch.full_like(fitness, float("-inf")), fitness)  # standard.py:43
        self.algorithm.step(fitness)
        host = fitness.cpu()
        best = int(torch.argmax(host))
        if host[best] > self.best_fitness:
            self.best_fitness = host[best]
            self.best_tree = forest[best]
        return host

    def run(self):
        start = time.time()
        generation = 0
        while True:
            tic = time.time()
            self.fitness = self.step()
            if self.is_show_details:
                self.show_details(tic, generation, self.fitness)
            if self.fitness_target is not None and self.best_fitness >= self.fitness_target:
                print(f"stopped: best fitness {float(self.best_fitness):.6g} reached the target {self.fitness_target}")
                break
            if self.time_limit is not None and time.time() - start > self.time_limit:
                print(f"stopped: {self.time_limit} s time limit")
                break
            generation += 1
            if generation >= self.generation_limit:
                print(f"stopped: {self.generation_limit} generations")
                break
        return self.best_tree

    def show_details(self, tic, generation, fitness):
        b = self.valid_fitness_boundry
        valid = fitness[(fitness < b) & (fitness > -b)]
        ms = (time.time() - tic) * 1000
        line = f"gen {generation:4d}  {ms:8.2f} ms  {len(valid)} of {len(fitness)} finite"
        if len(valid):
            line += (f"  best {float(valid.max()):.4f}  mean {float(valid.mean()):.4f} +- {float(valid.std(unbiased=False)):.4f}"
                     f"  worst {float(valid.min()):.4f}")
        print(line)
