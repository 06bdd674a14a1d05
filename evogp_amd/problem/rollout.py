"""RolloutProblem — policy-tree rollouts in a batched environment.

The reference's BraxProblem / MujocoProblem / GenesisProblem (src/evogp/problem/brax_problem.py:54-93) all run the same
loop: ``max_episode_length`` times  obs -> Forest.forward -> output_transform -> env.step, one environment per tree,
accumulate reward until done.  Their simulators (jax / brax / mujoco / genesis) are not part of this image, so this
module provides the loop itself behind the same interface plus two small vectorised torch environments to drive it:

* the loop is environment-agnostic: anything with ``reset(n) -> state``, ``observe(state) -> (n, obs_dim)`` and
  ``step(state, action) -> (state, reward, done)`` written with torch ops works (DLPack hand-off to another
  framework goes where the reference has it: around ``observe`` / ``step``);
* with ``use_graph=True`` (default on a GPU) ONE step of the loop — the tree_evaluate launches and the environment's
  elementwise kernels — is captured into a HIP graph and replayed: at 50 k trees a step is ~55 us of tree
  evaluation plus a dozen tiny environment kernels, and replaying removes the per-launch host cost of all of them
  (the rollout is launch-latency bound).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

from ..tree import Forest
from .base import BaseProblem


class LinearTrackingEnv:
    """A deterministic linear system the policy has to regulate to the origin: x' = A x + B a, reward = -|x|^2 - c|a|^2,
    done when |x|_inf > limit.  obs = x.  Sized like halfcheetah by default (17 observations, 6 actions)."""

    def __init__(self, obs_dim: int = 17, act_dim: int = 6, seed: int = 0, limit: float = 50.0, device=None, randomize: float = 0.0):
        """``randomize`` > 0: every tree starts from its own perturbed state, a counter-based function of (seed, GLOBAL tree
        index) — the rank of a sharded run that owns trees [offset, offset + n) resets with ``reset(n, offset)`` and gets exactly
        the episodes those trees have in a single-process run (the reference's simulators are reset per environment the same
        way, brax_problem.py:60-66)."""
        self.randomize, self.seed = float(randomize), int(seed)
        g = torch.Generator().manual_seed(seed)
        a = torch.randn(obs_dim, obs_dim, generator=g) * (0.6 / obs_dim**0.5)
        self.A = (torch.eye(obs_dim) * 0.97 + a).to(device)
        self.B = (torch.randn(act_dim, obs_dim, generator=g) * 0.3).to(device)
        self.x0 = torch.randn(obs_dim, generator=g).to(device)
        self.obs_dim, self.act_dim, self.limit, self.device = obs_dim, act_dim, limit, device

    def reset(self, n: int, offset: int = 0) -> Tensor:
        x = self.x0[None, :].repeat(n, 1)      # randomize == 0: every tree faces the same episode, fitness depends on the tree only
        if self.randomize > 0:
            from ..parallel import random_words   # word k of tree i = hash(seed, 0, k, i): the same on every rank and for every world size

            w = random_words(self.seed, 0, self.obs_dim, offset, offset + n, x.device)          # (obs_dim, n) in [0, 2^31 - 1)
            x = x + self.randomize * (w.t().to(torch.float32) / float(2**30) - 1.0)
        return x

    def observe(self, state: Tensor) -> Tensor:
        return state

    def step(self, state: Tensor, action: Tensor):
        nxt = state @ self.A + action @ self.B
        reward = -(nxt * nxt).sum(1) - 0.1 * (action * action).sum(1)
        done = nxt.abs().amax(1) > self.limit
        return nxt, reward, done


class PendulumEnv:
    """The classic torque-limited pendulum swing-up, batched: obs = (cos th, sin th, thdot), one action."""

    obs_dim, act_dim = 3, 1

    def __init__(self, device=None, dt: float = 0.05):
        self.device, self.dt = device, dt

    def reset(self, n: int) -> Tensor:
        th = torch.full((n,), 3.0, device=self.device)          # hanging (almost) straight down
        return torch.stack([th, torch.zeros_like(th)], dim=1)

    def observe(self, state: Tensor) -> Tensor:
        th, thd = state[:, 0], state[:, 1]
        return torch.stack([torch.cos(th), torch.sin(th), thd], dim=1)

    def step(self, state: Tensor, action: Tensor):
        th, thd = state[:, 0], state[:, 1]
        u = 2.0 * action[:, 0].clamp(-1, 1)
        ang = torch.remainder(th + torch.pi, 2 * torch.pi) - torch.pi
        reward = -(ang * ang + 0.1 * thd * thd + 0.001 * u * u)
        thd = (thd + (15.0 * torch.sin(th) + 3.0 * u) * self.dt).clamp(-8, 8)
        return torch.stack([th + thd * self.dt, thd], dim=1), reward, torch.zeros_like(th, dtype=torch.bool)


class RolloutProblem(BaseProblem):
    """fitness[t] = total reward of tree t acting as the policy of its own copy of ``env`` for ``max_episode_length``
    steps (an environment that is done stops collecting reward, brax_problem.py:81-88)."""

    def __init__(self, env, max_episode_length: int, output_transform: Callable = torch.tanh, use_graph: Optional[bool] = None):
        self.env = env
        self.max_episode_length = max_episode_length
        self.output_transform = output_transform
        self.use_graph = use_graph

    def _step(self, forest: Forest, state: Tensor, total: Tensor, done: Tensor):
        action = self.output_transform(forest.forward(self.env.observe(state)))
        nxt, reward, now_done = self.env.step(state, action)
        reward = torch.nan_to_num(reward, nan=-1e6, posinf=-1e6, neginf=-1e6)   # a NaN action ends as a large penalty
        total = total + torch.where(done, torch.zeros_like(reward), reward)
        done = done | now_done | ~torch.isfinite(nxt).all(1)
        nxt = torch.where(done[:, None], state, torch.nan_to_num(nxt))
        return nxt, total, done

    def evaluate(self, forest: Forest, tree_index_offset: int = 0) -> Tensor:
        """``tree_index_offset``: global index of the forest's first tree when it is a rank's shard of a larger population
        (evogp_amd/parallel.py): environments that reset per tree are then reset for exactly those trees.  No collective inside
        the episode: every rank rolls out its own trees; the fitness values meet in ShardedGeneticProgramming.step."""
        n = forest.pop_size
        state = self.env.reset(n, tree_index_offset) if tree_index_offset else self.env.reset(n)
        dev = state.device
        total = torch.zeros(n, device=dev)
        done = torch.zeros(n, dtype=torch.bool, device=dev)
        graph = self.use_graph if self.use_graph is not None else dev.type == "cuda"
        if not graph:
            forest.prepare_forward()
            for _ in range(self.max_episode_length):
                state, total, done = self._step(forest, state, total, done)
            return total
        # capture ONE step on static buffers, replay it max_episode_length times.  A multi-output forest is decoded once into its
        # operation lists first (Forest.prepare_forward: one host sync, not allowed inside the capture): every replayed step
        # then runs evaluate_prepared.hip instead of re-interpreting the three tree arrays
        forest.prepare_forward()
        s_state, s_total, s_done = state.clone(), total.clone(), done.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up outside capture (allocator, lazy initialisation)
            self._step(forest, s_state, s_total, s_done)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # thread_local: in a multi-rank process the RCCL watchdog thread queries events while this thread captures; under the
        # default ("global") that invalidates the capture although the two have nothing to do with each other
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            a, b, c = self._step(forest, s_state, s_total, s_done)
            s_state.copy_(a); s_total.copy_(b); s_done.copy_(c)
        for _ in range(self.max_episode_length):
            g.replay()
        return s_total.clone()

    @property
    def problem_dim(self):
        return self.env.obs_dim

    @property
    def solution_dim(self):
        return self.env.act_dim
