"""Why does the threaded code leave trees of an evolved example/uci_sr.py population to the register kernels?  Evolves the script's
population for GENS generations, then, on a sample: the operand-stack height of the compiled program (intermediate results only)
in the compiler's order (right operand first) and with the larger subtree first; the share the threaded code left marked."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, GeneticProgramming
from evogp_amd.algorithm.selection import TournamentSelection
from evogp_amd.tree import Forest, GenerateDescriptor

sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(int(os.environ.get("SEED", "42")))   # (the algorithm draws its word seed from torch's CPU generator: the trajectory)
GENS = int(os.environ.get("GENS", "30"))
POP = int(os.environ.get("POP", "100000"))
_, Xd, yd, X, y = bench.sr_inputs(0, 1000, dev)
desc = GenerateDescriptor(max_tree_len=512, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"],
                          max_layer_cnt=9, const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
algo = GeneticProgramming(Forest.random_generate(POP, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)),
                          DefaultCrossover(), DefaultMutation(0.1, desc.update(max_layer_cnt=4)), TournamentSelection(20, 0.5, 0.1))
neg = torch.full((POP,), float("-inf"), device=dev)


def depths(ty, sz):
    """(height in array-reverse order, height with the larger subtree first) of one tree's program: leaves are operands of their
    parents, only function results live on the stack"""
    n = int(sz[0])
    arity = np.where(ty[:n] == 2, 1, np.where(ty[:n] == 3, 2, np.where(ty[:n] == 4, 3, 0)))

    def need(i, larger_first):   # -> (peak while subtree i runs, relative to the height before it), iterative post-order
        stack = [(i, 0, [])]
        res = {}
        while stack:
            v, state, kids = stack.pop()
            if state == 0:
                ks = []
                c = v + 1
                for _ in range(arity[v]):
                    ks.append(c); c += int(sz[c])
                stack.append((v, 1, ks))
                for k in ks:
                    stack.append((k, 0, []))
            else:
                if arity[v] == 0:
                    res[v] = 0
                    continue
                order = list(reversed(kids))
                if larger_first and len(kids) == 2 and sz[kids[0]] > sz[kids[1]]:
                    order = kids
                h, peak = 0, 0
                for k in order:
                    if arity[k] > 0:
                        peak = max(peak, h + res[k]); h += 1
                res[v] = max(peak, h, 1)
        return res[i]
    return need(0, False), need(0, True)


for g in range(GENS + 1):
    if g in (0, 10, 20, GENS):
        f = algo.forest
        _lib.check(_lib.lib.evogp_hip_debug_profile(2), "profile")
        words = f.SR_fitness(Xd, yd).view(torch.int32)
        _lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile")
        left = ((words == 0x7FC0FEED) | (words == 0x7FC0BEEF) | (words == 0x7FC0DEED)).cpu().numpy()
        ty, sz = f.batch_node_type.cpu().numpy() & 0x7F, f.batch_subtree_size.cpu().numpy()
        pick = np.random.default_rng(g).choice(POP, 1500, replace=False)
        d = np.array([depths(ty[t], sz[t]) for t in pick])
        lens = sz[pick, 0]
        print(f"gen {g}: mean len {sz[:, 0].mean():.1f}, left to the register kernels {left.mean():.3f}; sample: height > 9 in array order "
              f"{(d[:, 0] > 9).mean():.3f} (max {d[:, 0].max()}), larger-first {(d[:, 1] > 9).mean():.3f} (max {d[:, 1].max()}); "
              f"left among height <= 9: {left[pick][d[:, 0] <= 9].mean() if (d[:, 0] <= 9).any() else float('nan'):.3f}, among height > 9: "
              f"{left[pick][d[:, 0] > 9].mean() if (d[:, 0] > 9).any() else float('nan'):.3f}; len>64 {np.mean(lens > 64):.3f}")
        w = words.cpu().numpy().view(np.uint32)
        kinds = {name: int((w == code).sum()) for name, code in (("heavy (compile time)", 0x7FC0FEED), ("general", 0x7FC0BEEF), ("run time", 0x7FC0DEED))}
        alllen = sz[:, 0]
        vals = f.batch_node_value.cpu().numpy()
        trig = ((ty == 2) & (np.arange(ty.shape[1])[None, :] < alllen[:, None])).sum(1)
        buckets = [(1, 64), (65, 128), (129, 256), (257, 384), (385, 512)]
        print("   sentinels:", kinds, "| left share by length:", {f"{a}-{b}": (round(float(left[(alllen >= a) & (alllen <= b)].mean()), 3), int(((alllen >= a) & (alllen <= b)).sum()))
                                                               for a, b in buckets if ((alllen >= a) & (alllen <= b)).any()},
              f"| unary nodes per tree: left {trig[left].mean() if left.any() else 0:.1f}, taken {trig[~left].mean():.1f}")
        if g == GENS and os.environ.get("DUMP") and left.any():   # some of the trees that were left, for a look at them on the host
            rows = np.nonzero(left)[0][:200]
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", "uci_left.npz"), ty=f.batch_node_type.cpu().numpy()[rows], va=vals[rows], sz=sz[rows],
                                words=w[rows])
    if g < GENS:
        fit = -algo.forest.SR_fitness(Xd, yd)
        algo.step(torch.where(torch.isnan(fit), neg, fit))
