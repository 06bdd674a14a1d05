"""Soak of tree_SR_fitness across population sizes, function sets, row lengths and dataset sizes: every configuration is called through the
reference's operator (no function mask) four times in a row, then under the forest's mask, then on a second stream while the first
stream repeats the call; all fitness words must be equal.  One line per configuration with a checksum of the words: run the script
under different work distributions (EVOGP_TC_STATIC=0 -- every batch from the XCD counters, as in round 5 --, the default pools,
EVOGP_TC_DYNSHIFT) and diff the logs: a tree's fitness is computed by one workgroup whichever one draws it, so the logs must be identical.
SOAK_CASES (default 60), SOAK_SEED (default 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import evogp_amd  # noqa: F401
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
set_default_device(dev)
SETS = [["+", "-", "*", "/"], ["+", "-", "*", "/", "sin", "cos", "tan"], ["+", "-", "*", "/", "neg", "abs", "sqrt"], ["+", "-", "*", "/", "max", "min", "if"],
        ["+", "*", "exp", "log", "inv"]]
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
cases = int(os.environ.get("SOAK_CASES", "60"))
side = torch.cuda.Stream()


def unhinted(f, X, y):
    return torch.ops.evogp_cuda.tree_SR_fitness(f.pop_size, X.shape[0], f.max_tree_len, f.input_len, f.output_len, True, f.batch_node_value,
                                                f.batch_node_type, f.batch_subtree_size, X, y, 4)


def words(t):
    return torch.where(torch.isnan(t), torch.full_like(t, 7.0), t).view(torch.int32)


bad = 0
for c in range(cases):
    pop = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20_000)), int(rng.integers(20_000, 150_000)), int(rng.integers(150_000, 450_000))]))
    funcs = SETS[int(rng.integers(0, len(SETS)))]
    L = int(rng.choice([16, 32, 64, 64, 128]))
    D = int(rng.choice([8, 100, 256, 1000, 1024, 1024, 2000]))
    V = int(rng.integers(1, 12))
    ar = 3 if "if" in funcs else 2
    top = max(m for m in range(1, 9) if (ar ** m - 1) // (ar - 1) <= L)   # (the descriptor refuses depths whose full tree overflows the row)
    mlc = int(rng.integers(max(2, top - 2), top + 1))
    desc = GenerateDescriptor(max_tree_len=L, input_len=V, output_len=1, using_funcs=funcs, max_layer_cnt=mlc, const_samples=[-1, 0, 1, 0.5])
    f = Forest.random_generate(pop, desc, keys=torch.tensor([int(rng.integers(1, 1 << 30)), c], dtype=torch.uint32, device=dev))
    g = torch.Generator(device="cpu").manual_seed(c)
    X = (torch.rand((D, V), generator=g) * 10 - 5).to(dev)
    y = (torch.rand((D, 1), generator=g) * 4 - 2).to(dev)
    outs = [unhinted(f, X, y) for _ in range(4)]
    outs.append(f.SR_fitness(X, y))
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        outs.append(unhinted(f, X, y))
        outs.append(f.SR_fitness(X, y))
    outs.append(unhinted(f, X, y))          # concurrently with the side stream's calls
    torch.cuda.synchronize()
    w = [words(o) for o in outs]
    same = all(bool(torch.equal(w[0], x)) for x in w[1:])
    bad += 0 if same else 1
    s = int(w[0].to(torch.int64).sum()); x = int(torch.bitwise_xor(w[0][::2][: w[0].numel() // 2], w[0][1::2][: w[0].numel() // 2]).to(torch.int64).sum()) if pop > 1 else 0
    print(f"case {c:3d} pop {pop:7d} L {L:3d} D {D:4d} V {V:2d} depth {mlc} funcs {'|'.join(funcs):28s} equal {same}  sum {s} mix {x}", flush=True)
    del f
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
