"""the pieces of one default generation step outside the fitness call, timed one by one (HIP events), at a few population sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd.tree import Forest, GenerateDescriptor

dev = torch.device("cuda", 0)


def timed(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6


for pop in [int(a) for a in sys.argv[1:]] or [125_000, 1_000_000]:
    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
    d = desc.update(max_layer_cnt=3)
    keys = torch.tensor([42, 0], dtype=torch.uint32, device=dev)
    f = Forest.random_generate(pop, desc, keys=keys)
    fit = torch.randn(pop, device=dev)
    n_elite, n_surv = pop // 100, pop * 3 // 10
    n_new = pop - n_elite
    below = int(0.2 * (2**31 - 1))
    value, ntype, size = f._tensors()
    order = torch.ops.evogp_hip.select_survivors(fit, n_elite, n_surv)
    rnd = torch.randint(0, 2**31 - 1, (6, n_new), dtype=torch.int32, device=dev)
    gen = lambda: torch.ops.evogp_hip.tree_generate_masked(n_new, 64, d.input_len, d.output_len, d.const_samples.shape[0], d.out_prob, d.const_prob, keys,
                                                           d.depth2leaf_probs, d.roulette_funcs, d.const_samples, 0, rnd[4], below)
    donors = gen()
    print(f"pop {pop}: generate(full) {timed(lambda: Forest.random_generate(pop, desc, keys=keys)):.1f} us | select {timed(lambda: torch.ops.evogp_hip.select_survivors(fit, n_elite, n_surv)):.1f}"
          f" | randint {timed(lambda: torch.randint(0, 2**31 - 1, (6, n_new), dtype=torch.int32, device=dev)):.1f} | donors (20 % live, depth 3) {timed(gen):.1f}"
          f" | breed {timed(lambda: torch.ops.evogp_hip.breed_default(pop, 64, n_elite, n_surv, value, ntype, size, order, rnd, below, *donors, False)):.1f}")
    # the breeding pass that also compiles its rows (needs the geometry of a fitness call), and what the next fitness call then costs
    import numpy as np
    rng = np.random.default_rng(1234)
    Xn = rng.uniform(-5, 5, (1024, 10)).astype(np.float32)
    X = torch.from_numpy(Xn).to(dev); y = torch.from_numpy((Xn[:, 0] * Xn[:, 1] + Xn[:, 2] * Xn[:, 3] - Xn[:, 4] + 0.5 * Xn[:, 5] ** 2)[:, None].copy()).to(dev)
    f.SR_fitness(X, y)
    elites, parents = order[:n_elite], order[:n_surv]
    bc = lambda: torch.ops.evogp_hip.breed_rows_compiled(pop, 64, value, ntype, size, elites, parents, rnd, below, *donors, 0, pop)
    t_bc = timed(bc)
    nv, nt, ns, stamp = bc()
    child = Forest(f.input_len, f.output_len, nv, nt, ns)
    t_plain = timed(lambda: child.SR_fitness(X, y))
    def ahead():
        v2, t2, s2, st = bc()
        return Forest(f.input_len, f.output_len, v2, t2, s2).set_compiled_records(st).SR_fitness(X, y)
    t_pair = timed(ahead)
    print(f"           breed + compile {t_bc:.1f} us (stamp {stamp}) | fitness of the children, compiled in the call {t_plain:.1f} us | breed + compile + fitness {t_pair:.1f} us"
          f" (separately: {timed(lambda: torch.ops.evogp_hip.breed_rows(pop, 64, value, ntype, size, elites, parents, rnd, below, *donors, 0, pop)) + t_plain:.1f})")
