"""The headline through the reference's OWN operator: torch.ops.evogp_cuda.tree_SR_fitness (no function mask -- what the reference's
unchanged tree/forest.py:340-351 calls) against Forest.SR_fitness (which hands the engine the forest's function mask), on the headline
population and on configs[1], in one process: per-call time, bit equality of the fitness words, record memory after the unhinted
call.  Then a forest with unary functions: the first unhinted call (no observation yet: the arithmetic guess), the calls after it.
EVOGP_TC_LEARN=0 python scripts/unhinted_ab.py gives round 5's behaviour (generic line, three record arrays) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evogp_amd
import bench
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
set_default_device(dev)


def unhinted(f, X, y):
    return torch.ops.evogp_cuda.tree_SR_fitness(f.pop_size, X.shape[0], f.max_tree_len, f.input_len, f.output_len, True, f.batch_node_value,
                                                f.batch_node_type, f.batch_subtree_size, X, y, 4)


def timeit(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("EVOGP_TC_LEARN =", os.environ.get("EVOGP_TC_LEARN", "(default 1)"))
for n in (1_000_000, 100_000):
    forest, Xd, yd, _, _ = bench.sr_inputs(0, n, dev)
    first = unhinted(forest, Xd, yd); torch.cuda.synchronize()
    mb_first = evogp_amd.program_buffer_bytes() / 1e6
    for _ in range(45):
        unhinted(forest, Xd, yd)
    ms_u = min(timeit(lambda: unhinted(forest, Xd, yd), 20) for _ in range(3))
    mb_u = evogp_amd.program_buffer_bytes() / 1e6
    for _ in range(10):
        forest.SR_fitness(Xd, yd)
    ms_h = min(timeit(lambda: forest.SR_fitness(Xd, yd), 20) for _ in range(3))
    a, b = unhinted(forest, Xd, yd), forest.SR_fitness(Xd, yd)
    same = bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) and bool(torch.equal(a.view(torch.int32), first.view(torch.int32)))
    print(f"{n:>8} trees + - * /: reference op {ms_u:.4f} ms | hinted {ms_h:.4f} ms | gap {(ms_u / ms_h - 1) * 100:+.2f} % | words equal (first, later, hinted): {same} | "
          f"record memory after the first unhinted call {mb_first:.0f} MB, after all {mb_u:.0f} MB")
    del forest
    evogp_amd.release_workspaces()

# unary functions: the guess of the first call is wrong once
desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"], max_layer_cnt=6, const_samples=[-1, 0, 1])
f = Forest.random_generate(100_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
_, Xd, yd, _, _ = bench.sr_inputs(0, 8, dev)
calls = []
for i in range(4):
    calls.append(timeit(lambda: unhinted(f, Xd, yd), 1))
for _ in range(10):
    unhinted(f, Xd, yd)
ms_u = timeit(lambda: unhinted(f, Xd, yd), 20)
for _ in range(10):
    f.SR_fitness(Xd, yd)
ms_h = timeit(lambda: f.SR_fitness(Xd, yd), 20)
a, b = unhinted(f, Xd, yd), f.SR_fitness(Xd, yd)
ok = torch.isfinite(b)
rel = float(((a[ok] - b[ok]).abs() / b[ok].abs().clamp_min(1e-30)).max())
print(f"  100000 trees + - * / sin cos tan: unhinted calls 1-4 {', '.join(f'{c:.3f}' for c in calls)} ms; steady {ms_u:.4f} ms | hinted {ms_h:.4f} ms | "
      f"words equal {bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))} (max rel diff {rel:.2e}) | record memory {evogp_amd.program_buffer_bytes() / 1e6:.0f} MB")
