"""BASELINE configs[3] (classifier, pop 200 k, digits): fitness passes of generation 0 and of an evolved population, for
rocprofv3 --kernel-trace --stats (which kernels the fitness pass consists of)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import evogp_amd  # noqa: F401
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
from evogp_amd.problem import Classification
from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

dev = torch.device("cuda", 0); torch.cuda.set_device(0); set_default_device(dev)
synthetic = len(sys.argv) > 1 and sys.argv[1] == "synthetic"
if synthetic:
    cg = torch.Generator().manual_seed(5)
    prob = Classification((torch.rand(1797, 64, generator=cg) * 16).to(dev), torch.randint(0, 10, (1797,), generator=cg).float().to(dev))
else:
    prob = Classification(dataset="digits")
cdesc = GenerateDescriptor(max_tree_len=128, input_len=64, output_len=10, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
pop = 200_000
algo = GeneticProgramming(Forest.random_generate(pop, cdesc, keys=torch.tensor([7, 0], dtype=torch.uint32, device=dev)),
                          DefaultCrossover(), DefaultMutation(0.2, cdesc.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.01))
for gen in range(7):
    for _ in range(2): prob.evaluate(algo.forest)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fit = prob.evaluate(algo.forest)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    ln = algo.forest.batch_subtree_size[:, 0].float()
    print(f"generation {gen}: fitness pass {ms:.3f} ms, mean tree length {float(ln.mean()):.1f} (max {int(ln.max())}), best accuracy {float(fit.max()):.4f}", flush=True)
    algo.step(fit)
