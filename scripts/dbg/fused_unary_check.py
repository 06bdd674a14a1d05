"""The one-kernel fitness call (EVOGP_TC_FUSED=1) against the two-kernel path (=0) on forests WITH unary functions (its packed unary
compiler inlines the library's folding functions): fitness words bit for bit.  Two processes: the switch is read once."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
SETS = {"trig": ["+", "-", "*", "/", "sin", "cos", "tan"], "mixed": ["+", "-", "*", "/", "neg", "abs", "sqrt", "inv", "exp", "log"], "arith": ["+", "-", "*", "/"]}


def child(tag):
    import torch

    import bench
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    _, Xd, yd, _, _ = bench.sr_inputs(0, 1000, dev)
    for name, funcs in SETS.items():
        desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=funcs, max_layer_cnt=6, const_samples=[-1, 0, 1, 0.5, 2])
        f = Forest.random_generate(200_000, desc, keys=torch.tensor([5, 6], dtype=torch.uint32, device=dev))
        w = f.SR_fitness(Xd, yd)
        torch.cuda.synchronize()
        np.save(os.path.join(OUT, f"fu_{tag}_{name}.npy"), w.cpu().numpy().view(np.uint32))


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    os.makedirs(OUT, exist_ok=True)
    for tag, v in (("two", "0"), ("one", "1")):
        subprocess.run([sys.executable, __file__, tag], env={**os.environ, "EVOGP_TC_FUSED": v}, check=True)
    for name in SETS:
        a, b = (np.load(os.path.join(OUT, f"fu_{t}_{name}.npy")) for t in ("two", "one"))
        d = np.nonzero(a != b)[0]
        print(f"{name}: {len(a)} fitness words, {len(d)} differ between the one-kernel and the two-kernel call", d[:5])
    for f in os.listdir(OUT):
        if f.startswith("fu_"):
            os.remove(os.path.join(OUT, f))


if __name__ == "__main__":
    main()
