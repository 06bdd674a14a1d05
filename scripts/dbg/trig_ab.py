"""A/B of two engine builds on forests of sin / cos / tan (and the other library functions): fitness words saved per build and
compared bit for bit, call times.   python scripts/dbg/trig_ab.py run <tag> | cmp <tagA> <tagB>   (scripts/gpu_div_ab.sh with AB_SCRIPT)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def forests():
    import torch

    import bench
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)   # (const_range / sample_cnt draws its constants from torch's generator: the same in every build's run)
    _, Xd, yd, X, y = bench.sr_inputs(0, 1000, dev)
    keys = lambda a, b: torch.tensor([a, b], dtype=torch.uint32, device=dev)  # noqa: E731
    p7 = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"], max_layer_cnt=6,
                            const_samples=[-1, 0, 1])
    f7 = Forest.random_generate(100_000, p7, keys=keys(42, 0))
    yield "paper7_100k", f7, Xd, yd
    yield "paper7_100k_large_arguments", f7, (Xd * 3.0e4).contiguous(), yd        # blocks with |x| >= 2^17: the library's whole function
    yield "paper7_100k_mixed_arguments", f7, (Xd * torch.tensor([1, 1, 1e5, 1, 1, 1e7, 1, 1, 1, 1e30], device=dev)).contiguous(), yd
    yield "paper7_300rows", f7[:50_000], Xd[:300].contiguous(), yd[:300].contiguous()   # K = 4
    yield "paper7_40rows", f7[:50_000], Xd[:40].contiguous(), yd[:40].contiguous()      # K = 1
    tr = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["sin", "cos", "tan", "+", "*"], max_layer_cnt=6,
                            const_samples=[-1, 0.5, 2, 100.0, 1e6])
    yield "trig_heavy_100k", Forest.random_generate(100_000, tr, keys=keys(7, 7)), Xd, yd
    uci = GenerateDescriptor(max_tree_len=512, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"], max_layer_cnt=9,
                             const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
    yield "uci_shape_100k", Forest.random_generate(100_000, uci, keys=keys(42, 0)), Xd, yd
    el = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "exp", "log", "pow"], max_layer_cnt=6,
                            const_samples=[-1, 0, 1])
    yield "exp_log_pow_100k", Forest.random_generate(100_000, el, keys=keys(42, 0)), Xd, yd
    # pow / loose pow / sinh / cosh / tanh: the library's sequences over row pairs (round 5, gen/pair_rows.py) -- 8, 4 and 1 rows per lane
    hy = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "*", "pow", "loose_pow", "sinh", "cosh", "tanh", "/"], max_layer_cnt=6,
                            const_samples=[-1, 0, 1, 0.5, 2, 3.5])
    fh = Forest.random_generate(100_000, hy, keys=keys(9, 9))
    yield "pow_hyperbolic_100k", fh, Xd, yd
    yield "pow_hyperbolic_wide_arguments", fh, (Xd * torch.tensor([1, 20, 1e-3, 1, 1e4, 1, -1, 1, 1e-20, 1e30], device=dev)).contiguous(), yd
    yield "pow_hyperbolic_300rows", fh[:50_000], Xd[:300].contiguous(), yd[:300].contiguous()
    yield "pow_hyperbolic_40rows", fh[:50_000], Xd[:40].contiguous(), yd[:40].contiguous()


def run(tag):
    import torch

    res = {}
    for name, forest, Xd, yd in forests():
        fit = forest.SR_fitness(Xd, yd)
        torch.cuda.synchronize()
        res[name] = fit.cpu().numpy().view(np.uint32)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            forest.SR_fitness(Xd, yd)
        a.record()
        for _ in range(20):
            forest.SR_fitness(Xd, yd)
        b.record()
        torch.cuda.synchronize()
        f = res[name].view(np.float32)
        print(f"{tag}: {name} call {a.elapsed_time(b) / 20:.4f} ms; NaN {np.isnan(f).sum()}, inf {np.isinf(f).sum()}")
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"divab_{tag}.npz"), **res)


def cmp(a, b):
    A, B = np.load(os.path.join(OUT, f"divab_{a}.npz")), np.load(os.path.join(OUT, f"divab_{b}.npz"))
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        nan = np.isnan(x.view(np.float32)) & np.isnan(y.view(np.float32))
        d = (x != y) & ~nan
        print(f"{k}: {len(x)} words, {int(d.sum())} differ" + (f" (first at {np.flatnonzero(d)[:5]}: {x[d][:3]} vs {y[d][:3]})" if d.any() else ""))
        bad += int(d.sum())
    print("IDENTICAL" if bad == 0 else f"DIFFERENT: {bad} words")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
