"""A/B of two engine builds on the division rows (gen_tc_asm.py DIVRANGE): fitness WORDS of a set of forests, saved so that the
same script run against the other build (scripts/gpu_div_ab.sh swaps the library) can be compared bit for bit, and the time
of the headline call.

    python scripts/dbg/div_range_ab.py run <tag>         -> gpurun_out/divab_<tag>.npz, timings on stdout
    python scripts/dbg/div_range_ab.py cmp <tagA> <tagB>
"""
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
    forest, Xd, yd, X, y = bench.sr_inputs(0, 1_000_000, dev)
    yield "headline_1M", forest, Xd, yd
    # operands of every magnitude: columns scaled by 10^-38 .. 10^38 (denormals, zeros and infinities among the intermediate
    # results), constants to match -- the blocks that must NOT take the unscaled rows
    rng = np.random.default_rng(7)
    scale = np.array([1e-38, 1e-25, 1e-14, 1e-13, 1.0, 3.0, 1e13, 1e14, 1e25, 1e38], np.float64)
    Xw = (rng.uniform(-5, 5, (1024, 10)) * scale[None, :]).astype(np.float32)
    Xw[rng.random((1024, 10)) < 0.02] = 0.0
    yw = rng.uniform(-5, 5, (1024, 1)).astype(np.float32)
    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6,
                              const_samples=[1e-30, 1e30, 3.0, -0.37, 0.0, 1.4e-14, 7.1e13, 2.0 ** -46, 2.0 ** 46, 2.0 ** -47, 2.0 ** 47])
    keys = torch.tensor([11, 5], dtype=torch.uint32, device=dev)
    wide = Forest.random_generate(200_000, desc, keys=keys)
    yield "wide_200k", wide, torch.from_numpy(Xw).to(dev), torch.from_numpy(yw).to(dev)
    # the boundary itself: every variable a power of two around 2^+-46 with a random sign
    e = rng.integers(-49, 50, (1024, 10))
    e = np.where(rng.random((1024, 10)) < 0.5, e, np.sign(e) * 46 + rng.integers(-2, 3, (1024, 10)))
    Xb = (np.ldexp(1.0 + (rng.random((1024, 10)) < 0.3) * rng.random((1024, 10)), e) * rng.choice([-1.0, 1.0], (1024, 10))).astype(np.float32)
    yield "boundary_200k", wide, torch.from_numpy(Xb).to(dev), torch.from_numpy(yw).to(dev)
    # some variables trusted (their column in range), some not: one zero in column 3, column 7 beyond 2^46
    Xm = X.copy()
    Xm[5, 3] = 0.0
    Xm[:, 7] *= 1e20
    yield "mixed_trust_200k", forest[:200_000], torch.from_numpy(Xm).to(dev), yd
    # three tiles (the distance between two variables in LDS is no power of two: nobody is trusted) and four
    Xm2 = np.concatenate([Xm, Xm[::-1]]).astype(np.float32)
    ym2 = np.concatenate([y, y[::-1]]).astype(np.float32)
    yield "mixed_trust_1536rows", forest[:100_000], torch.from_numpy(Xm2[:1536].copy()).to(dev), torch.from_numpy(ym2[:1536].copy()).to(dev)
    yield "mixed_trust_2048rows", forest[:100_000], torch.from_numpy(Xm2).to(dev), torch.from_numpy(ym2).to(dev)
    # more variables than trust bits (14)
    d20 = GenerateDescriptor(max_tree_len=64, input_len=20, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1, 2.5])
    f20 = Forest.random_generate(100_000, d20, keys=torch.tensor([3, 9], dtype=torch.uint32, device=dev))
    X20 = rng.uniform(-5, 5, (1024, 20)).astype(np.float32)
    X20[:, 16] *= 1e-30
    X20[7, 2] = 0.0
    yield "vars20_100k", f20, torch.from_numpy(X20).to(dev), yd
    # multi-output programs (accumulators, END_MO), K = 8 and K = 4
    for outs in (4, 10):
        dm = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=outs, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1, 0.5])
        fm = Forest.random_generate(50_000, dm, keys=torch.tensor([17, outs], dtype=torch.uint32, device=dev))
        ym = torch.from_numpy(rng.uniform(-3, 3, (1024, outs)).astype(np.float32)).to(dev)
        yield f"outputs{outs}_50k", fm, Xd, ym
    # K = 4 and K = 1 interpreters (short datasets)
    yield "wide_200rows", wide[:50_000], torch.from_numpy(Xw[:200].copy()).to(dev), torch.from_numpy(yw[:200].copy()).to(dev)
    yield "wide_50rows", wide[:50_000], torch.from_numpy(Xw[:50].copy()).to(dev), torch.from_numpy(yw[:50].copy()).to(dev)
    yield "headline_200rows", forest[:100_000], Xd[:200].contiguous(), yd[:200].contiguous()


def run(tag):
    import torch

    res = {}
    for name, forest, Xd, yd in forests():
        fit = forest.SR_fitness(Xd, yd)
        torch.cuda.synchronize()
        res[name] = fit.cpu().numpy().view(np.uint32)
        if name == "headline_1M":
            ts = []
            for _ in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    forest.SR_fitness(Xd, yd)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 5)
            print(f"{tag}: headline call {np.median(ts):.4f} ms (min {min(ts):.4f})")
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                forest.SR_fitness(Xd, yd)
            a.record()
            for _ in range(20):
                forest.SR_fitness(Xd, yd)
            b.record()
            torch.cuda.synchronize()
            print(f"{tag}: {name} call {a.elapsed_time(b) / 20:.4f} ms")
        f = res[name].view(np.float32)
        print(f"{tag}: {name}: {len(f)} trees, NaN {np.isnan(f).sum()}, inf {np.isinf(f).sum()}, finite median {np.nanmedian(np.where(np.isfinite(f), f, np.nan)):.6g}")
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, f"divab_{tag}.npz"), **res)


def cmp(a, b):
    A, B = np.load(os.path.join(OUT, f"divab_{a}.npz")), np.load(os.path.join(OUT, f"divab_{b}.npz"))
    bad = 0
    for k in A.files:
        x, y = A[k], B[k]
        nan = np.isnan(x.view(np.float32)) & np.isnan(y.view(np.float32))
        d = (x != y) & ~nan
        print(f"{k}: {len(x)} words, {int(d.sum())} differ" + (f" (first at {np.flatnonzero(d)[:5]}: {x[d][:3]} vs {y[d][:3]})" if d.any() else ""),
              f"; NaN in both {int(nan.sum())}, NaN words with other payloads {int(((x != y) & nan).sum())}")
        bad += int(d.sum())
    print("IDENTICAL" if bad == 0 else f"DIFFERENT: {bad} words")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
