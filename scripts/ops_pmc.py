#!/usr/bin/env python3
"""The HBM-bound operators under the TCC counters (north_star: "rocprof HBM GB/s against the gfx950 roofline").

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/ops_pmc.py run          (one pass per counter set)
    python scripts/ops_pmc.py table fetch.md write.md stats.md ops.json > profiles/r05_ops_pmc.md

`run` launches tree_generate, tree_crossover, tree_mutate, generate_masked, the breeding pass and tree_evaluate (C5 shape) REPS times each
on the workloads of scripts/bench_ops.py and writes their algorithmic bytes to OPS_JSON; `table` joins the
per-dispatch counter averages of scripts/rocpd_summary.py with them: counter bytes (FETCH_SIZE doubled: the calibration of
profiles/r04Z_06_calib.md) / algorithmic bytes, and GB/s of both kinds against 8 TB/s."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REPS = 6
OPS_JSON = os.environ.get("OPS_JSON", os.path.join(ROOT, "gpurun_out", "ops_pmc_ops.json"))


def run():
    import numpy as np, torch
    import gpu_capi as g
    L_, S = g.L, g._stream

    def d2l(m, leaf=0.2): return np.array([leaf] * (m - 1) + [1.0] * (10 - (m - 1)), np.float32)
    def rou(funcs):
        w = np.zeros(29, np.float64); w[list(funcs)] = 1.0 / len(funcs)
        return np.cumsum(w.astype(np.float32), dtype=np.float32)
    ops = {}
    rng = np.random.default_rng(0)
    pop, L = 100_000, 64
    keys = g.dev([42, 0], np.uint32); d6 = g.dev(d2l(6), np.float32); d3 = g.dev(d2l(3), np.float32); r4 = g.dev(rou([1, 2, 3, 4]), np.float32); cs = g.dev([-1, 0, 1], np.float32)
    E = lambda n, w, dt: torch.empty((n, w), dtype=dt, device=g.DEV)
    v, t, s = E(pop, L, torch.float32), E(pop, L, torch.int16), E(pop, L, torch.int16)
    for _ in range(REPS):
        assert L_.evogp_hip_generate(pop, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d6.data_ptr(), r4.data_ptr(), cs.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), 0, S()) == 0
    sizes = s[:, 0].to(torch.int64)
    ops["tree_generate (100 k x L 64)"] = {"kernel": "generate_staged_kernel<false>", "bytes": 8.0 * pop * L}
    n_s, n_new = 30_000, 99_000
    li = g.dev(rng.integers(0, n_s, n_new), np.int32); ri = g.dev(rng.integers(0, n_s, n_new), np.int32)
    sz = sizes[:n_s].cpu().numpy()
    ln = g.dev(rng.integers(0, 2**31 - 1, n_new) % sz[li.cpu().numpy()], np.int32); rn = g.dev(rng.integers(0, 2**31 - 1, n_new) % sz[ri.cpu().numpy()], np.int32)
    ov, ot, os_ = E(n_new, L, torch.float32), E(n_new, L, torch.int16), E(n_new, L, torch.int16)
    for _ in range(REPS):
        assert L_.evogp_hip_crossover(n_s, n_new, L, v.data_ptr(), t.data_ptr(), s.data_ptr(), li.data_ptr(), ri.data_ptr(), ln.data_ptr(), rn.data_ptr(), ov.data_ptr(), ot.data_ptr(), os_.data_ptr(), S()) == 0
    len_left = sizes[li.long()].sum().item(); sub = s[ri.long(), rn.long()].to(torch.int64).sum().item()
    ops["tree_crossover (30 k -> 99 k)"] = {"kernel": "crossover_group_kernel", "bytes": 8.0 * len_left + 8.0 * sub + 18.0 * n_new + 8.0 * n_new * L}
    n_m = 19_800
    mi = g.dev(rng.integers(0, 1024, n_m) % os_[:n_m, 0].cpu().numpy().clip(1), np.int32)
    dv, dt, ds = E(n_m, L, torch.float32), E(n_m, L, torch.int16), E(n_m, L, torch.int16)
    assert L_.evogp_hip_generate(n_m, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d3.data_ptr(), r4.data_ptr(), cs.data_ptr(), dv.data_ptr(), dt.data_ptr(), ds.data_ptr(), 0, S()) == 0
    mv, mt, ms = E(n_m, L, torch.float32), E(n_m, L, torch.int16), E(n_m, L, torch.int16)
    for _ in range(REPS):
        assert L_.evogp_hip_mutate(n_m, L, ov.data_ptr(), ot.data_ptr(), os_.data_ptr(), mi.data_ptr(), dv.data_ptr(), dt.data_ptr(), ds.data_ptr(), mv.data_ptr(), mt.data_ptr(), ms.data_ptr(), S()) == 0
    ops["tree_mutate (19.8 k)"] = {"kernel": "mutate_group_kernel", "bytes": 8.0 * os_[:n_m, 0].to(torch.int64).sum().item() + 8.0 * ds[:, 0].to(torch.int64).sum().item() + 4.0 * n_m + 8.0 * n_m * L}
    n_el = 1000
    order = torch.argsort(torch.rand(pop, device=g.DEV), descending=True)[:n_s].to(torch.int32).contiguous()
    rnd = torch.randint(0, 2**31 - 1, (6, pop - n_el), dtype=torch.int32, device=g.DEV)
    below = int(0.2 * (2**31 - 1))
    Dv, Dt, Ds = E(pop - n_el, L, torch.float32), E(pop - n_el, L, torch.int16), E(pop - n_el, L, torch.int16)
    for _ in range(REPS + 1):   # (one launch more than the others: the same kernel as tree_generate, told apart by the count)
        assert L_.evogp_hip_generate_masked(pop - n_el, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d3.data_ptr(), r4.data_ptr(), cs.data_ptr(), Dv.data_ptr(), Dt.data_ptr(), Ds.data_ptr(), 0, rnd[4].data_ptr(), below, S()) == 0
    n_act = int((rnd[4] < below).sum())
    ops["generate_masked (19.8 k donors of 99 k slots)"] = {"kernel": "generate_staged_kernel<false>", "bytes": 8.0 * n_act * L, "reps": REPS + 1}
    NV, NT, NS = E(pop, L, torch.float32), E(pop, L, torch.int16), E(pop, L, torch.int16)
    for _ in range(REPS):
        assert L_.evogp_hip_breed_default(pop, L, n_el, n_s, v.data_ptr(), t.data_ptr(), s.data_ptr(), order.data_ptr(), rnd.data_ptr(), below, Dv.data_ptr(), Dt.data_ptr(), Ds.data_ptr(), NV.data_ptr(), NT.data_ptr(), NS.data_ptr(), None, S()) == 0
    ops["breeding pass (100 k)"] = {"kernel": "breed_group_kernel", "bytes": 8.0 * pop * L + 2 * 8.0 * float(sizes.float().mean()) * pop * 0.75 + 8.0 * n_act * 6}
    pe, Le = 50_000, 256
    cse = g.dev(np.linspace(-1, 1, 100), np.float32)
    ev, et, es = E(pe, Le, torch.float32), E(pe, Le, torch.int16), E(pe, Le, torch.int16)
    assert L_.evogp_hip_generate(pe, Le, 17, 6, 100, 0.5, 0.5, keys.data_ptr(), d6.data_ptr(), r4.data_ptr(), cse.data_ptr(), ev.data_ptr(), et.data_ptr(), es.data_ptr(), 0, S()) == 0
    obs = torch.randn(pe, 17, device=g.DEV); res = torch.empty(pe, 6, device=g.DEV)
    for _ in range(REPS):
        assert L_.evogp_hip_evaluate(pe, Le, 17, 6, ev.data_ptr(), et.data_ptr(), es.data_ptr(), obs.data_ptr(), res.data_ptr(), S()) == 0
    ops["tree_evaluate (C5 shape, 50 k x L 256)"] = {"kernel": "eval_direct_kernel", "bytes": 6.0 * es[:, 0].to(torch.int64).sum().item() + 2.0 * pe + 4.0 * pe * 23}
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(OPS_JSON), exist_ok=True)
    json.dump(ops, open(OPS_JSON, "w"), indent=1)
    print("ops written:", OPS_JSON)


def table(fetch_md, write_md, stats_md, ops_json):
    def counters(path):
        out = {}
        for line in open(path):
            m = re.match(r"\| `(.+?)` \| (\w+) \| (\d+) \| ([\d.e+-]+) \| ([\d.e+-]+) \|", line)
            if m:
                out[m.group(1)] = (int(m.group(3)), float(m.group(5)))
        return out
    def durations(path):
        out = {}
        for line in open(path):
            m = re.match(r"\| `(.+?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", line)
            if m:
                out[m.group(1)] = (int(m.group(2)), float(m.group(4)), float(m.group(5)))   # calls, avg us, min us
        return out
    f, w, d = counters(fetch_md), counters(write_md), durations(stats_md)
    ops = json.load(open(ops_json))
    def pick(table_, key, reps=REPS):
        # rows are per kernel AND grid (ROCPD_BY_GRID=1): the measured launches are the ones with REPS dispatches (FETCH_SIZE / WRITE_SIZE
        # come as one row per launch; SQ counters would come as 32)
        key = key.replace(" ", "")
        hits = [(k, v) for k, v in table_.items() if key in k.replace(" ", "") and v[0] == reps]
        return hits[0] if hits else (None, None)
    print("| operator | kernel | avg µs (rocprofv3, counters on) | algorithmic MB | FETCH_SIZE raw MB | × 2 (calibrated) | WRITE_SIZE MB | counter bytes ÷ algorithmic | algorithmic GB/s (of 8 TB/s) | counter GB/s (of 8 TB/s) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, o in ops.items():
        r = o.get("reps", REPS)
        kf, vf = pick(f, o["kernel"], r); kw, vw = pick(w, o["kernel"], r); kd, vd = pick(d, o["kernel"], r)
        if not vf or not vw or not vd:
            print(f"| {name} | `{o['kernel']}` | (no rows: fetch {bool(vf)}, write {bool(vw)}, stats {bool(vd)}) |")
            continue
        fb, wb, us, alg = vf[1] * 1024, vw[1] * 1024, vd[1], o["bytes"]
        hbm = 2 * fb + wb
        print(f"| {name} | `{kd[:60]}` | {us:.1f} | {alg / 1e6:.1f} | {fb / 1e6:.1f} | {2 * fb / 1e6:.1f} | {wb / 1e6:.1f} | {hbm / alg:.2f} | "
              f"{alg / us / 1e3:.0f} ({alg / us / 1e3 / 80:.1f} %) | {hbm / us / 1e3:.0f} ({hbm / us / 1e3 / 80:.1f} %) |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        table(*sys.argv[2:6])
