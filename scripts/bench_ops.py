#!/usr/bin/env python3
"""Per-operator timings on one MI355X against the algorithmic bytes of SURVEY.md §8d (the rows of §8a other than the
headline fitness kernel): tree_generate, tree_crossover, tree_mutate, the fused breeding pass, tree_evaluate (C5 shape),
batch_evaluate (C4 shape, reduced population).  HIP events via torch on the launch stream; prints a markdown table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_capi as g


def depth2leaf(max_layer_cnt, leaf_prob=0.2):
    """f32[10]: leaf probability per depth (descriptor.py:33-38)"""
    return np.array([leaf_prob] * (max_layer_cnt - 1) + [1.0] * (10 - (max_layer_cnt - 1)), np.float32)


def roulette_uniform(funcs):
    """f32[29]: cumulative weights of the functions in use, equal shares (descriptor.py:106-111)"""
    w = np.zeros(29, np.float64)
    w[list(funcs)] = 1.0 / len(funcs)
    return np.cumsum(w.astype(np.float32), dtype=np.float32)

L_ = g.L
S = g._stream


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


rows = []
def row(name, us, bytes_, what):
    rows.append(f"| {name} | {what} | {us:.1f} | {bytes_ / 1e6:.1f} | {bytes_ / us / 1e6:.2f} | {bytes_ / us / 1e6 / 8.0 * 100:.1f} % |")

rng = np.random.default_rng(0)
pop, L = 100_000, 64
keys = g.dev([42, 0], np.uint32); d2l = g.dev(depth2leaf(6), np.float32); rou = g.dev(roulette_uniform([1, 2, 3, 4]), np.float32); cs = g.dev([-1, 0, 1], np.float32)
v = torch.empty((pop, L), dtype=torch.float32, device=g.DEV); t = torch.empty((pop, L), dtype=torch.int16, device=g.DEV); s = torch.empty((pop, L), dtype=torch.int16, device=g.DEV)
def gen():
    assert L_.evogp_hip_generate(pop, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d2l.data_ptr(), rou.data_ptr(), cs.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), 0, S()) == 0
us = timed(gen)
sizes = s[:, 0].to(torch.int64)
row("tree_generate", us, 8.0 * pop * L, f"pop {pop}, L {L}, mean len {float(sizes.float().mean()):.1f} (full rows written)")

# crossover: 30k survivors -> 99k offspring
n_s, n_new = 30_000, 99_000
li = g.dev(rng.integers(0, n_s, n_new), np.int32); ri = g.dev(rng.integers(0, n_s, n_new), np.int32)
sz = sizes[:n_s].cpu().numpy()
ln = g.dev(rng.integers(0, 2**31 - 1, n_new) % sz[li.cpu().numpy()], np.int32); rn = g.dev(rng.integers(0, 2**31 - 1, n_new) % sz[ri.cpu().numpy()], np.int32)
ov = torch.empty((n_new, L), dtype=torch.float32, device=g.DEV); ot = torch.empty((n_new, L), dtype=torch.int16, device=g.DEV); os_ = torch.empty((n_new, L), dtype=torch.int16, device=g.DEV)
def cross():
    assert L_.evogp_hip_crossover(n_s, n_new, L, v.data_ptr(), t.data_ptr(), s.data_ptr(), li.data_ptr(), ri.data_ptr(), ln.data_ptr(), rn.data_ptr(), ov.data_ptr(), ot.data_ptr(), os_.data_ptr(), S()) == 0
us = timed(cross)
len_left = sizes[li.long()].sum().item(); sub = s[ri.long(), rn.long()].to(torch.int64).sum().item()
row("tree_crossover", us, 8.0 * len_left + 8.0 * sub + 18.0 * n_new + 8.0 * n_new * L, f"{n_s} survivors -> {n_new} offspring (full rows written)")

# mutate: 19.8k trees
n_m = 19_800
mi = g.dev(rng.integers(0, 1024, n_m) % os_[:n_m, 0].cpu().numpy().clip(1), np.int32)
dv = torch.empty((n_m, L), dtype=torch.float32, device=g.DEV); dt = torch.empty((n_m, L), dtype=torch.int16, device=g.DEV); ds = torch.empty((n_m, L), dtype=torch.int16, device=g.DEV)
d2l3 = g.dev(depth2leaf(3), np.float32)
assert L_.evogp_hip_generate(n_m, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d2l3.data_ptr(), rou.data_ptr(), cs.data_ptr(), dv.data_ptr(), dt.data_ptr(), ds.data_ptr(), 0, S()) == 0
mv = torch.empty((n_m, L), dtype=torch.float32, device=g.DEV); mt = torch.empty((n_m, L), dtype=torch.int16, device=g.DEV); ms = torch.empty((n_m, L), dtype=torch.int16, device=g.DEV)
def mut():
    assert L_.evogp_hip_mutate(n_m, L, ov.data_ptr(), ot.data_ptr(), os_.data_ptr(), mi.data_ptr(), dv.data_ptr(), dt.data_ptr(), ds.data_ptr(), mv.data_ptr(), mt.data_ptr(), ms.data_ptr(), S()) == 0
us = timed(mut)
row("tree_mutate", us, 8.0 * os_[:n_m, 0].to(torch.int64).sum().item() + 8.0 * ds[:, 0].to(torch.int64).sum().item() + 4.0 * n_m + 8.0 * n_m * L, f"{n_m} trees (full rows written)")

# fused breeding pass
n_el = 1000
order = torch.argsort(torch.rand(pop, device=g.DEV), descending=True)[:n_s].to(torch.int32).contiguous()
rnd = torch.randint(0, 2**31 - 1, (6, pop - n_el), dtype=torch.int32, device=g.DEV)
below = int(0.2 * (2**31 - 1))
Dv = torch.empty((pop - n_el, L), dtype=torch.float32, device=g.DEV); Dt = torch.empty((pop - n_el, L), dtype=torch.int16, device=g.DEV); Ds = torch.empty((pop - n_el, L), dtype=torch.int16, device=g.DEV)
def genm():
    assert L_.evogp_hip_generate_masked(pop - n_el, L, 10, 1, 3, 0.5, 0.5, keys.data_ptr(), d2l3.data_ptr(), rou.data_ptr(), cs.data_ptr(), Dv.data_ptr(), Dt.data_ptr(), Ds.data_ptr(), 0, rnd[4].data_ptr(), below, S()) == 0
us = timed(genm)
n_act = int((rnd[4] < below).sum())
row("generate_masked", us, 8.0 * n_act * L, f"{n_act} donors of {pop - n_el} slots")
NV = torch.empty((pop, L), dtype=torch.float32, device=g.DEV); NT = torch.empty((pop, L), dtype=torch.int16, device=g.DEV); NS = torch.empty((pop, L), dtype=torch.int16, device=g.DEV)
def breed():
    assert L_.evogp_hip_breed_default(pop, L, n_el, n_s, v.data_ptr(), t.data_ptr(), s.data_ptr(), order.data_ptr(), rnd.data_ptr(), below, Dv.data_ptr(), Dt.data_ptr(), Ds.data_ptr(), NV.data_ptr(), NT.data_ptr(), NS.data_ptr(), None, S()) == 0
us = timed(breed)
row("breed_default", us, 8.0 * pop * L + 2 * 8.0 * float(sizes.float().mean()) * pop * 0.75 + 8.0 * n_act * 6, "elites + crossover + mutation -> next generation (approximate algorithmic bytes)")

# tree_evaluate at the policy shape (C5): pop 50k, L 256, in 17, out 6
pe, Le = 50_000, 256
d2l6 = g.dev(depth2leaf(6), np.float32); cse = g.dev(np.linspace(-1, 1, 100), np.float32)
ev = torch.empty((pe, Le), dtype=torch.float32, device=g.DEV); et = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV); es = torch.empty((pe, Le), dtype=torch.int16, device=g.DEV)
assert L_.evogp_hip_generate(pe, Le, 17, 6, 100, 0.5, 0.5, keys.data_ptr(), d2l6.data_ptr(), rou.data_ptr(), cse.data_ptr(), ev.data_ptr(), et.data_ptr(), es.data_ptr(), 0, S()) == 0
obs = torch.randn(pe, 17, device=g.DEV); res = torch.empty(pe, 6, device=g.DEV)
def evaluate():
    assert L_.evogp_hip_evaluate(pe, Le, 17, 6, ev.data_ptr(), et.data_ptr(), es.data_ptr(), obs.data_ptr(), res.data_ptr(), S()) == 0
us_eager = timed(evaluate, 50)
# the same call recorded into a HIP graph, 20 calls per replay: the device's time per call (the eager figure is bound by this script's
# own ctypes call: ~20 us of Python per call, more than the two kernels take)
gs = torch.cuda.Stream()
with torch.cuda.stream(gs):
    evaluate(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=gs):
        for _ in range(20): evaluate()
us = timed(gr.replay, 20) / 20
row("tree_evaluate (C5 shape), eager calls from Python", us_eager, 6.0 * es[:, 0].to(torch.int64).sum().item() + 2.0 * pe + 4.0 * pe * 23, f"pop {pe}, L {Le}, in 17, out 6, one policy step; bound by the caller")
row("tree_evaluate (C5 shape)", us, 6.0 * es[:, 0].to(torch.int64).sum().item() + 2.0 * pe + 4.0 * pe * 23, f"pop {pe}, L {Le}, in 17, out 6, one policy step")

# batch_evaluate at the classifier shape (C4, reduced pop): pop 20k, L 128, in 64, out 10, D 1797
pc, Lc, Dc = 20_000, 128, 1797
cv = torch.empty((pc, Lc), dtype=torch.float32, device=g.DEV); ct = torch.empty((pc, Lc), dtype=torch.int16, device=g.DEV); cs_ = torch.empty((pc, Lc), dtype=torch.int16, device=g.DEV)
assert L_.evogp_hip_generate(pc, Lc, 64, 10, 3, 0.5, 0.5, keys.data_ptr(), d2l6.data_ptr(), rou.data_ptr(), cs.data_ptr(), cv.data_ptr(), ct.data_ptr(), cs_.data_ptr(), 0, S()) == 0
Xc = torch.rand(Dc, 64, device=g.DEV) * 16; out = torch.empty((pc, Dc, 10), dtype=torch.float32, device=g.DEV)
def bev():
    assert L_.evogp_hip_batch_evaluate(pc, Dc, Lc, 64, 10, cv.data_ptr(), ct.data_ptr(), cs_.data_ptr(), Xc.data_ptr(), out.data_ptr(), S()) == 0
us = timed(bev, 5)
row("batch_evaluate (C4 shape, pop 20k)", us, 4.0 * pc * Dc * 10 + 6.0 * cs_[:, 0].to(torch.int64).sum().item(), f"pop {pc}, L {Lc}, in 64, out 10, D {Dc}: {pc * Dc / us / 1e3:.1f} G tree-evals/s, results written once")

labels = torch.randint(0, 10, (Dc,), dtype=torch.int32, device=g.DEV)
counts = torch.empty(pc, dtype=torch.int32, device=g.DEV)
def acc():
    assert L_.evogp_hip_batch_argmax_count(pc, Dc, Lc, 64, 10, cv.data_ptr(), ct.data_ptr(), cs_.data_ptr(), Xc.data_ptr(), labels.data_ptr(), counts.data_ptr(), S()) == 0
us = timed(acc, 5)
row("batch_argmax_count (C4 shape, pop 20k)", us, 6.0 * cs_[:, 0].to(torch.int64).sum().item() + 4.0 * pc + 4.0 * Dc * 65, f"fused classification epilogue: {pc * Dc / us / 1e3:.1f} G tree-evals/s, only the per-tree counts leave the chip")

# BASELINE configs[3] in full: pop 200k (example/uci_classifier.py), the fitness pass of the Classification problem
pf_ = 200_000
fv = torch.empty((pf_, Lc), dtype=torch.float32, device=g.DEV); ft = torch.empty((pf_, Lc), dtype=torch.int16, device=g.DEV); fs = torch.empty((pf_, Lc), dtype=torch.int16, device=g.DEV)
assert L_.evogp_hip_generate(pf_, Lc, 64, 10, 3, 0.5, 0.5, keys.data_ptr(), d2l6.data_ptr(), rou.data_ptr(), cs.data_ptr(), fv.data_ptr(), ft.data_ptr(), fs.data_ptr(), 0, S()) == 0
fcounts = torch.empty(pf_, dtype=torch.int32, device=g.DEV)
def accf():
    assert L_.evogp_hip_batch_argmax_count(pf_, Dc, Lc, 64, 10, fv.data_ptr(), ft.data_ptr(), fs.data_ptr(), Xc.data_ptr(), labels.data_ptr(), fcounts.data_ptr(), S()) == 0
us = timed(accf, 5)
row("batch_argmax_count (configs[3]: pop 200k)", us, 6.0 * fs[:, 0].to(torch.int64).sum().item() + 4.0 * pf_ + 4.0 * Dc * 65, f"pop {pf_}, L {Lc}, in 64, out 10, D {Dc}: {pf_ * Dc / us / 1e3:.1f} G tree-evals/s")
del fv, ft, fs

# single-output batch evaluation over a long dataset (Transformation / torch-mode SR): pop 50k, L 64, in 10, D 4096
ps, Ds = 50_000, 4096
Xs = torch.rand(Ds, 10, device=g.DEV) * 10 - 5; outs_ = torch.empty((ps, Ds, 1), dtype=torch.float32, device=g.DEV)
def bso():
    assert L_.evogp_hip_batch_evaluate(ps, Ds, L, 10, 1, v.data_ptr(), t.data_ptr(), s.data_ptr(), Xs.data_ptr(), outs_.data_ptr(), S()) == 0
us = timed(bso, 5)
row("batch_evaluate (single output, long dataset)", us, 4.0 * ps * Ds + 6.0 * s[:ps, 0].to(torch.int64).sum().item(), f"pop {ps}, L {L}, in 10, out 1, D {Ds}: {ps * Ds / us / 1e3:.1f} G tree-evals/s")

print("| operator | workload | us per call | algorithmic MB | TB/s | of 8 TB/s |\n|---|---|---|---|---|---|")
print("\n".join(rows))
