"""GPU: the widened threaded-code fitness path against the CPU oracle.

Round 1's threaded code took single-output trees of at most 64 nodes over + - * / and ten unary functions; everything else
fell to the 3-6x slower register kernels.  These tests cover what `compile_general` (csrc/sr_tc.hip) and the new handlers add
— trees of up to gp_len nodes (chained program blocks), loose division / inverse, max min, the four comparisons, IF, unknown
unary ids, MULTI-OUTPUT trees (forward.cu:237-243) — always `evogp_hip_sr_fitness` through the C ABI against
`oracle.sr_fitness` on the same inputs, 1e-5 relative on finite values and identical NaN / inf classes.

Each test also reads the handler histogram of the compiled population (`evogp_hip_debug_tc_histogram`): the share of trees
whose program is SKIP (left to the register kernels) must be small, so that a green result is the threaded code's result and
not the fallback's."""
import ctypes
import json
import os

import numpy as np
import pytest

from helpers import (ARITH, assert_close_classes, assert_within_sensitivity, c2_dataset, depth2leaf, per_tree_tolerance, roulette_uniform,
                     torch_rule_counts)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = [-1.0, 0.0, 1.0, 0.5, 2.0]
RTOL = 1e-5
# function ids (defs.h:24-57)
IF, ADD, SUB, MUL, DIV, LDIV, POW, LPOW, MAX, MIN, LT, GT, LE, GE = range(14)
SIN, COS, TAN, SINH, COSH, TANH, LOG, LLOG, EXP, INV, LINV, NEG, ABS, SQRT, LSQRT = range(14, 29)
EXACT_WIDE = [IF, ADD, SUB, MUL, DIV, LDIV, MAX, MIN, LT, GT, LE, GE, INV, LINV, NEG, ABS, SQRT, LSQRT]  # IEEE-exact on both sides


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import gpu_capi

    return gpu_capi


def handler_histogram(g, pop, fold_twins=True):
    """{handler name: words} of the programs the last sr_fitness call compiled (both flavours added)"""
    import torch

    nh = g.L.evogp_hip_debug_tc_nhandlers()
    hist = torch.zeros(2 * nh, dtype=torch.int64, device=g.DEV)
    rc = g.L.evogp_hip_debug_tc_histogram(pop, ctypes.c_void_p(hist.data_ptr()), 2 * nh, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, g.L.evogp_hip_error_string(rc)
    h = hist.cpu().numpy()
    table = json.load(open(os.path.join(ROOT, "evogp_amd", "lib", "tc_handlers.json")))["K8_short"]["handlers"]
    out = {}
    for name, v in table.items():   # (a handler and its twin that does not prefetch, name + "_np", count as one)
        base = name[:-3] if fold_twins and name.endswith("_np") else name
        out[base] = out.get(base, 0) + int(h[v["id"]] + h[nh + v["id"]])
    return out


def check(g, oracle, forest, X, y, what, max_skipped=0.02, mse=(True, False)):
    pop = forest[0].shape[0]
    for m in mse:
        got = g.sr_fitness(*forest, X, y, m)
        if m is mse[0]:
            h = handler_histogram(g, pop)
            assert h["skip"] <= max_skipped * pop, f"{what}: {h['skip']} of {pop} trees were left to the register kernels"
        assert_close_classes(got, oracle.sr_fitness(*forest, X, y, m), RTOL, what=f"{what} mse={m}")
    return h


# ---- long trees: chained program blocks ----------------------------------------------------------------------------------
@pytest.mark.parametrize("L,mlc,funcs", [(128, 7, ARITH), (256, 8, ARITH), (256, 7, ARITH + [NEG, ABS, SQRT, INV]), (1024, 10, ARITH), (130, 7, ARITH)])
def test_long_trees(g, oracle, rng, L, mlc, funcs):
    assert 2 ** mlc - 1 <= L  # GenerateDescriptor's check_tree_length (descriptor.py:19-31): a full tree must fit the row
    d2l = depth2leaf(mlc, 0.1)
    forest = oracle.generate(3000, L, 6, 1, 0.5, 0.5, [L, mlc], d2l, roulette_uniform(funcs), CS)
    sizes = forest[2][:, 0]
    assert sizes.max() > 64, "the forest must contain trees beyond one 64-node chunk"
    X = rng.uniform(-3, 3, (700, 6)).astype(np.float32); y = rng.uniform(-3, 3, (700, 1)).astype(np.float32)
    # (trees whose operand stack would be deeper than the interpreter's register stack are compiled with the larger subtree first)
    h = check(g, oracle, forest, X, y, f"L={L}", max_skipped=0.01)
    assert h["next"] > 0, "no program needed a second block: the test does not exercise the chaining"


def test_evolved_long_trees_after_crossover(g, oracle, rng):
    """second-generation trees at the length cap (L = 128): crossover products, lengths up to exactly gp_len"""
    L = 128
    f0 = oracle.generate(4000, L, 5, 1, 0.5, 0.5, [9, 9], depth2leaf(7, 0.1), roulette_uniform(ARITH + [NEG]), CS)
    sizes = f0[2][:, 0].astype(np.int64)
    li = rng.integers(0, 4000, 4000).astype(np.int32); ri = rng.integers(0, 4000, 4000).astype(np.int32)
    ln = (rng.integers(0, 2**31 - 1, 4000) % sizes[li]).astype(np.int32); rn = (rng.integers(0, 2**31 - 1, 4000) % sizes[ri]).astype(np.int32)
    f1 = oracle.crossover(*f0, li, ri, ln, rn)
    X = rng.uniform(-3, 3, (1024, 5)).astype(np.float32); y = rng.uniform(-3, 3, (1024, 1)).astype(np.float32)
    check(g, oracle, f1, X, y, "crossover products, L=128", max_skipped=0.01)


# ---- the program compilers: one tree per pass, several trees per pass in batches of 8 ... 64 ------------------------------
@pytest.mark.parametrize("masked,L,unary", [(False, 64, [NEG, ABS, SQRT, INV]), (True, 64, []), (True, 33, []), (False, 20, [NEG, ABS, SQRT, INV]),
                                             (False, 64, [SIN, COS, TAN, EXP, LOG, NEG]), (False, 128, [NEG, ABS, SQRT, INV]), (True, 200, []),
                                             (False, 256, [SIN, COS, TAN, NEG])])
def test_every_program_compiler_gives_the_same_fitness_words(g, oracle, rng, masked, L, unary):
    """tc_compile_packed_kernel (several trees per pass, largest first) against tc_compile_kernel (one tree per pass), fitness WORDS:
    a population whose size is no multiple of any batch, rows without a tree, a wrong subtree size, the generator's "no function"
    node (unary, id 29: generate.cu:77-84) over trees of every kind, a function the mask does not announce.  `masked`: the forest
    carries its function mask, so the arithmetic-only instantiation runs (and compiles the id-29 nodes itself).  Function sets that are
    IEEE-exact on both sides are also held against the oracle."""
    import torch

    from evogp_amd import _lib
    from evogp_amd.tree import Forest

    pop, V = 20011, 6
    funcs = [ADD, SUB, MUL, DIV] + unary
    v, t, s = oracle.generate(pop, L, V, 1, 0.0, 0.5, [L, 5 + masked], depth2leaf(7 if L > 64 else 6 if L > 40 else 5 if L > 30 else 4), roulette_uniform(funcs), CS)
    v, t, s = v.copy(), t.copy(), s.copy()
    s[17, 0] = 0; s[18, 0] = -3; s[40000 % pop, 0] = 0          # rows without a tree: NaN
    r = int(np.nonzero(s[:, 0] >= 5)[0][0]); s[r, 1] += 1        # a subtree size that does not add up: the register kernels
    wrapped = 0
    if L > 64:
        assert (s[:, 0] > 64).sum() > 5, "no tree beyond 64 nodes: the long-row case is not covered"
    for r in np.nonzero(s[:, 0] < L - 2)[0][100:160]:            # "no function" over the whole tree (some twice), leaves included
        for _ in range(1 + (r & 1)):
            n = int(s[r, 0])
            v[r, 1:n + 1] = v[r, :n].copy(); t[r, 1:n + 1] = t[r, :n].copy(); s[r, 1:n + 1] = s[r, :n].copy()
            v[r, 0] = 29.0; t[r, 0] = 2; s[r, 0] = n + 1
            wrapped += 1
    r = int(np.nonzero((s[:, 0] >= 3) & (t[:, 0] == 3))[0][7]); v[r, 0] = float(MAX)   # not in the mask / a generic stub
    assert wrapped >= 60
    X = rng.uniform(-3, 3, (300, V)).astype(np.float32); y = rng.uniform(-3, 3, (300, 1)).astype(np.float32)
    mask = sum(1 << f for f in funcs) if masked else 0
    forest = Forest(V, 1, torch.from_numpy(v).to(g.DEV), torch.from_numpy(t).to(g.DEV), torch.from_numpy(s).to(g.DEV), func_mask=mask)
    Xd, yd = torch.from_numpy(X).to(g.DEV), torch.from_numpy(y).to(g.DEV)
    words = {}
    try:
        for batch in (0, 8, 16, 32, 64, -1):
            assert _lib.lib.evogp_hip_debug_compile_batch(batch) == 0
            words[batch] = forest.SR_fitness(Xd, yd).cpu().numpy().view(np.uint32).copy()
        # the handlers' twins that do not prefetch (a launch of this size does without them by default): always / never, packed and one tree per pass
        for twins in (1, 0):
            assert _lib.lib.evogp_hip_debug_twins(twins) == 0
            for batch in (0, 16, -1):
                assert _lib.lib.evogp_hip_debug_compile_batch(batch) == 0
                words[f"twins {twins}, batch {batch}"] = forest.SR_fitness(Xd, yd).cpu().numpy().view(np.uint32).copy()
            if twins:
                h = handler_histogram(g, pop, fold_twins=False)
                assert sum(n for name, n in h.items() if name.endswith("_np")) > pop // 4, "no word names a twin"
        assert _lib.lib.evogp_hip_debug_twins(-1) == 0 and _lib.lib.evogp_hip_debug_compile_batch(-1) == 0
        if L > 64:   # trees of more than 64 nodes: the general compiler's staged passes against the straight-line staged compiler (round 5)
            assert _lib.lib.evogp_hip_debug_long_compiler(0) == 0
            words["long trees through compile_general"] = forest.SR_fitness(Xd, yd).cpu().numpy().view(np.uint32).copy()
    finally:
        _lib.lib.evogp_hip_debug_compile_batch(-1)
        _lib.lib.evogp_hip_debug_long_compiler(-1)
        _lib.lib.evogp_hip_debug_twins(-1)
    for batch, w in words.items():
        diff = np.nonzero(w != words[0])[0]
        assert len(diff) == 0, f"batch {batch}: {len(diff)} fitness words differ from the one-tree compiler's, first tree {diff[:5]}"
    got = words[-1].view(np.float32)
    assert np.isnan(got[[17, 18, 40000 % pop]]).all()
    if SIN not in unary:   # (the library functions are judged against the oracle elsewhere, with their own envelope)
        keep = np.ones(pop, bool); keep[[17, 18, 40000 % pop]] = False   # (a row without a tree is NaN here; the reference reads what lies there)
        assert_close_classes(got[keep], oracle.sr_fitness(v, t, s, X, y, True)[keep], RTOL, what=f"packed compilers, masked={masked}")


@pytest.mark.parametrize("funcs,out_len,L,exact", [(EXACT_WIDE, 1, 64, True), (EXACT_WIDE, 1, 128, True), ([ADD, SUB, MUL, DIV, POW, LPOW, SINH, COSH, TANH, IF, MAX], 1, 64, False),
                                                   (EXACT_WIDE, 4, 64, True), (EXACT_WIDE, 10, 128, True), ([ADD, MUL, POW, LPOW, TANH, SINH, IF, EXP], 3, 64, False)],
                         ids=["exact", "exact-L128", "library", "mo4", "mo10-L128", "mo3-library"])
def test_generic_and_multi_output_lines_of_the_packed_compiler(g, oracle, rng, funcs, out_len, L, exact):
    """Round 5: the packed compiler's generic line (functions behind the generic stubs: csrc/sr_tc.hip compile_pack_generic) and its
    multi-output line (compile_pack_mo) against tc_compile_general_kernel, one tree per wave pass (evogp_hip_debug_compile_batch(0)):
    the same fitness WORDS -- on forests with rows without a tree, a subtree size that does not add up, a row that does not parse, OUT
    indices beyond out_len, "no function" nodes, a population that is no multiple of a batch.  IEEE-exact function sets are also held
    against the oracle; sets with library functions within their sensitivity envelope."""
    from evogp_amd import _lib

    pop, V = 20011, 5
    stored_outs = max(out_len + 2, 2) if out_len > 1 else 1
    v, t, s = oracle.generate(pop, L, V, stored_outs, 0.6 if out_len > 1 else 0.0, 0.5, [L, out_len], depth2leaf(7 if L > 64 else 5, 0.15), roulette_uniform(funcs), CS)
    v, t, s = v.copy(), t.copy(), s.copy()
    s[17, 0] = 0; s[18, 0] = -3; s[40000 % pop, 0] = 0          # rows without a tree: NaN
    r = int(np.nonzero(s[:, 0] >= 5)[0][0]); s[r, 1] += 1        # a subtree size that does not add up: the register kernels
    trunc = int(np.nonzero(s[:, 0] >= 7)[0][3]); s[trunc, 0] -= 2   # a truncated tree: it does not parse (NaN; the reference reads beyond it)
    nofn = 0
    for r in np.nonzero((s[:, 0] < min(L, 64) - 2) & (s[:, 0] >= 1))[0][200:240]:   # "no function" (unary, id 29) over whole trees
        n = int(s[r, 0])
        v[r, 1:n + 1] = v[r, :n].copy(); t[r, 1:n + 1] = t[r, :n].copy(); s[r, 1:n + 1] = s[r, :n].copy()
        v[r, 0] = 29.0; t[r, 0] = 2; s[r, 0] = n + 1
        nofn += 1
    assert nofn >= 30
    if L > 64:
        assert (s[:, 0] > 64).sum() > 5, "no tree beyond 64 nodes: the long-row case is not covered"
    X = rng.uniform(0.2, 2.5, (300, V)).astype(np.float32); y = rng.uniform(-3, 3, (300, out_len)).astype(np.float32)
    words, skipped = {}, {}
    try:
        for batch in (0, 8, 32, -1):
            assert _lib.lib.evogp_hip_debug_compile_batch(batch) == 0
            words[batch] = g.sr_fitness(v, t, s, X, y).view(np.uint32).copy()
            skipped[batch] = handler_histogram(g, pop)["skip"]
    finally:
        _lib.lib.evogp_hip_debug_compile_batch(-1)
    for batch, w in words.items():
        diff = np.nonzero(w != words[0])[0]
        if exact or out_len > 1:   # (multi-output programs are the same words; so are single-output ones of IEEE-exact functions)
            assert len(diff) == 0, f"batch {batch}: {len(diff)} fitness words differ from the general compiler's, first trees {diff[:5]}: {w[diff[:5]]} vs {words[0][diff[:5]]}"
        else:   # constants under library functions are folded at one level more than compile_general folds them: the library's value either way
            a, b = w.view(np.float32).astype(np.float64), words[0].view(np.float32).astype(np.float64)
            fin = np.isfinite(a) & np.isfinite(b)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(a[fin], b[fin], rtol=1e-4, atol=1e-30), f"batch {batch} against the general compiler"
            assert len(diff) <= 0.02 * pop, f"batch {batch}: {len(diff)} fitness words differ from the general compiler's"
        assert skipped[batch] <= skipped[0] + 2, f"batch {batch}: {skipped[batch]} trees left to the register kernels, {skipped[0]} by the general compiler"
    got = words[-1].view(np.float32)
    # (with IF in the set a tree of seven levels can outgrow the row: the generator then announces more nodes than the row holds, the
    # engine answers NaN for the cut-off tree, the reference reads on into the next row)
    cut = np.nonzero(s[:, 0] > L)[0]
    gone = np.concatenate([[17, 18, 40000 % pop, trunc], cut]).astype(np.int64)
    assert np.isnan(got[gone]).all()
    keep = np.ones(pop, bool); keep[gone] = False
    if exact:
        assert_close_classes(got[keep], oracle.sr_fitness(*(a[keep] for a in (v, t, s)), X, y, True), RTOL, what=f"packed lines, out_len={out_len}")
    else:
        fo = tuple(a[keep] for a in (v, t, s))
        want, tol, unstable = per_tree_tolerance(oracle, fo, X, y)
        assert_within_sensitivity(got[keep], want, tol, unstable, f"packed lines with library functions, out_len={out_len}", max_unstable=0.15, min_tight=0.2)


def _leaning_forest(rng, pop, L, var_len, funcs, right_funcs, lean_left=True, max_levels=60):
    """trees that lean to one side: level k is f_k(level k - 1, small) (lean_left) or f_k(small, level k - 1), `small` a function of two
    leaves.  In the interpreter's order -- the LAST operand first -- a left-leaning tree keeps one value per level on the operand
    stack while the levels below run (example/uci_sr.py's populations look like this after 30 generations)."""
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)

    def leaf(out):
        if rng.random() < 0.6:
            out.append((0, float(rng.integers(0, var_len)), 1))
        else:
            out.append((1, float(rng.choice(CS)), 1))

    def small(out):
        out.append((3, float(rng.choice(right_funcs)), 3)); leaf(out); leaf(out)

    for r in range(pop):
        levels = int(rng.integers(3, max_levels + 1))
        levels = min(levels, (L - 3) // 4)
        nodes = []
        for k in range(levels, 0, -1):          # prefix order: the outermost level first
            size = 4 * k + 3
            nodes.append((3, float(rng.choice(funcs)), size))
            if not lean_left:
                small(nodes)
        small(nodes)                             # level 0
        if lean_left:
            for _ in range(levels):
                small(nodes)
        n = len(nodes)
        assert n == 4 * levels + 3 and n <= L
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
    return v, t, s


@pytest.mark.parametrize("arith_only", [False, True], ids=["generic", "arith"])
@pytest.mark.parametrize("lean_left", [True, False], ids=["left", "right"])
def test_leaning_trees_deeper_than_the_register_stack_are_reordered(g, oracle, rng, lean_left, arith_only):
    """Left-leaning trees of up to 60 levels need up to 60 operand-stack entries in the interpreter's order; compile_general's second
    pass (the larger subtree of a binary function first, SWAP in front of a non-commutative function whose operands came in the
    other order) keeps every one of them in the threaded code: no SKIP records, SWAP words present, the oracle's values."""
    # (arith: + - * / alone -- the straight-line staged compiler of round 5, compile_long_arith; generic: compile_general's passes)
    funcs = [ADD, SUB, MUL, DIV] if arith_only else [ADD, SUB, MUL, DIV, MAX, LT, LDIV]
    forest = _leaning_forest(rng, 1500, 256, 5, funcs, [ADD, SUB, MUL, DIV], lean_left=lean_left)
    X = rng.uniform(-3, 3, (700, 5)).astype(np.float32); y = rng.uniform(-3, 3, (700, 1)).astype(np.float32)
    h = check(g, oracle, forest, X, y, f"leaning {'left' if lean_left else 'right'}", max_skipped=0.0)
    if lean_left:
        assert h["swap"] > 1000, h["swap"]
    else:
        assert h["swap"] == 0, "a right-leaning tree runs in the interpreter's own order"


def test_trig_high_on_the_stack_with_huge_operands_is_reordered_instead_of_bailing_out(g, oracle, rng):
    """sin / cos / tan of operands of 2^17 and more run the device library's whole function in the registers above the operand stack
    (gen_tc_asm.py triglib_*); a stack of seven and more entries reaches into them and the tree bails out at RUN time to the register
    kernels (an evolved example/uci_sr.py population: up to 8 % of its trees, more than half of the call's time).  In a forest of long
    rows the compilers reorder such a tree -- larger subtree first -- although its stack would fit: the trig node then runs low.
    Trees f_k(.. f_1(trig(x0 * 3e5), small_1) .., small_k), k = 6 .. 8 (natural height 7 .. 9), bare (at most 36 nodes: the one-chunk
    compiler hands them on) and under 20 right-leaning levels (116 nodes: the staged compiler directly)."""
    import torch

    from evogp_amd import _lib
    from evogp_amd.tree import Forest

    L, V, pop = 256, 4, 900
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)
    for r in range(pop):
        k, wrap = 6 + r % 3, 20 * ((r // 3) % 2)
        small = lambda: [(3, float(rng.choice([ADD, SUB, MUL])), 3), (0, float(rng.integers(0, V)), 1), (0, float(rng.integers(0, V)), 1)]
        nodes = [(2, float([SIN, COS, TAN][r % 3]), 4), (3, float(MUL), 3), (0, 0.0, 1), (1, 3e5, 1)]
        for _ in range(k):      # left-leaning: the trig node's level runs LAST in the interpreter's order, on top of k values
            nodes = [(3, float(rng.choice([ADD, SUB, MUL])), len(nodes) + 4)] + nodes + small()
        for _ in range(wrap):   # right-leaning: adds nodes, no height
            nodes = [(3, float(rng.choice([ADD, SUB])), len(nodes) + 4)] + small() + nodes
        assert len(nodes) == 4 + 4 * (k + wrap)
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
        assert oracle.validate_tree(t[r], s[r]) == 0
    X = rng.uniform(-3, 3, (1024, V)).astype(np.float32); y = rng.uniform(-3, 3, (1024, 1)).astype(np.float32)
    forest = Forest(V, 1, torch.from_numpy(v).to(g.DEV), torch.from_numpy(t).to(g.DEV), torch.from_numpy(s).to(g.DEV))
    Xd, yd = torch.from_numpy(X).to(g.DEV), torch.from_numpy(y).to(g.DEV)
    _lib.check(_lib.lib.evogp_hip_debug_profile(2), "profile")          # the call stops behind the threaded code: marked trees keep their sentinel
    try:
        words = forest.SR_fitness(Xd, yd).view(torch.int32).cpu().numpy()
    finally:
        _lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile")
    left = (words == 0x7FC0FEED) | (words == 0x7FC0BEEF) | (words == 0x7FC0DEED)
    assert not left.any(), f"{int(left.sum())} of {pop} trees were left to the register kernels (bare: {int(left[(np.arange(pop) // 3) % 2 == 0].sum())})"
    got = g.sr_fitness(v, t, s, X, y)
    be = g.batch_evaluate(v, t, s, X, 1).astype(np.float64)             # the register kernels call the same library
    with np.errstate(all="ignore"):
        ref = ((be - y[None, :, :].astype(np.float64)) ** 2).sum(2).mean(1)
    ok = np.isfinite(ref) & (np.abs(ref) < 1e30)
    assert ok.mean() > 0.5
    assert np.allclose(got[ok], ref[ok], rtol=2e-5, atol=1e-30), "threaded code vs the register kernels on the same trees"


# ---- the functions behind the generic stubs -------------------------------------------------------------------------------
@pytest.mark.parametrize("funcs", [[LDIV, ADD, MUL], [MAX, MIN, ADD, SUB], [LT, GT, LE, GE, ADD, MUL], [IF, ADD, SUB, LT], [LINV, INV, ADD, MUL],
                                   EXACT_WIDE], ids=["ldiv", "maxmin", "cmp", "if", "linv", "all-exact"])
def test_generic_functions(g, oracle, rng, funcs):
    mlc = 4 if IF in funcs else 5   # a full ternary tree of depth 4 has 40 nodes, of depth 5 121: it must fit the 64-node row
    forest = oracle.generate(6000, 64, 4, 1, 0.5, 0.4, [sum(funcs), 1], depth2leaf(mlc, 0.15), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5, 2.0, 1e-10, -1e-10])
    X = rng.uniform(-2, 2, (520, 4)).astype(np.float32)
    X[::7, 0] = 0.0; X[::11, 1] = np.float32(1e-10); X[::13, 2] = -0.0   # the loose functions' clamp, exact zeros
    y = rng.uniform(-2, 2, (520, 1)).astype(np.float32)
    check(g, oracle, forest, X, y, f"funcs {funcs}")


def test_generic_functions_every_operand_form(g, oracle, rng):
    """hand-built two-level trees: op(x, y) with x, y each a stack value, a variable or a constant (the eight forms)"""
    ops = [LDIV, MAX, MIN, LT, GT, LE, GE]
    rows = []
    for f in ops:
        for la in "SVC":
            for rb in "SVC":
                def operand(kind, var, const):
                    if kind == "S":
                        return [(3, float(ADD), 3), (0, float(var), 1), (1, const, 1)]   # ADD(var, const)
                    return [(0, float(var), 1)] if kind == "V" else [(1, const, 1)]
                a, b = operand(la, 0, 0.5), operand(rb, 1, -0.25)
                rows.append([(3, float(f), 1 + len(a) + len(b))] + a + b)
    # IF with every mix of leaf / subtree operands
    for mix in range(8):
        kids = []
        for j in range(3):
            kids += [(3, float(SUB), 3), (0, float(j), 1), (1, 0.1 * j, 1)] if (mix >> j) & 1 else [(0, float(j), 1)]
        rows.append([(4, float(IF), 1 + len(kids))] + kids)
    rows.append([(2, 29.0, 4), (3, float(ADD), 3), (0, 0.0, 1), (0, 1.0, 1)])   # unknown unary id over a subtree: 0
    rows.append([(2, 29.0, 2), (0, 2.0, 1)])                                    # ... over a leaf
    L = 16
    pop = len(rows)
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)
    for r, nodes in enumerate(rows):
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
        assert oracle.validate_tree(t[r], s[r]) == 0
    X = rng.uniform(-1, 1, (300, 3)).astype(np.float32); X[:40, 1] = 0.0; X[40:60, 1] = np.float32(-1e-10)
    y = rng.uniform(-1, 1, (300, 1)).astype(np.float32)
    check(g, oracle, (v, t, s), X, y, "operand forms", max_skipped=0.0)


def test_pow_with_a_deciding_constant_operand(g, oracle):
    """pow(x, 0) = 1, pow(1, y) = 1, pow(x, 1) = x, pow(x, -1) = 1 / x are compiled without the library's sequence
    (csrc/sr_tc.hip pow_fold_kind; scripts/ubench/pow_identities.hip ran all 2^32 operands through the device library).  One
    datapoint per special operand, labels 0, mean absolute error: the fitness IS |value|, compared bit for bit with the exact
    result (which is what the oracle's host library returns) -- for a variable operand, a subtree operand (ADD(x, -0) keeps
    every x but +0 -> its bits; x + x doubles) and a constant one, single- and multi-output."""
    xs = np.array([0.0, -0.0, 1.0, -1.0, 2.5, -2.5, 1e-30, -1e-30, 3e38, -3e38, np.inf, -np.inf, np.nan, 1e-42, 0.1], np.float32)
    cases = []   # (nodes, function of x giving the expected value)
    def S(var): return [(3, float(ADD), 3), (0, float(var), 1), (0, float(var), 1)]                     # x + x
    with np.errstate(all="ignore"):
        for c, f in ((0.0, lambda x: np.float32(1)), (-0.0, lambda x: np.float32(1)), (1.0, lambda x: x), (-1.0, lambda x: np.float32(1) / x)):
            cases.append(([(3, float(POW), 3), (0, 0.0, 1), (1, c, 1)], f))                              # pow(x, c)
            cases.append(([(3, float(POW), 5)] + S(0) + [(1, c, 1)], (lambda f_: lambda x: f_(x + x))(f)))   # pow(x + x, c)
            cases.append(([(3, float(POW), 3), (1, 2.5, 1), (1, c, 1)], (lambda f_: lambda x: f_(np.float32(2.5)) + 0 * np.float32(0))(f)))  # pow(2.5, c)
            cases.append(([(3, float(POW), 3), (1, 0.0, 1), (1, c, 1)], (lambda f_: lambda x: f_(np.float32(0.0)))(f)))   # pow(0, c)
        cases.append(([(3, float(POW), 3), (1, 1.0, 1), (0, 0.0, 1)], lambda x: np.float32(1)))          # pow(1, x)
        cases.append(([(3, float(POW), 5), (1, 1.0, 1)] + S(0), lambda x: np.float32(1)))                # pow(1, x + x)
        # folded nodes inside a larger tree: pow(x, 1) * pow(x + x, 0) - pow(x, -1)
        big = [(3, float(SUB), 13), (3, float(MUL), 9), (3, float(POW), 3), (0, 0.0, 1), (1, 1.0, 1), (3, float(POW), 5)] + S(0) + [(1, 0.0, 1)] + \
              [(3, float(POW), 3), (0, 0.0, 1), (1, -1.0, 1)]
        cases.append((big, lambda x: x * np.float32(1) - np.float32(1) / x))
    L = 16
    pop = len(cases)
    v = np.zeros((pop, L), np.float32); t = np.zeros((pop, L), np.int16); s = np.zeros((pop, L), np.int16)
    for r, (nodes, _) in enumerate(cases):
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
        assert oracle.validate_tree(t[r], s[r]) == 0
    y = np.zeros((1, 1), np.float32)
    for x in xs:
        X = np.array([[x]], np.float32)
        got = g.sr_fitness(v, t, s, X, y, False)
        assert handler_histogram(g, pop)["skip"] == 0
        with np.errstate(all="ignore"):
            want = np.array([np.abs(np.float32(f(np.float32(x)))) for _, f in cases], np.float32)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (float(x), np.flatnonzero(~same), got[~same], want[~same])
        # and the oracle (host pow) agrees with the exact values
        assert_close_classes(got, oracle.sr_fitness(v, t, s, X, y, False), 1e-6, what=f"x={x}")
    # multi-output: the same functions as OUT nodes over leaves
    vm = np.zeros((4, L), np.float32); tm = np.zeros((4, L), np.int16); sm = np.zeros((4, L), np.int16)
    for r, c in enumerate((0.0, 1.0, -1.0, 2.0)):
        nodes = [(3 | 0x80, np.array([POW | (1 << 16)], np.uint32).view(np.float32)[0], 3), (0, 0.0, 1), (1, c, 1)]
        for i, (ty, val, sz) in enumerate(nodes):
            tm[r, i], vm[r, i], sm[r, i] = ty, val, sz
    Xm = np.array([[0.5], [-3.0], [0.0], [7.0]], np.float32); ym = np.zeros((4, 2), np.float32)
    assert_close_classes(g.sr_fitness(vm, tm, sm, Xm, ym, True), oracle.sr_fitness(vm, tm, sm, Xm, ym, True), 1e-6, what="multi-output pow folds")


@pytest.mark.parametrize("D", [8, 100, 600], ids=["K1", "K4", "K8"])
@pytest.mark.parametrize("zeros", [False, True])
def test_division_in_place_and_gather_forms_at_every_stack_height(g, oracle, rng, D, zeros):
    """S / S, S / c and c / S divisions have an in-place handler (operands read where they are, temporaries in the stack entries
    above them) that the compiler picks where those entries are free, and a gather form elsewhere; a block with a zero divisor
    is handed from the first to the second.  Trees ADD^p(DIV(..), S1 .. Sp) put the division at stack height p + 2 for every p
    up to beyond the register stack of each build, with and without zero divisors in the rows."""
    def S(i, j, f=MUL): return [(3, float(f), 3), (0, float(i), 1), (0, float(j), 1)]
    rows = []
    for p in range(0, 15):
        for div in ([(3, float(DIV), 7)] + S(0, 1, ADD) + S(2, 3, SUB),        # (x0 + x1) / (x2 - x3)
                    [(3, float(DIV), 5), (1, 1.5, 1)] + S(2, 3, SUB),          # 1.5 / (x2 - x3)
                    [(3, float(DIV), 5)] + S(0, 1, ADD) + [(1, 3.0, 1)],       # (x0 + x1) / 3
                    [(2, float(INV), 4)] + S(2, 3, SUB)):                      # inv(x2 - x3)
            nodes = div
            for q in range(p):
                nodes = [(3, float(ADD), 1 + len(nodes) + 3)] + nodes + S(q % 4, (q + 1) % 4)
            rows.append(nodes)
    L = 64
    assert max(len(r) for r in rows) <= L
    v = np.zeros((len(rows), L), np.float32); t = np.zeros((len(rows), L), np.int16); s = np.zeros((len(rows), L), np.int16)
    for r, nodes in enumerate(rows):
        for i, (ty, val, sz) in enumerate(nodes):
            t[r, i], v[r, i], s[r, i] = ty, val, sz
        assert oracle.validate_tree(t[r], s[r]) == 0
    X = rng.uniform(-3, 3, (D, 4)).astype(np.float32); y = rng.uniform(-3, 3, (D, 1)).astype(np.float32)
    if zeros:
        X[::7, 3] = X[::7, 2]   # x2 - x3 = 0 in every seventh row: NaN there (forward.cu:183-187), so NaN fitness for these trees
        X[1, :] = 0.0
    for m in (True, False):
        assert_close_classes(g.sr_fitness(v, t, s, X, y, m), oracle.sr_fitness(v, t, s, X, y, m), RTOL, what=f"D={D} zeros={zeros} mse={m}")
    h = handler_histogram(g, len(rows))
    assert h["divip_SS"] > 0 and h["divip_CS"] > 0 and h["divip_SC"] > 0, h
    if D > 64:   # (the one-row build's 44-entry stack has room above every one of these trees)
        assert h["div_SS"] > 0 and h["div_CS"] > 0, h


# ---- multi-output trees ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("out_len,D,funcs,L", [(2, 1024, ARITH, 64), (4, 1024, ARITH, 64), (6, 700, ARITH, 64), (4, 1024, EXACT_WIDE, 128),
                                               (10, 200, ARITH, 128), (12, 1797, ARITH, 64), (3, 8, ARITH, 64), (4, 1024, ARITH + [SIN, EXP, LOG], 64),
                                               # 7-10 outputs over more than 256 rows: the 8-row interpreter's wide-stack build
                                               (8, 1024, ARITH, 64), (10, 1797, EXACT_WIDE, 128), (7, 600, ARITH + [SIN, EXP, LOG], 64)])
def test_multi_output(g, oracle, rng, out_len, D, funcs, L):
    mlc = 5 if IF in funcs else 6   # full trees must fit the row: ternary depth 5 = 121 nodes, binary depth 6 = 63
    forest = oracle.generate(5000, L, 7, out_len, 0.5, 0.5, [out_len, D], depth2leaf(mlc, 0.15), roulette_uniform(funcs), CS)
    X = rng.uniform(-2, 2, (D, 7)).astype(np.float32); y = rng.uniform(-2, 2, (D, out_len)).astype(np.float32)
    trans = any(f in (SIN, EXP, LOG) for f in funcs)
    if trans:  # library functions: a looser bar on this one case (operands are leaves: no amplification through the tree)
        got, want = g.sr_fitness(*forest, X, y), oracle.sr_fitness(*forest, X, y)
        assert handler_histogram(g, 5000)["skip"] <= 100
        assert_close_classes(got, want, 1e-4, what="multi-output with library functions")
    else:
        h = check(g, oracle, forest, X, y, f"out_len={out_len} D={D}")
        assert h["acc_s"] > 0 and h["mo_begin"] == 5000 - h["skip"]


def test_multi_output_out_index_beyond_out_len_and_nested_outs(g, oracle, rng):
    """OUT nodes whose stored index is >= out_len add nothing (forward.cu:239); OUT nodes nested in OUT nodes pass their LAST
    operand upward, not their result (forward.cu:241-243)"""
    out_len = 3
    forest = oracle.generate(3000, 128, 5, 6, 0.9, 0.5, [6, 6], depth2leaf(5, 0.1), roulette_uniform(ARITH + [IF, MAX]), CS)  # indices 0..5 stored
    X = rng.uniform(-2, 2, (300, 5)).astype(np.float32); y = rng.uniform(-2, 2, (300, out_len)).astype(np.float32)
    check(g, oracle, forest, X, y, "stored out indices up to 5, out_len 3")


@pytest.mark.parametrize("funcs,out_len", [([ADD, MUL, POW, TANH], 1), ([ADD, SUB, LPOW, SINH, COSH], 1), ([ADD, SUB, LOG, SQRT, POW, DIV, INV], 1),
                                           ([ADD, MUL, POW, TANH, SINH], 3)], ids=["pow-tanh", "lpow-sinh-cosh", "vis.ipynb", "3 outputs"])
def test_library_functions_in_the_threaded_code(g, oracle, rng, funcs, out_len):
    """pow, loose pow and the hyperbolic functions run the device library's transcribed sequences row by row
    (gen/ocml_bodies.py).  Function-level accuracy is pinned in test_gpu_ulp.py; here whole trees against the oracle (host libm):
    identical NaN sets, and 1e-4 on all but the ill-conditioned few (pow amplifies an ulp of its base by its exponent)."""
    forest = oracle.generate(4000, 64, 4, out_len, 0.5, 0.5, [len(funcs), out_len], depth2leaf(5, 0.15), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5, 2.0])
    X = rng.uniform(0.1, 2, (300, 4)).astype(np.float32); y = rng.uniform(-2, 2, (300, out_len)).astype(np.float32)
    got = g.sr_fitness(*forest, X, y)
    h = handler_histogram(g, 4000)
    assert h["skip"] <= 0.05 * 4000, f"{h['skip']} trees left to the register kernels"
    want, tol, unstable = per_tree_tolerance(oracle, forest, X, y)     # 1e-5 + the tree's own sensitivity to 3 ulp: EVERY tree
    ok = np.isfinite(want)
    assert_within_sensitivity(got, want, tol, unstable, "library functions vs oracle", min_tight=0.2)
    # the same trees on the register kernels (which call the library): the threaded code must agree to the last bit on every
    # datapoint-independent tree and to rounding of the final sum elsewhere
    be = g.batch_evaluate(*forest, X, out_len).astype(np.float64)
    with np.errstate(all="ignore"):
        ref = ((be - y[None, :, :].astype(np.float64)) ** 2).sum(2).mean(1)
    both = np.isfinite(ref) & ok & (np.abs(ref) < 1e30) & ~unstable
    # (2e-5, or the tree's own sensitivity to 3-ulp changes of its library calls: the threaded code's compiler decides pow(x, 1),
    # pow(x, -1) ... with the correctly rounded value where the library's powf is within an ulp of it, sr_tc.hip pow_fold_kind)
    err = np.abs(got[both].astype(np.float64) - ref[both])
    grant = np.maximum(2e-5 * np.abs(ref[both]) + 1e-30, tol[both])
    worst = int(np.argmax(err - grant))
    assert (err <= grant).all(), (f"threaded code vs the register kernels on the same trees: tree {np.flatnonzero(both)[worst]} {got[both][worst]!r} "
                                  f"vs {ref[both][worst]!r}, granted {grant[worst]:.3g}; {(err > grant).sum()} trees beyond")


# ---- small datasets: one row per lane (the K = 1 interpreter) --------------------------------------------------------------
@pytest.mark.parametrize("D", [1, 2, 7, 8, 33, 63, 64, 65])
@pytest.mark.parametrize("funcs,out_len,L,mlc", [(ARITH, 1, 64, 6), (EXACT_WIDE, 1, 128, 4), ([ADD, SUB, MUL, MAX, NEG], 4, 64, 5)], ids=["arith", "wide-L128", "4 outputs"])
def test_small_datasets(g, oracle, rng, D, funcs, out_len, L, mlc):
    """Datasets of at most 64 rows (the reference's XOR-3d examples have 8) run the interpreter variant with ONE row per lane;
    65 rows is the first size that takes the four-row variant.  IEEE-exact function sets: 1e-5 on every tree, both losses."""
    forest = oracle.generate(3000, L, 3, out_len, 0.5, 0.5, [D, L], depth2leaf(mlc, 0.1), roulette_uniform(funcs), CS)
    X = rng.uniform(-3, 3, (D, 3)).astype(np.float32); y = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
    check(g, oracle, forest, X, y, f"D={D}", max_skipped=0.05)


def test_small_dataset_library_functions_match_the_register_kernels(g, oracle, rng):
    """the reference's notebook configuration (XOR-3d: 8 rows, + - log sqrt pow / inv, L = 128): identical NaN / inf classes
    against the oracle, and agreement with the register kernels (which call the device library) on the same trees"""
    funcs = [ADD, SUB, LOG, SQRT, POW, DIV, INV]
    forest = oracle.generate(4000, 128, 3, 1, 0.5, 0.5, [8, 128], depth2leaf(5, 0.15), roulette_uniform(funcs), [-1.0, 0.0, 1.0])
    X = np.array([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], dtype=np.float32)
    y = (X.sum(1) % 2)[:, None].astype(np.float32)
    got = g.sr_fitness(*forest, X, y)
    h = handler_histogram(g, 4000)
    assert h["skip"] <= 0.05 * 4000, f"{h['skip']} trees left to the register kernels"
    want, tol, unstable = per_tree_tolerance(oracle, forest, X, y)
    ok = np.isfinite(want)
    assert_within_sensitivity(got, want, tol, unstable, "notebook set vs oracle", min_tight=0.2)
    be = g.batch_evaluate(*forest, X, 1).astype(np.float64)
    with np.errstate(all="ignore"):
        ref = ((be - y[None, :, :].astype(np.float64)) ** 2).sum(2).mean(1)
    both = np.isfinite(ref) & ok & (np.abs(ref) < 1e30)
    assert np.allclose(got[both], ref[both], rtol=2e-5, atol=1e-30)


# ---- classification epilogue on the threaded code ---------------------------------------------------------------------------------
@pytest.mark.parametrize("var_len,out_len,D", [(4, 2, 8), (4, 3, 64), (4, 3, 65), (4, 3, 200), (3, 5, 512), (4, 3, 700), (64, 10, 300), (64, 10, 1797), (30, 6, 2500),
                                               (3, 2, 20000)])   # (the last: three pieces of 9216 rows, the label sort over twenty passes of its one workgroup)
def test_classifier_count_on_the_threaded_code(g, oracle, rng, var_len, out_len, D):
    """evogp_hip_batch_argmax_count through compiled programs and the END_CLS handler (one column of int32 class labels staged
    behind X, per row the arg-max over the output accumulators as torch.argmax(clip(softmax(x))) sees it, hits counted per tree;
    datasets beyond LDS in pieces whose counts add up) against the torch rule on the oracle's outputs.  IEEE-exact function set
    with divisions: rows with NaN / infinite outputs (arg-max 0) occur; exact ties between soft-max probabilities that differ
    before rounding are the only legitimate difference."""
    import torch

    funcs = [ADD, SUB, MUL, DIV, MAX, NEG]
    pop = 1200
    f = oracle.generate(pop, 64, var_len, out_len, 0.5, 0.5, [D, out_len], depth2leaf(5, 0.1), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5])
    X = rng.uniform(-4, 4, (D, var_len)).astype(np.float32)
    X[rng.random((D, var_len)) < 0.05] = 0.0
    labels = rng.integers(0, out_len, D).astype(np.int32)
    got = g.batch_argmax_count(*f, X, labels, out_len)
    h = handler_histogram(g, pop)
    assert h["end_cls"] >= 0.9 * pop, f"only {h['end_cls']} of {pop} programs end in the classifier handler"
    outs_np = oracle.batch_evaluate(*f, X, out_len)                       # (pop, D, out); IEEE-exact functions: the device's bits
    # torch.argmax(clip(softmax(x))) computed by torch's own kernels on the device (classification.py:62-67): EQUALITY -- rows in
    # which two soft-max probabilities that differ before rounding come out equal are detected and recounted with aten's arithmetic
    want = torch_rule_counts(outs_np, labels)
    assert np.array_equal(got, want), (np.abs(got - want).max(), (got != want).mean(), np.flatnonzero(got != want)[:5])


@pytest.mark.parametrize("var_len,out_len,D,classes", [(5, 4, 150, "skewed"), (6, 10, 700, "sparse"), (4, 3, 1300, "out-of-range"), (3, 2, 40, "one")])
def test_classifier_rows_in_label_order(g, oracle, rng, var_len, out_len, D, classes):
    """The interpreter launches of a classifier call stage the rows sorted by label (tc_label_groups_kernel) and END_CLS judges a
    SEGMENT of lanes against its scalar label.  Label distributions that shape the segments: one class with almost all rows, classes with a
    handful of rows each (several segments per row register), labels outside [0, out_len) (-3, out_len, 1000: never a hit,
    classification.py:62-75 compares an index below out_len with them), a single class.  Outputs that tie exactly (two outputs carrying
    the same variable or constant: the FIRST maximum wins) are frequent in these forests of few variables."""
    funcs = [ADD, SUB, MUL, DIV, MAX, NEG]
    pop = 1500
    f = oracle.generate(pop, 64, var_len, out_len, 0.5, 0.5, [D, out_len + 1], depth2leaf(5, 0.1), roulette_uniform(funcs), [-1.0, 0.0, 1.0, 0.5])
    X = rng.uniform(-4, 4, (D, var_len)).astype(np.float32)
    X[rng.random((D, var_len)) < 0.05] = 0.0
    if classes == "skewed":
        labels = np.where(rng.random(D) < 0.9, 1, rng.integers(0, out_len, D))
    elif classes == "sparse":
        labels = np.repeat(np.arange(out_len), [3, 90, 1, 64, 65, 200, 2, 127, 128, 20])[:D]
        labels = np.concatenate([labels, np.full(D - len(labels), 5)])
        rng.shuffle(labels)
    elif classes == "out-of-range":
        labels = rng.choice(np.array([-3, 0, 1, 2, out_len, 1000]), D)
    else:
        labels = np.full(D, 1)
    labels = labels.astype(np.int32)
    got = g.batch_argmax_count(*f, X, labels, out_len)
    h = handler_histogram(g, pop)
    assert h["end_cls"] >= 0.9 * pop, h
    want = torch_rule_counts(oracle.batch_evaluate(*f, X, out_len), labels)
    assert np.array_equal(got, want), (classes, np.abs(got - want).max(), (got != want).mean(), np.flatnonzero(got != want)[:5])


def test_divisions_by_a_variable_read_the_reciprocal_columns(g, oracle, rng):
    """SHORT division mode: a launch whose dataset lies in [2^-46, 2^46] stages the variables' reciprocals behind the labels and its
    divisions by a variable (S / v, v / w, c / v, inv(v)) read them (divr_* words, gen_tc_asm.py); with a zero anywhere in the dataset
    the same words run the division's own handlers.  Both must meet the oracle as the IEEE mode does, and the two modes may differ
    by rounding only."""
    pop, L, var_len = 20000, 64, 6
    for funcs, what in ((ARITH, "arith"), (ARITH + [NEG, INV], "unary")):
        f = oracle.generate(pop, L, var_len, 1, 0.5, 0.5, [3, 9], depth2leaf(6), roulette_uniform(funcs), CS)
        y = rng.uniform(-2, 2, (300, 1)).astype(np.float32)
        for zero in (False, True):
            X = (rng.uniform(0.3, 3, (300, var_len)) * rng.choice([-1.0, 1.0], (300, var_len))).astype(np.float32)
            if zero:
                X[17, 2] = 0.0
            want = oracle.sr_fitness(*f, X, y)
            got = g.sr_fitness(*f, X, y)
            h = handler_histogram(g, pop)
            assert h["divr_SV"] > 0 and h["divr_VV"] > 0 and h["divr_CV"] > 0, h
            assert g.L.evogp_hip_set_sr_division(0) == 0
            try:
                ieee = g.sr_fitness(*f, X, y)
                assert sum(handler_histogram(g, pop)[k] for k in ("divr_SV", "divr_VV", "divr_CV")) == 0
                assert g.L.evogp_hip_set_sr_division(1) == 0
                fast = g.sr_fitness(*f, X, y)   # (FAST differs from SHORT only in blocks with an operand outside [2^-46, 2^46]: rare here)
                assert handler_histogram(g, pop)["divr_VV"] > 0
            finally:
                assert g.L.evogp_hip_set_sr_division(2) == 0
            same = (fast.view(np.uint32) == got.view(np.uint32)) | (np.isnan(fast) & np.isnan(got))
            assert same.mean() > 0.995, f"{what} zero={zero}: FAST differs from SHORT in {(~same).sum()} of {pop} trees"
            assert_close_classes(ieee, want, RTOL, what=f"{what} zero={zero}: IEEE division")
            assert_close_classes(got, want, RTOL, what=f"{what} zero={zero}: reciprocal columns")
            assert_close_classes(got, ieee, 1e-6, what=f"{what} zero={zero}: SHORT against IEEE")


@pytest.mark.parametrize("bad_column", [None, 3, 17])
def test_reciprocal_columns_with_more_variables_than_trust_bits(g, oracle, rng, bad_column):
    """Twenty variables (the interpreter keeps one "whole column in range" bit for the first fourteen only; the reciprocal copy needs EVERY
    column in range): all in range -> the divisions by a variable read reciprocals; a zero in column 3 or in column 17 -> the same words run
    the division's own handlers.  The oracle's fitness either way."""
    pop, L, var_len, D = 8000, 64, 20, 512
    f = oracle.generate(pop, L, var_len, 1, 0.5, 0.5, [var_len, 2], depth2leaf(6), roulette_uniform(ARITH), CS)
    X = (rng.uniform(0.3, 3, (D, var_len)) * rng.choice([-1.0, 1.0], (D, var_len))).astype(np.float32)
    y = rng.uniform(-2, 2, (D, 1)).astype(np.float32)
    if bad_column is not None:
        X[100, bad_column] = 0.0
    got = g.sr_fitness(*f, X, y)
    h = handler_histogram(g, pop)
    assert h["divr_SV"] > 0 and h["divr_VV"] > 0 and h["divr_CV"] > 0, h
    assert_close_classes(got, oracle.sr_fitness(*f, X, y), RTOL, what=f"20 variables, zero in column {bad_column}")
    if bad_column is not None:   # x / 0 is NaN (forward.cu:183-187): the trees that divide by that variable in that row
        assert np.isnan(got).sum() > 0


def test_chunked_pipeline_on_a_large_population(g, oracle):
    """a population beyond the sizes of the other tests (and, with EVOGP_TC_CHUNKS set, the chunked two-stream pipeline of
    sr_tc.hip): the result must equal that of the halves run on their own, and the oracle's on a sample"""
    pop = 450_000
    forest = g.generate(pop, 64, 10, 1, 0.5, 0.5, [7, 7], depth2leaf(6), roulette_uniform(ARITH + [SIN, MAX]), CS)
    X, y = c2_dataset()
    full = g.sr_fitness(*forest, X, y)
    h = pop // 2
    halves = np.concatenate([g.sr_fitness(*(a[:h] for a in forest), X, y), g.sr_fitness(*(a[h:] for a in forest), X, y)])
    assert np.array_equal(full.view(np.uint32), halves.view(np.uint32))
    again = g.sr_fitness(*forest, X, y)
    assert np.array_equal(full.view(np.uint32), again.view(np.uint32)), "run-to-run reproducible"
    pick = np.arange(0, pop, 1013)
    want, tol, unstable = per_tree_tolerance(oracle, tuple(a[pick] for a in forest), X, y)
    assert_within_sensitivity(full[pick], want, tol, unstable, "sample of the large forest vs oracle (sin in the set)", min_tight=0.2)


@pytest.mark.parametrize("D,var_len,out_len,funcs", [(5000, 10, 1, ARITH), (12000, 10, 1, ARITH), (20001, 6, 1, ARITH + [MAX, NEG]), (6000, 8, 4, ARITH),
                                                      (3000, 40, 1, ARITH), (9000, 10, 1, ARITH + [SIN, TAN])])
def test_datasets_larger_than_lds_run_in_pieces(g, oracle, rng, D, var_len, out_len, funcs):
    """the dataset of a call lives in LDS (150 KiB); a larger one is run in pieces over the same programs, the interpreter adding
    each piece's error sums to the fitness words (sr_tc.hip run_tc, gen_tc_asm.py batch finalisation).  The last piece is
    ragged on every tile.  With tan in the set a tree can bail out to the register kernels in ANY piece."""
    pop = 1500
    forest = oracle.generate(pop, 64, var_len, out_len, 0.5, 0.5, [D % 1000, var_len], depth2leaf(5, 0.15), roulette_uniform(funcs), CS)
    X = rng.uniform(-3, 3, (D, var_len)).astype(np.float32); y = rng.uniform(-3, 3, (D, out_len)).astype(np.float32)
    if SIN in funcs:
        X[D - 50:, 0] = 3e5   # beyond 2^17 in the LAST piece only: run-time bail-out after earlier pieces stored partial sums
    for mse in (True, False):
        got = g.sr_fitness(*forest, X, y, mse)
        if mse:
            h = handler_histogram(g, pop)
            assert h["skip"] <= 0.02 * pop, f"{h['skip']} trees left to the register kernels: the pieces path was not taken"
        want = oracle.sr_fitness(*forest, X, y, mse)
        if SIN in funcs:
            ok = np.isfinite(want)
            assert np.array_equal(np.isnan(got), np.isnan(want))
            # (operands of 3e5: one ulp of the operand is 0.03 rad, so the device library and the host libm differ visibly on the
            # trees that see them; function-level accuracy is pinned in test_gpu_ulp.py, here the pieces / bail-out mechanics)
            assert (np.abs(got[ok] - want[ok]) <= 1e-4 * np.abs(want[ok]) + 1e-6).mean() > 0.9
        else:
            assert_close_classes(got, want, RTOL, what=f"D={D} mse={mse}")


# ---- the division's block paths (gen_tc_asm.py DIVRANGE / TRUST): every operand form x every kind of block ---------------------
DIV_SPECIALS = np.array([0.0, -0.0, 1.0, -3.0, 0.1, 1e-20, -1e-15, 2.0**-46, 2.0**-47, 2.0**46, -2.0**47, 1e30, -3e38, np.inf, np.nan, 1e-42],
                        np.float32)


def _division_forest(form, n):
    """n * n trees `a / b`: a (index i) and b (index j) are a Stack operand (x + -0: every value but the sign of a zero passes
    through, and the compiler cannot fold it), a Variable or a Constant (DIV_SPECIALS[index]), as the two letters of `form` say"""
    L = 8
    rows = []
    for i in range(n):
        for j in range(n):
            def operand(kind, idx):
                if kind == "S":
                    return [(3, float(ADD), 3), (0, float(idx), 1), (1, -0.0, 1)]
                if kind == "V":
                    return [(0, float(idx), 1)]
                return [(1, float(DIV_SPECIALS[idx]), 1)]
            a, b = operand(form[0], i), operand(form[1], j)
            rows.append([(3, float(DIV), 1 + len(a) + len(b))] + a + b)
    v = np.zeros((len(rows), L), np.float32); t = np.zeros((len(rows), L), np.int16); s = np.zeros((len(rows), L), np.int16)
    for r, nodes in enumerate(rows):
        for k, (ty, val, sz) in enumerate(nodes):
            t[r, k], v[r, k], s[r, k] = ty, val, sz
    return v, t, s


@pytest.mark.parametrize("D", [64, 200, 512, 1100, 1536])
def test_division_blocks_of_every_kind(g, oracle, D):
    """`a / b` in the eight operand forms of the interpreter's division handlers (S / S, S / c and c / S in place; s / v, v / s,
    v / w, c / v, v / c through the banks) over datasets that make every kind of 64 x K-row block: all operands in
    [2^-46, 2^46] (the rows without range scaling, with and without TRUSTED variables), a numerator or denominator that is
    zero everywhere, one special value among ordinary ones, specials everywhere.  D covers the three interpreter builds
    (K = 1, 4, 8), one / two (ragged) / three tiles.  Mean absolute error against labels 0 = mean |a / b| per tree, compared
    with the oracle: identical NaN / inf classes, 1e-6 on the values (all terms are >= 0: only the rounding of the sum differs)."""
    rng = np.random.default_rng(D)
    n = len(DIV_SPECIALS)
    y = np.zeros((D, 1), np.float32)
    ordinary = rng.uniform(0.5, 4.0, (D, n)).astype(np.float32) * rng.choice(np.float32([-1.0, 1.0]), (D, n))
    uniform = np.tile(DIV_SPECIALS[None, :], (D, 1))
    shuffled = DIV_SPECIALS[(np.arange(n)[None, :] + 5 * np.arange(D)[:, None]) % n]
    one_row = ordinary.copy(); one_row[min(77, D - 1)] = DIV_SPECIALS
    zero_cols = ordinary.copy(); zero_cols[:, 0] = 0.0; zero_cols[:, 1] = -0.0; zero_cols[::2, 2] = 0.0
    for form in ("SS", "SC", "CS", "SV", "VS", "VV", "CV", "VC"):
        f = _division_forest(form, n)
        for name, X in (("ordinary", ordinary), ("uniform", uniform), ("shuffled", shuffled), ("one special row", one_row), ("zero columns", zero_cols)):
            X = np.ascontiguousarray(X, np.float32)
            want = oracle.sr_fitness(*f, X, y, use_mse=False)
            got = g.sr_fitness(*f, X, y, use_mse=False)
            assert handler_histogram(g, n * n)["skip"] == 0
            assert_close_classes(got, want, 1e-6, 0.0, f"{form}, {name} rows, D={D}")


def test_nan_fitness_words_are_the_canonical_nan(g, oracle):
    """A NaN sum leaves the interpreter as 0x7FC00000 whatever payloads and signs its operands carried (the payload an instruction
    hands on depends on its operand order; a fitness word must never look like one of the register kernels' sentinels)."""
    f = oracle.generate(5000, 64, 10, 1, 0.0, 0.5, [42, 0], depth2leaf(6), roulette_uniform(ARITH), [-1, 0, 1])
    X, y = c2_dataset()
    X = X.copy(); X[3, 4] = np.float32(np.nan); X[700, 2] = np.frombuffer(np.uint32(0xFFC12345).tobytes(), np.float32)[0]
    for m in (True, False):
        got = g.sr_fitness(*f, X, y, m)
        assert handler_histogram(g, 5000)["skip"] == 0   # (every tree ran in the interpreter)
        nan = np.isnan(got)
        assert nan.sum() > 1000
        assert (got.view(np.uint32)[nan] == 0x7FC00000).all()


def test_a_dataset_beyond_64_kib_of_lds_still_runs_in_the_interpreter(g, oracle, rng):
    """Ten outputs over 1024 rows of ten variables: 80 KiB of LDS (the kernel needs its dynamic-LDS ceiling raised for it).  A refused
    ceiling sends the call to the register kernels -- same results, a sixth of the speed -- so the result alone proves nothing:
    the call's stage timer must have seen the interpreter."""
    pop, L, var_len, out_len, D = 3000, 64, 10, 10, 1024
    f = oracle.generate(pop, L, var_len, out_len, 0.5, 0.5, [21, 4], depth2leaf(5), roulette_uniform(ARITH), CS)
    X = rng.uniform(-2, 2, (D, var_len)).astype(np.float32)
    y = rng.uniform(-2, 2, (D, out_len)).astype(np.float32)
    assert g.L.evogp_hip_debug_profile(1) == 0
    try:
        got = g.sr_fitness(*f, X, y, True)
        st = (ctypes.c_float * 3)(); n = ctypes.c_int(0)
        assert g.L.evogp_hip_debug_profile_read(st, ctypes.byref(n)) == 0
    finally:
        g.L.evogp_hip_debug_profile(0)
    assert n.value == 1 and st[1] > 0.0, f"the interpreter did not run (stage times {list(st)})"
    assert_close_classes(got, oracle.sr_fitness(*f, X, y, True), RTOL, 0.0, "ten outputs, 80 KiB of LDS")
