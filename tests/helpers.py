"""Shared test helpers: standard descriptors, random index batteries, comparison utilities."""
import numpy as np

from oracle.pyoracle import depth2leaf, live_mask, roulette_uniform  # noqa: F401

ARITH = [1, 2, 3, 4]                 # + - * /
PAPER7 = [1, 2, 3, 4, 14, 15, 16]    # + - * / sin cos tan (example/uci_sr.py:50)
ALLF = list(range(29))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def fbits(a):
    """Bit patterns of float32 data with every NaN mapped to one canonical pattern (the sign and
    payload of a NaN depend on the compiler's choice of operand order and carry no meaning)."""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return u


def assert_forest_equal(a, b, what="forest", live_only=False):
    """Bit-exact comparison of (value, type, size) triples; live_only compares [0, len) only
    (the reference leaves the tail uninitialised)."""
    assert np.array_equal(a[2][:, 0], b[2][:, 0]), f"{what}: tree lengths differ"
    m = live_mask(a[2]) if live_only else np.ones_like(a[2], dtype=bool)
    for name, x, y in zip(("value", "type", "size"), a, b):
        bx, by = bits(x), bits(y)
        if not np.array_equal(bx[m], by[m]):
            bad = np.argwhere((bx != by) & m)[0]
            raise AssertionError(f"{what}: {name} differs at tree {bad[0]} node {bad[1]}: {x[tuple(bad)]} vs {y[tuple(bad)]}")


def assert_close_classes(got, want, rtol, atol=0.0, what="values"):
    """Same NaN / +inf / -inf classes, and rtol/atol agreement on the finite values."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN sets differ ({np.isnan(got).sum()} vs {np.isnan(want).sum()})"
    assert np.array_equal(np.isposinf(got), np.isposinf(want)), f"{what}: +inf sets differ"
    assert np.array_equal(np.isneginf(got), np.isneginf(want)), f"{what}: -inf sets differ"
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin])
    tol = atol + rtol * np.abs(want[fin])
    if not (err <= tol).all():
        i = np.argmax(err - tol)
        raise AssertionError(f"{what}: max violation got {got[fin][i]!r} want {want[fin][i]!r} (rtol {rtol})")


def c2_dataset(D=1024, var_len=10, seed=1234):
    """X ~ U(-5,5), y = x0*x1 + x2*x3 - x4 + 0.5*x5^2."""
    r = np.random.default_rng(seed)
    X = r.uniform(-5, 5, (D, var_len)).astype(np.float32)
    y = (X[:, 0] * X[:, 1] + X[:, 2] * X[:, 3] - X[:, 4] + 0.5 * X[:, 5] ** 2).astype(np.float32)[:, None]
    return X, y


def random_crossover_indices(rng, sizes, n, invalid_frac=0.02):
    pop = len(sizes)
    li = rng.integers(0, pop, n).astype(np.int32)
    ri = rng.integers(0, pop, n).astype(np.int32)
    bad = rng.random(n) < invalid_frac
    ri[bad] = rng.choice([-1, -7, pop, pop + 5], bad.sum())
    ln = (rng.integers(0, 2**31 - 1, n) % sizes[li]).astype(np.int32)
    rn = (rng.integers(0, 2**31 - 1, n) % sizes[np.clip(ri, 0, pop - 1)]).astype(np.int32)
    return li, ri, ln, rn


def middle_child_roots(type_, size):
    """bool[pop][L]: positions that are the root of the MIDDLE operand of a ternary node.  Replacing a subtree exactly there is
    undefined in the reference: its ancestor walk (mutation.cu:66-82) reads ``subtree_size_stack[midTreeIndex]``, but only positions
    below the replaced node were copied to that stack (:30-35), so with midTreeIndex == old_node_idx the walk continues from
    whatever the uninitialised local memory holds (on the host build: sometimes an endless loop).  The oracle and the engine
    compute the evidently intended result; comparisons against the compiled reference avoid these positions."""
    type_, size = np.asarray(type_), np.asarray(size).astype(np.int64)
    pop, L = type_.shape
    live = np.arange(L)[None, :] < size[:, :1]
    tern = ((type_ & 0x7F) == 4) & live
    mask = np.zeros((pop, L), bool)
    t, i = np.nonzero(tern)
    first = i + 1
    ok = first < L
    t, first = t[ok], first[ok]
    mid = first + size[t, first]
    ok = mid < L
    mask[t[ok], mid[ok]] = True
    return mask


def avoid_middle_child_roots(type_, size, trees, positions):
    """positions with the undefined ones (see middle_child_roots) moved to the node in front of them (its left sibling's last node)"""
    positions = np.asarray(positions).copy()
    bad = middle_child_roots(type_, size)
    tr = np.clip(np.asarray(trees), 0, np.asarray(type_).shape[0] - 1)
    inside = (positions >= 0) & (positions < np.asarray(type_).shape[1])
    hit = np.zeros(positions.shape, bool)
    hit[inside] = bad[tr[inside], positions[inside]]
    positions[hit] -= 1
    return positions


# ---- per-entry tolerance for trees of library functions (device OCML vs host libm) -----------------------------------
def the_oracle():
    from oracle.pyoracle import Oracle

    return Oracle("port")


def sensitivity(oracle, fn, base_rtol=1e-5, ulps=3, seeds=4):
    """How far may a tree's result move when every libm-backed function result moves by up to `ulps` ulp?  The device library and
    the host libm both stay within ~2 ulp of the true value (tests/test_gpu_ulp.py pins the device side), so two correct
    evaluations of a tree differ by at most what such a perturbation does to THAT tree: well-conditioned trees must agree to
    1e-5 (the north star's bar), tan(tan(x)) may not.  The oracle's sensitivity probe (evogp_oracle_set_jitter) measures it per
    entry of `fn()` (an oracle evaluation).  -> (want, tolerance, unstable: the NaN / inf class itself hangs on an ulp)"""
    want = fn()
    spread = np.zeros(want.shape, dtype=np.float64)
    unstable = np.zeros(want.shape, bool)
    try:
        # random signs per call, plus the two systematic probes (every library result +ulps / -ulps): an entry that hangs on ONE
        # call (a single huge row under pow) would otherwise depend on the luck of four draws
        for seed in [1000 + i for i in range(seeds)] + [0xFFFFFFFF, 0xFFFFFFFE]:
            oracle.set_jitter(ulps, seed)
            j = fn()
            with np.errstate(all="ignore"):
                spread = np.maximum(spread, np.nan_to_num(np.abs(j.astype(np.float64) - want), nan=0.0, posinf=0.0))
            unstable |= (np.isnan(j) != np.isnan(want)) | (np.isinf(j) != np.isinf(want))
    finally:
        oracle.set_jitter(0)
    # an entry that a 3-ulp nudge of its library calls moves by more than 1 % carries no digits (condition number beyond 3e4:
    # tan next to a pole, a / (sin - sin)): the perturbation is no longer in the linear regime "twice the spread" assumes, a
    # fourth ulp can cross the pole.  Such entries are compared by class only, like those whose class itself flips
    with np.errstate(all="ignore"):
        unstable |= spread > 0.01 * np.abs(want.astype(np.float64))
    return want, base_rtol * np.abs(want.astype(np.float64)) + 2.0 * spread, unstable


def assert_within_sensitivity(got, want, tol, unstable, name, max_unstable=0.1, min_tight=0.3):
    """EVERY entry within its own tolerance — no allowed-bad fraction.  Entries whose class flips under the perturbation are
    compared by class membership only (there must be few of them); the grant must not be a blank cheque: for a good share of
    the finite entries it is the 1e-5 bar itself, give or take the rounding of the perturbed runs."""
    got, want = np.asarray(got, np.float64), np.asarray(want)
    assert got.shape == want.shape
    stable = ~unstable
    assert unstable.mean() <= max_unstable, f"{name}: {unstable.sum()} of {unstable.size} entries have an ulp-dependent NaN/inf class"
    assert np.array_equal(np.isnan(got[stable]), np.isnan(want[stable])), f"{name}: NaN sets differ on ulp-stable entries"
    assert np.array_equal(np.isinf(got[stable]), np.isinf(want[stable])), f"{name}: inf sets differ on ulp-stable entries"
    fin = stable & np.isfinite(want)
    if not fin.any():
        return
    err = np.abs(got[fin] - want[fin].astype(np.float64))
    worst = np.argmax(err - tol[fin])
    assert (err <= tol[fin]).all(), (f"{name}: entry {np.flatnonzero(fin.ravel())[worst]} off by {err[worst]:.6g} (value {want[fin][worst]:.6g}, "
                                     f"granted {tol[fin][worst]:.6g}); {(err > tol[fin]).sum()} entries beyond their own sensitivity")
    tight = (tol[fin] <= 1e-4 * np.abs(want[fin]) + 1e-12).mean()
    assert tight >= min_tight, f"{name}: the granted tolerance is below 1e-4 relative for only {tight:.0%} of the finite entries"


def per_tree_tolerance(oracle, forest, X, y, base_rtol=1e-5, ulps=3, seeds=4, use_mse=True):
    return sensitivity(oracle, lambda: oracle.sr_fitness(*forest, X, y, use_mse), base_rtol, ulps, seeds)


# ---- the Classification problem's prediction rule, computed by torch ON THE DEVICE (the reference runs it there) ------------------
def torch_rule_counts(outs_np, labels, block=256):
    """counts[t] = #rows whose torch.argmax(torch.clip(torch.softmax(outs[t]), 1e-15, 1 - 1e-15)) equals the label
    (src/evogp/problem/classification.py:62-67), with torch's own kernels on cuda:0 — the definition the fused count must meet
    with EQUALITY (near-ties between soft-max probabilities included)."""
    import torch

    lab = torch.from_numpy(np.asarray(labels).astype(np.int64)).cuda()
    out = []
    for i in range(0, outs_np.shape[0], block):
        o = torch.from_numpy(np.ascontiguousarray(outs_np[i:i + block])).cuda()
        pred = torch.argmax(torch.clip(torch.softmax(o, dim=2), 1e-15, 1 - 1e-15), dim=2)
        out.append((pred == lab[None, :]).sum(1).cpu())
    return torch.cat(out).numpy()
