"""ctypes/numpy front end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/evogp_oracle.h).  Two back ends with the same call shapes:

    Oracle("port")       oracle/libevogp_oracle.so      plain-C restatement (evogp_oracle.c)
    Oracle("reference")  oracle/_ref/libevogp_ref.so    the reference's own device code compiled
                                                         for the host (build_ref.py)

All arrays are numpy, C-contiguous: value float32, type/size int16, indices int32, keys uint32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libevogp_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libevogp_ref.so")

_f32 = np.float32
_i16 = np.int16
_i32 = np.int32
_u32 = np.uint32


def build_port(force: bool = False) -> str:
    src = os.path.join(HERE, "evogp_oracle.c")
    if force or not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "libevogp_oracle.so"], check=True, capture_output=True)
    return PORT_SO


def _cpu_tag() -> str:
    """identifies the host CPU (model + ISA flags): a -march=native build is only valid on the machine that made it"""
    import hashlib

    try:
        lines = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    return hashlib.sha1("".join(lines).encode()).hexdigest()[:10]


def build_port_native() -> str | None:
    """The port compiled -O3 -march=native ON THIS HOST (the CPU baseline runs native code of the box it is timed
    on; the default library is -march=x86-64-v3 because it is built in the dev container and travels).  The file name carries
    the CPU's tag, so a library made elsewhere is never loaded.  None when no compiler is available."""
    so = os.path.join(HERE, f"libevogp_oracle_native_{_cpu_tag()}.so")
    src = os.path.join(HERE, "evogp_oracle.c")
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    cmd = ["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-o", so, src, "-lm"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
    except Exception:
        return None
    return so


def have_reference() -> bool:
    return os.path.exists(REF_SO)


def _p(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, kind: str = "port", native: bool = False):
        assert kind in ("port", "reference")
        self.kind = kind
        self.flags = "-O3 -march=x86-64-v3"
        if kind == "port":
            so = build_port_native() if native else None
            if so is not None:
                self.flags = "-O3 -march=native"
            self.lib = C.CDLL(so or build_port())
            self.pfx = "evogp_oracle_"
        else:
            if not have_reference():
                raise FileNotFoundError(f"{REF_SO} missing: run oracle/build_ref.py where /root/reference exists")
            self.lib = C.CDLL(REF_SO)
            self.pfx = "ref_"

    def _fn(self, name):
        return getattr(self.lib, self.pfx + name)

    # -- RNG -------------------------------------------------------------------------------
    def hash(self, n, k1, k2) -> int:
        f = self._fn("hash")
        f.restype = C.c_uint32
        return int(f(C.c_uint32(n), C.c_uint32(k1), C.c_uint32(k2)))

    def taus88(self, seed, count):
        out = np.zeros(count, _u32)
        fout = np.zeros(count, _f32)
        self._fn("taus88")(C.c_uint32(seed), C.c_int(count), out.ctypes.data_as(C.c_void_p), fout.ctypes.data_as(C.c_void_p))
        return out, fout

    # -- genetic operators -----------------------------------------------------------------
    def generate(self, pop, gp_len, var_len, out_len, out_prob, const_prob, keys, depth2leaf, roulette, consts,
                 tree_index_offset: int = 0):
        keys, pk = _p(keys, _u32)
        d2l, pd = _p(depth2leaf, _f32)
        rou, pr = _p(roulette, _f32)
        cs, pc = _p(consts, _f32)
        assert d2l.shape == (10,) and rou.shape == (29,) and keys.shape == (2,)
        # the reference leaves the tail uninitialised: pre-zero so outputs are deterministic
        v = np.zeros((pop, gp_len), _f32)
        t = np.zeros((pop, gp_len), _i16)
        s = np.zeros((pop, gp_len), _i16)
        args = [C.c_uint(pop), C.c_uint(gp_len), C.c_uint(var_len), C.c_uint(out_len), C.c_uint(cs.shape[0]),
                C.c_float(out_prob), C.c_float(const_prob), pk, pd, pr, pc,
                v.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)]
        if self.kind == "port":
            args.append(C.c_uint(tree_index_offset))
        else:
            assert tree_index_offset == 0, "the reference has no tree index offset"
        self._fn("generate")(*args)
        return v, t, s

    def mutate(self, value, type_, size, idx, nvalue, ntype, nsize):
        v, pv = _p(value, _f32); t, pt = _p(type_, _i16); s, ps = _p(size, _i16)
        nv, pnv = _p(nvalue, _f32); nt, pnt = _p(ntype, _i16); ns, pns = _p(nsize, _i16)
        ix, pix = _p(idx, _i32)
        pop, L = v.shape
        rv = np.zeros((pop, L), _f32); rt = np.zeros((pop, L), _i16); rs = np.zeros((pop, L), _i16)
        self._fn("mutate")(C.c_int(pop), C.c_int(L), pv, pt, ps, pix, pnv, pnt, pns,
                           rv.ctypes.data_as(C.c_void_p), rt.ctypes.data_as(C.c_void_p), rs.ctypes.data_as(C.c_void_p))
        return rv, rt, rs

    def crossover(self, value, type_, size, left_idx, right_idx, left_node, right_node):
        v, pv = _p(value, _f32); t, pt = _p(type_, _i16); s, ps = _p(size, _i16)
        li, pli = _p(left_idx, _i32); ri, pri = _p(right_idx, _i32)
        ln, pln = _p(left_node, _i32); rn, prn = _p(right_node, _i32)
        pop, L = v.shape
        n = li.shape[0]
        rv = np.zeros((n, L), _f32); rt = np.zeros((n, L), _i16); rs = np.zeros((n, L), _i16)
        self._fn("crossover")(C.c_int(pop), C.c_int(n), C.c_int(L), pv, pt, ps, pli, pri, pln, prn,
                              rv.ctypes.data_as(C.c_void_p), rt.ctypes.data_as(C.c_void_p), rs.ctypes.data_as(C.c_void_p))
        return rv, rt, rs

    # -- evaluation ------------------------------------------------------------------------
    def evaluate(self, value, type_, size, variables, out_len):
        v, pv = _p(value, _f32); t, pt = _p(type_, _i16); s, ps = _p(size, _i16)
        x, px = _p(variables, _f32)
        pop, L = v.shape
        assert x.shape[0] == pop
        res = np.zeros((pop, out_len), _f32)
        self._fn("evaluate")(C.c_uint(pop), C.c_uint(L), C.c_uint(x.shape[1]), C.c_uint(out_len), pv, pt, ps, px,
                             res.ctypes.data_as(C.c_void_p))
        return res

    def sr_fitness(self, value, type_, size, X, y, use_mse=True, threads=0):
        v, pv = _p(value, _f32); t, pt = _p(type_, _i16); s, ps = _p(size, _i16)
        X, pX = _p(X, _f32); y, py = _p(y, _f32)
        pop, L = v.shape
        D, var_len = X.shape
        out_len = y.shape[1]
        fit = np.zeros(pop, _f32)
        args = [C.c_uint(pop), C.c_uint(D), C.c_uint(L), C.c_uint(var_len), C.c_uint(out_len), C.c_int(int(use_mse)),
                pv, pt, ps, pX, py, fit.ctypes.data_as(C.c_void_p)]
        if self.kind == "port":
            args.append(C.c_int(threads))
        f = self._fn("sr_fitness")
        f.restype = C.c_int
        r = f(*args)
        self.threads_used = int(r) if self.kind == "port" else 1
        return fit

    def batch_evaluate(self, value, type_, size, X, out_len):
        assert self.kind == "port"
        v, pv = _p(value, _f32); t, pt = _p(type_, _i16); s, ps = _p(size, _i16)
        X, pX = _p(X, _f32)
        pop, L = v.shape
        D, var_len = X.shape
        res = np.zeros((pop, D, out_len), _f32)
        self._fn("batch_evaluate")(C.c_uint(pop), C.c_uint(D), C.c_uint(L), C.c_uint(var_len), C.c_uint(out_len),
                                   pv, pt, ps, pX, res.ctypes.data_as(C.c_void_p))
        return res

    def set_jitter(self, ulps: int, seed: int = 0) -> None:
        """Sensitivity probe (evogp_oracle.h): libm-backed results are moved by up to +-ulps; 0 switches it off."""
        assert self.kind == "port"
        self._fn("set_jitter")(C.c_int(ulps), C.c_uint(seed))

    def validate_tree(self, type_row, size_row) -> int:
        assert self.kind == "port"
        t, pt = _p(type_row, _i16); s, ps = _p(size_row, _i16)
        f = self._fn("validate_tree")
        f.restype = C.c_int
        return int(f(C.c_int(t.shape[0]), pt, ps))


def live_mask(size):
    """Boolean mask of the live prefix [0, size[:,0]) of every row."""
    size = np.asarray(size)
    return np.arange(size.shape[1])[None, :] < size[:, :1].astype(np.int64)


# ---- standard descriptors used by tests and bench ---------------------------
def roulette_uniform(func_ids):
    p = np.zeros(29, np.float64)
    for f in func_ids:
        p[f] = 1.0 / len(func_ids)
    return np.cumsum(p.astype(np.float32), dtype=np.float32)


def depth2leaf(max_layer_cnt, layer_leaf_prob=0.2):
    k = max_layer_cnt - 1
    return np.array([layer_leaf_prob] * k + [1.0] * (10 - k), np.float32)
