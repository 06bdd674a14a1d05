#!/usr/bin/env python3
"""Two rows per pass through a transcribed library sequence (gen/ocml_bodies.py), for gen_tc_asm.py (round 5).

pow / sinh / cosh run ROW BY ROW through 120-190 instructions of compiler output, and while VGPR indexing is on every one of them
costs a full issue slot.  Most of those instructions are plain fp32 additions, multiplications and fused multiply-adds: the same
IEEE operations exist as v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 over a register PAIR, so two rows can share them -- the
treatment sin / cos / tan got by hand in round 4 (gen_tc_asm.py TRIGPK), here as a mechanical transformation of the body:

  * every virtual VGPR of the body becomes an aligned pair (row k in the low, row k + 1 in the high register);
  * v_add / v_sub / v_mul / v_fma / v_fmac / v_fmamk / v_fmaak / v_mov on registers become ONE packed instruction (a constant --
    literal, inline or SGPR -- goes through one scratch SGPR pair with op_sel_hi 0; sources with |abs| stay per row: VOP3P has none);
  * the packed instructions the compiler itself emitted (two-float arithmetic on adjacent registers) are taken apart into their two
    scalar operations first (op_sel / op_sel_hi / neg_lo / neg_hi) and then packed ACROSS the rows like the others;
  * everything else -- compares, selects, conversions, ldexp / frexp, the transcendental instructions -- is issued once per row; the
    comparison masks live in SGPR pairs, one per row (row 0 keeps vcc where the compiler used it, row 1 gets a pair of its own through
    the _e64 encodings);
  * registers are assigned by a linear scan over the straight line (values are renamed at every definition first), so the pairs need
    hardly more registers than the single rows did.

`check()` executes the original body (once per row) and the paired program symbolically and compares the expression trees of the
results: run as a script it verifies every body it is used for."""
import re

PACKABLE = {"v_add_f32": "add", "v_sub_f32": "sub", "v_mul_f32": "mul", "v_fma_f32": "fma", "v_fmac_f32": "fmac", "v_fmamk_f32": "fmamk",
            "v_fmaak_f32": "fmaak", "v_mov_b32": "mov"}
MODS = re.compile(r"\s+(op_sel_hi|op_sel|neg_lo|neg_hi):\[([^\]]*)\]")
INLINE = {"0", "1.0", "-1.0", "0.5", "-0.5", "2.0", "-2.0", "4.0", "-4.0"}


def _split(line):
    mods = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in MODS.finditer(line)}
    core = MODS.sub("", line).strip()
    op, _, rest = core.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    return op, ops, mods


class Opnd:
    """one source or destination: kind v (single VGPR), vp (VGPR pair), s, sp, vcc, imm"""

    def __init__(self, text):
        t = text.strip()
        self.neg = t.startswith("-") and "{" in t
        if self.neg:
            t = t[1:]
        self.abs = t.startswith("|")
        if self.abs:
            t = t.strip("|")
        m = re.fullmatch(r"\{([vs])(\d+)(?:_(\d+))?\}", t)
        if m:
            self.kind = m.group(1) + ("p" if m.group(3) else "")
            self.n = int(m.group(2))
        elif t == "vcc":
            self.kind, self.n = "vcc", None
        else:
            self.kind, self.n = "imm", t

    def is_const(self):
        return self.kind in ("imm", "s")


class Pairing:
    def __init__(self, body, name="body"):
        self.body, self.name = body, name
        self.out = []          # emitted lines with %...% tokens, each with its group number
        self.group = 0
        self.vver, self.sver, self.cver = {}, {}, {}   # current version of a virtual VGPR / shared SGPR / condition pair
        self.nv = self.ns = self.nc = 0
        self.pending_cvt = None
        self.invariant = {}    # version name -> literal: a VGPR loaded with a constant is the same in both rows -- ONE register
        self.wide = set()      # ... those that a packed instruction reads next to another constant: the low half of an aligned pair (one SGPR per instruction)

    # ---- renaming ----------------------------------------------------------------------------------------------------------------------
    def vuse(self, n):
        if n not in self.vver:   # (an input)
            self.vver[n] = f"V{n}in"
        return self.vver[n]

    def idef(self, n, literal):
        self.nv += 1
        self.vver[n] = f"I{n}d{self.nv}"
        self.invariant[self.vver[n]] = literal
        return self.vver[n]

    def vdef(self, n):
        self.nv += 1
        self.vver[n] = f"V{n}d{self.nv}"
        return self.vver[n]

    def suse(self, n):
        return self.sver[n]

    def sdef(self, n):
        self.ns += 1
        self.sver[n] = f"S{n}d{self.ns}"
        return self.sver[n]

    def cuse(self, n):
        return self.cver[n]

    def cdef(self, n):
        self.nc += 1
        self.cver[n] = f"C{n}d{self.nc}"
        return self.cver[n]

    def emit(self, text):
        self.out.append((self.group, text))

    # ---- one scalar fp operation over both rows ---------------------------------------------------------------------------------------------
    def packed(self, kind, dst_n, srcs):
        """srcs: list of (what, neg) with what = ('v', virtual number) or ('k', constant text or shared SGPR token)"""
        names, sel, neg = [], [], []
        nconst = sum(1 for what, _ in srcs if what[0] == "k")
        for what, ng in srcs:
            if what[0] == "v" and self.vuse(what[1]) in self.invariant:   # a constant that sits in a register
                if nconst == 0:
                    what = ("k", self.invariant[self.vuse(what[1])]); nconst += 1   # ... the literal itself
                else:                                                           # ... its register, the low half for both rows
                    self.wide.add(self.vuse(what[1]))
                    names.append(f"%{self.vuse(what[1])}.p%"); sel.append(0); neg.append(1 if ng else 0)
                    continue
            if what[0] == "v":
                names.append(f"%{self.vuse(what[1])}.p%"); sel.append(1)
            else:
                self.emit(f"s_mov_b32 %K.lo%, {what[1]}")
                names.append("%K.p%"); sel.append(0)
            neg.append(1 if ng else 0)
        d = self.vdef(dst_n)
        if kind == "mov":
            assert not neg[0]
            self.emit(f"v_pk_mov_b32 %{d}.p%, {names[0]}, {names[0]} op_sel:[0,1]" if sel[0] else f"v_pk_mov_b32 %{d}.p%, {names[0]}, {names[0]} op_sel:[0,0]")
            return
        ins = {"add": "v_pk_add_f32", "mul": "v_pk_mul_f32", "fma": "v_pk_fma_f32"}[kind]
        tail = ""
        if not all(sel):
            tail += f" op_sel_hi:[{','.join(map(str, sel))}]"
        if any(neg):
            tail += f" neg_lo:[{','.join(map(str, neg))}] neg_hi:[{','.join(map(str, neg))}]"
        self.emit(f"{ins} %{d}.p%, {', '.join(names)}{tail}")

    def src_of(self, o):
        if o.kind == "v":
            return ("v", o.n)
        if o.kind == "s":
            return ("k", f"%{self.suse(o.n)}%")
        assert o.kind == "imm", o.kind
        return ("k", o.n)

    def scalar_fp(self, op, ops):
        """a packable VOP2 / VOP3 instruction on single registers; returns False when an operand form is not packable"""
        kind = PACKABLE[op]
        o = [Opnd(x) for x in ops]
        if any(x.abs for x in o) or any(x.kind not in ("v", "s", "imm") for x in o):
            return False
        if sum(1 for x in o[1:] if x.is_const()) > 1 or o[0].kind != "v":
            return False
        d = o[0].n
        if kind == "mov":
            if o[1].kind == "imm":   # a constant into a register: one register for both rows
                self.emit(f"v_mov_b32 %{self.idef(d, o[1].n)}.0%, {o[1].n}")
                return True
            if o[1].kind != "v" or self.vuse(o[1].n) in self.invariant:
                return False
            self.packed("mov", d, [(self.src_of(o[1]), o[1].neg)])
        elif kind in ("add", "mul"):
            self.packed(kind, d, [(self.src_of(o[1]), o[1].neg), (self.src_of(o[2]), o[2].neg)])
        elif kind == "sub":
            self.packed("add", d, [(self.src_of(o[1]), o[1].neg), (self.src_of(o[2]), not o[2].neg)])
        elif kind == "fma":
            self.packed("fma", d, [(self.src_of(x), x.neg) for x in o[1:4]])
        elif kind == "fmac":      # d += a * b
            self.packed("fma", d, [(self.src_of(o[1]), o[1].neg), (self.src_of(o[2]), o[2].neg), (("v", d), False)])
        elif kind == "fmamk":     # d = a * K + c
            self.packed("fma", d, [(self.src_of(o[1]), o[1].neg), (self.src_of(o[2]), False), (self.src_of(o[3]), o[3].neg)])
        else:                     # fmaak: d = a * b + K
            self.packed("fma", d, [(self.src_of(o[1]), o[1].neg), (self.src_of(o[2]), o[2].neg), (self.src_of(o[3]), False)])
        return True

    # ---- an instruction once per row ------------------------------------------------------------------------------------------------------
    def per_row(self, op, ops):
        o = [Opnd(x) for x in ops]
        vopc_e32 = op.startswith("v_cmp") and not op.endswith("_e64")
        if op.startswith("v_cmp"):
            ndst = 1
        elif op.startswith("v_subbrev_co") or op.startswith("v_subb_co") or op.startswith("v_addc_co") or op.startswith("v_add_co") or op.startswith("v_sub_co"):
            ndst = 2
        else:
            ndst = 1
        # sources first (the versions in front of the instruction), then the definitions
        src_txt = [[None, None] for _ in o]
        uses_vcc_src = False
        for i, x in enumerate(o):
            if i < ndst:
                continue
            for r in (0, 1):
                if x.kind == "v":
                    t = f"%{self.vuse(x.n)}.{r}%"
                    t = f"|{t}|" if x.abs else t
                    src_txt[i][r] = ("-" if x.neg else "") + t
                elif x.kind == "s":
                    src_txt[i][r] = f"%{self.suse(x.n)}%"
                elif x.kind == "sp":
                    src_txt[i][r] = f"%{self.cuse(x.n)}.{r}%"
                elif x.kind == "vcc":
                    uses_vcc_src = True
                    src_txt[i][r] = "vcc" if r == 0 else "%VCC1%"
                elif x.kind == "imm":
                    src_txt[i][r] = x.n
                else:
                    raise ValueError((op, ops))
        dst_txt = [[None, None] for _ in range(ndst)]
        writes_vcc = False
        for i in range(ndst):
            x = o[i]
            if x.kind == "v":
                d = self.vdef(x.n)
                dst_txt[i] = [f"%{d}.0%", f"%{d}.1%"]
            elif x.kind == "sp":
                d = self.cdef(x.n)
                dst_txt[i] = [f"%{d}.0%", f"%{d}.1%"]
            elif x.kind == "vcc":
                writes_vcc = True
                dst_txt[i] = ["vcc", "%VCC1%"]
            else:
                raise ValueError((op, ops))
        for r in (0, 1):
            rop = op
            if r == 1 and (writes_vcc or uses_vcc_src) and not op.endswith("_e64"):   # row 1's mask is an SGPR pair of its own: the long encodings
                rop = op + "_e64"
                lits = [i for i, x in enumerate(o) if x.kind == "imm" and x.n not in INLINE and not re.fullmatch(r"-?\d+", x.n)]
                if lits:   # (VOP3 takes no literal)
                    assert len(lits) == 1
                    self.emit(f"s_mov_b32 %K.lo%, {o[lits[0]].n}")
                    src_txt[lits[0]][1] = "%K.lo%"
            parts = [dst_txt[i][r] for i in range(ndst)] + [src_txt[i][r] for i in range(ndst, len(o))]
            self.emit(f"{rop} {', '.join(parts)}")
        del vopc_e32

    # ---- the body ------------------------------------------------------------------------------------------------------------------------
    def run(self):
        for line in self.body["lines"]:
            self.group += 1
            op, ops, mods = _split(line)
            if op == "s_nop":
                self.emit(line)
            elif op in ("s_mov_b32", "s_movk_i32", "s_brev_b32"):
                d = Opnd(ops[0])
                assert d.kind == "s" and Opnd(ops[1]).kind == "imm", line
                self.emit(f"{op} %{self.sdef(d.n)}%, {ops[1]}")
            elif op in ("s_and_b64", "s_or_b64", "s_xor_b64", "s_andn2_b64", "s_orn2_b64"):
                o = [Opnd(x) for x in ops]
                src = [[("vcc" if r == 0 else "%VCC1%") if x.kind == "vcc" else f"%{self.cuse(x.n)}.{r}%" for r in (0, 1)] for x in o[1:]]
                dst = ["vcc", "%VCC1%"] if o[0].kind == "vcc" else None
                if dst is None:
                    d = self.cdef(o[0].n)
                    dst = [f"%{d}.0%", f"%{d}.1%"]
                for r in (0, 1):
                    self.emit(f"{op} {dst[r]}, {src[0][r]}, {src[1][r]}")
            elif op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_mov_b32"):
                o = [Opnd(x) for x in ops]
                assert o[0].kind == "vp" and all(x.kind in ("vp", "sp") for x in o[1:]), line
                n = len(o) - 1
                half = lambda x, h: ("v", x.n + h) if x.kind == "vp" else ("k", f"%{self.suse(x.n + h)}%")   # (an SGPR pair here is two constants)
                sel = mods.get("op_sel", [0] * n) + [0] * n
                selh = mods.get("op_sel_hi", [1] * n) + [1] * n
                ngl = mods.get("neg_lo", [0] * n) + [0] * n
                ngh = mods.get("neg_hi", [0] * n) + [0] * n
                if op == "v_pk_mov_b32":   # D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
                    lo = [(half(o[1], sel[0]), False)]
                    hi = [(half(o[2], sel[1]), False)]
                    kind = "mov"
                else:
                    lo = [(half(o[1 + i], sel[i]), bool(ngl[i])) for i in range(2)]
                    hi = [(half(o[1 + i], selh[i]), bool(ngh[i])) for i in range(2)]
                    kind = "add" if op == "v_pk_add_f32" else "mul"
                # both halves read the registers as they are IN FRONT of the instruction
                before = dict(self.vver)
                self.packed(kind, o[0].n, lo)
                after_lo = dict(self.vver)
                self.vver = dict(before)
                self.packed(kind, o[0].n + 1, hi)
                self.vver[o[0].n] = after_lo[o[0].n]
            elif op.startswith("v_cvt_f64_f32"):
                self.pending_cvt = (op, ops)   # (its 64-bit result feeds v_frexp_exp_i32_f64 alone: issued with that, row by row)
            elif op.startswith("v_frexp_exp_i32_f64"):
                cop, cops = self.pending_cvt
                self.pending_cvt = None
                src = Opnd(cops[1])
                assert Opnd(cops[0]).kind == "vp" and Opnd(ops[1]).kind == "vp" and Opnd(cops[0]).n == Opnd(ops[1]).n and src.kind == "v", line
                s = self.vuse(src.n)
                d = self.vdef(Opnd(ops[0]).n)
                for r in (0, 1):
                    t = f"%{s}.{r}%"
                    self.emit(f"{cop} %T64.p%, {'-' if src.neg else ''}{'|' + t + '|' if src.abs else t}")
                    self.emit(f"{op} %{d}.{r}%, %T64.p%")
            elif op.replace("_e64", "") in PACKABLE and self.scalar_fp(op.replace("_e64", ""), ops):
                pass
            elif op.startswith("v_"):
                assert not any(Opnd(x).kind == "vp" for x in ops), f"{self.name}: 64-bit operand in {line}"
                self.per_row(op, ops)
            else:
                raise ValueError(f"{self.name}: {line}")
        assert self.pending_cvt is None
        return self

    # ---- registers -----------------------------------------------------------------------------------------------------------------------
    def allocate(self, vpairs, spairs, verbose=False, vsingles=()):
        """vpairs: even VGPR numbers of free aligned pairs; vsingles: further single VGPRs; spairs: even SGPR numbers of free aligned
        pairs.  Returns the final lines, the registers of the inputs / output (pair bases) and what was used."""
        tok = re.compile(r"%([A-Za-z0-9]+)(?:\.(p|0|1|lo))?%")
        body = self.body
        last = {}
        for idx, (g, text) in enumerate(self.out):
            for m in tok.finditer(text):
                key = m.group(1) if m.group(1)[0] in "VSI" or m.group(1) in ("K", "T64", "VCC1") else f"{m.group(1)}.{m.group(2)}"
                last[key] = g
        out_name = self.vver[body["output"]]
        last[out_name] = 10 ** 9
        vfree, sfree_p, sfree_1, v1free = list(vpairs), list(spairs), [], list(vsingles)
        reg = {}
        fixed = {"K": sfree_p.pop(0), "VCC1": sfree_p.pop(0), "T64": vfree.pop(0) if any("T64" in t for _, t in self.out) else None}
        for n in body["inputs"]:
            reg[f"V{n}in"] = vfree.pop(0)
        peak_v = peak_s = peak_1 = 0
        lines = []
        live = {}   # key -> (class, register)
        for k in (f"V{n}in" for n in body["inputs"]):
            live[k] = ("v", reg[k])
        cur_group, to_free = None, []
        for idx, (g, text) in enumerate(self.out):
            if g != cur_group:   # registers whose last use lay in earlier groups come back only now: never inside the instruction that read them
                for key in [k for k in live if last.get(k, -1) < g]:
                    cls, r = live.pop(key)
                    (vfree if cls == "v" else v1free if cls == "v1" else sfree_p if cls == "sp" else sfree_1).append(r)
                cur_group = g

            def resolve(m):
                name, part = m.group(1), m.group(2)
                if name in fixed:
                    r = fixed[name]
                    return f"s[{r}:{r + 1}]" if (name in ("K", "VCC1") and part in (None, "p")) else (f"s{r}" if name == "K" else f"v[{r}:{r + 1}]")
                if name[0] == "I":   # one register for both rows
                    if name not in reg and name in self.wide:
                        assert vfree, f"{self.name}: out of VGPR pairs at '{text}'"
                        reg[name] = vfree.pop(0); live[name] = ("v", reg[name])
                    if name not in reg:
                        if not v1free:
                            assert vfree, f"{self.name}: out of VGPRs at '{text}'"
                            b = vfree.pop(0); v1free.extend([b, b + 1])
                        reg[name] = v1free.pop(0); live[name] = ("v1", reg[name])
                    return f"v[{reg[name]}:{reg[name] + 1}]" if part == "p" else f"v{reg[name]}"
                if name[0] == "V":
                    if name not in reg:
                        assert vfree, f"{self.name}: out of VGPR pairs at '{text}'"
                        reg[name] = vfree.pop(0); live[name] = ("v", reg[name])
                    r = reg[name]
                    return f"v[{r}:{r + 1}]" if part == "p" else f"v{r + int(part)}"
                if name[0] == "S":
                    if name not in reg:
                        if not sfree_1:
                            assert sfree_p, f"{self.name}: out of SGPRs at '{text}'"
                            b = sfree_p.pop(0); sfree_1.extend([b, b + 1])
                        reg[name] = sfree_1.pop(0); live[name] = ("s1", reg[name])
                    return f"s{reg[name]}"
                key = f"{name}.{part}"   # a condition pair of one row
                if key not in reg:
                    assert sfree_p, f"{self.name}: out of SGPR pairs at '{text}'"
                    reg[key] = sfree_p.pop(0); live[key] = ("sp", reg[key])
                r = reg[key]
                return f"s[{r}:{r + 1}]"
            lines.append(tok.sub(resolve, text))
            peak_v = max(peak_v, len(vpairs) - len(vfree))
            peak_1 = max(peak_1, len(vsingles) - len(v1free))
            peak_s = max(peak_s, 2 * (len(spairs) - len(sfree_p)) - len(sfree_1))
        if verbose:
            valu = lambda ls: sum(1 for ln in ls if ln.startswith("v_"))
            print(f"{self.name}: {valu(body['lines'])} vector instructions per row -> {valu(lines)} for two rows ({2 * valu(body['lines'])} row by row), "
                  f"{peak_v} VGPR pairs + {max(peak_1, 0)} singles, {peak_s} SGPRs")
        return {"lines": lines, "inputs": [reg[f"V{n}in"] for n in body["inputs"]], "output": reg[out_name], "vpairs_used": peak_v, "sgprs_used": peak_s,
                "k": fixed["K"], "vcc1": fixed["VCC1"]}


# ---- symbolic check -----------------------------------------------------------------------------------------------------------------------
class _Terms:
    """hash-consed expression DAG: a term is an integer, equal terms are equal integers (the trees themselves grow exponentially)"""

    def __init__(self):
        self.ids, self.defs = {}, []

    def mk(self, *key):
        if key[0] == "neg" and self.defs[key[1]][0] == "neg":          # -(-x) = x
            return self.defs[key[1]][1]
        if key[0] == "pair":                                           # the two halves of one 64-bit value read back as a pair
            lo, hi = self.defs[key[1]], self.defs[key[2]]
            if lo[0] == "lo" and hi[0] == "hi" and lo[1] == hi[1]:
                return lo[1]
        if key not in self.ids:
            self.ids[key] = len(self.defs)
            self.defs.append(key)
        return self.ids[key]


def _exec(lines, regs, T):
    """Execute straight-line gfx9 assembly on a register file of terms.  regs: {'v3': term, 's4': term, 'vcc': term, ...}; a 64-bit
    register pair is its two halves.  Only what the transcribed bodies and their paired forms use."""
    def rd(t, half=None):
        t = t.strip()
        neg = t.startswith("-") and not re.fullmatch(r"-[\d.]+", t)
        if neg:
            t = t[1:]
        ab = t.startswith("|")
        if ab:
            t = t.strip("|")
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", t)
        if m:
            if half is None:
                e = T.mk("pair", regs[f"{m.group(1)}{m.group(2)}"], regs[f"{m.group(1)}{int(m.group(2)) + 1}"])
            else:
                e = regs[f"{m.group(1)}{int(m.group(2)) + half}"]
        elif re.fullmatch(r"[vs]\d+", t) or t == "vcc":
            assert t in regs, f"read of an undefined register {t}"
            e = regs[t]
        else:
            e = T.mk("const", t)
        if ab:
            e = T.mk("abs", e)
        return T.mk("neg", e) if neg else e

    def wr(t, e, half=None):
        t = t.strip()
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", t)
        if m:
            if half is None:
                regs[f"{m.group(1)}{m.group(2)}"] = T.mk("lo", e); regs[f"{m.group(1)}{int(m.group(2)) + 1}"] = T.mk("hi", e)
            else:
                regs[f"{m.group(1)}{int(m.group(2)) + half}"] = e
        else:
            regs[t] = e

    for line in lines:
        op, ops, mods = _split(line)
        base = op.replace("_e64", "").replace("_e32", "")
        if base == "s_nop":
            continue
        if base in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
            n = len(ops) - 1
            sel = mods.get("op_sel", [0] * n); selh = mods.get("op_sel_hi", [1] * n)
            ngl = mods.get("neg_lo", [0] * n); ngh = mods.get("neg_hi", [0] * n)
            f = {"v_pk_add_f32": "add", "v_pk_mul_f32": "mul", "v_pk_fma_f32": "fma"}[base]
            res = []
            for ss, ng in ((sel, ngl), (selh, ngh)):
                args = []
                for i in range(n):
                    e = rd(ops[1 + i], ss[i])
                    args.append(T.mk("neg", e) if ng[i] else e)
                res.append(T.mk(f, *args))
            wr(ops[0], res[0], 0); wr(ops[0], res[1], 1)
        elif base == "v_pk_mov_b32":
            sel = mods.get("op_sel", [0, 0])
            lo, hi = rd(ops[1], sel[0]), rd(ops[2], sel[1])
            wr(ops[0], lo, 0); wr(ops[0], hi, 1)
        elif base == "v_sub_f32":
            wr(ops[0], T.mk("add", rd(ops[1]), T.mk("neg", rd(ops[2]))))
        elif base in ("v_add_f32", "v_mul_f32"):
            wr(ops[0], T.mk(base[2:5], rd(ops[1]), rd(ops[2])))
        elif base in ("v_fma_f32", "v_fmamk_f32", "v_fmaak_f32"):
            wr(ops[0], T.mk("fma", rd(ops[1]), rd(ops[2]), rd(ops[3])))
        elif base == "v_fmac_f32":
            wr(ops[0], T.mk("fma", rd(ops[1]), rd(ops[2]), rd(ops[0])))
        elif base in ("v_mov_b32", "s_mov_b32", "s_movk_i32"):
            wr(ops[0], rd(ops[1]))
        elif base == "s_brev_b32":
            wr(ops[0], T.mk("brev", rd(ops[1])))
        elif base.startswith("v_cvt_f64_f32"):
            wr(ops[0], T.mk("cvt_f64", rd(ops[1])))
        elif base.startswith("v_subbrev_co") or base.startswith("v_subb_co"):
            d0, d1 = T.mk(base, rd(ops[2]), rd(ops[3]), rd(ops[4])), T.mk("carry", rd(ops[2]), rd(ops[3]), rd(ops[4]))
            wr(ops[0], d0); wr(ops[1], d1)
        elif base.startswith("v_"):
            srcs = [rd(x) for x in ops[1:]]
            if base == "v_cndmask_b32" and len(ops) == 3:   # the short encoding reads vcc
                srcs.append(rd("vcc"))
            wr(ops[0], T.mk(base, *srcs))
        elif base in ("s_and_b64", "s_or_b64", "s_xor_b64", "s_andn2_b64", "s_orn2_b64"):
            wr(ops[0], T.mk(base, rd(ops[1]), rd(ops[2])))
        else:
            raise ValueError(line)
    return regs


def check(body, paired, name="body"):
    """the paired program computes, for each of its two rows, exactly the term the original body computes"""
    def fmt(ln):
        return re.sub(r"\{([vs])(\d+)(?:_(\d+))?\}", lambda m: f"{m.group(1)}[{m.group(2)}:{m.group(3)}]" if m.group(3) else f"{m.group(1)}{m.group(2)}", ln)
    T = _Terms()
    want = []
    for r in (0, 1):
        regs = {f"v{n}": T.mk("in", i, r) for i, n in enumerate(body["inputs"])}
        _exec([fmt(ln) for ln in body["lines"]], regs, T)
        want.append(regs[f"v{body['output']}"])
    regs = {}
    for i, p in enumerate(paired["inputs"]):
        regs[f"v{p}"] = T.mk("in", i, 0); regs[f"v{p + 1}"] = T.mk("in", i, 1)
    _exec(paired["lines"], regs, T)
    for r in (0, 1):
        assert regs[f"v{paired['output'] + r}"] == want[r], f"{name}: row {r} of the paired program differs from the body"
    return True


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ocml_bodies import BODIES
    for nm in ("pow", "sinh", "cosh"):
        pr = Pairing(BODIES[nm], nm).run()
        res = pr.allocate(list(range(100, 180, 2)), list(range(40, 100, 2)), verbose=True, vsingles=[9, 7])
        check(BODIES[nm], res, nm)
        print(f"  {nm}: the paired program equals the body on both rows (symbolic execution)")
