#!/usr/bin/env python3
"""Developer tool: transcribe the device math library's instruction sequences for use inside the threaded-code interpreter.

The interpreter (gen_tc_asm.py) is one assembly block, so it cannot call powf / sinhf / coshf.  Its handlers must nevertheless
return the LIBRARY's results, bit for bit — the register kernels call the library, and a tree has to evaluate to the same
value whichever kernel takes it (tests/test_gpu_ulp.py pins both against float64).  This script compiles one-line probe
kernels `o[i] = f(a[i] [, b[i]])` for gfx950 with the build's own flags, takes the straight-line body hipcc emits (everything
but the address arithmetic, the loads, the store and the waits on them), and rewrites every register as a placeholder:

    {v7}      a VGPR             {v2_3}   an aligned VGPR pair          {s6} / {s2_3}   the same for SGPRs

gen_tc_asm.py binds the placeholders to whatever registers are free inside a handler (pairs stay even-aligned pairs).
The result is written to ocml_bodies.py, which is committed: building the engine does not depend on this script, and a
new ROCm release changes the interpreter's arithmetic only when somebody re-runs it.

    python3 gen/ocml_transcribe.py            (needs /opt/rocm/bin/hipcc)
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only"]

PROBES = {
    "pow": ("powf(a[i], b[i])", 2),
    "sinh": ("sinhf(a[i])", 1),
    "cosh": ("coshf(a[i])", 1),
    # the WHOLE functions, Payne-Hanek reduction for operands of 2^17 and more included: the compiler keeps the two reductions apart
    # with exec masks (s_and_saveexec / s_andn2_saveexec / s_or exec) and skips a side nobody takes with s_cbranch_execz.  The masks
    # are kept (every lane gets the reduction the library gives it), the skips are dropped: the sequence is straight-line and gives the
    # library's value in every lane.  The interpreter's sin / cos / tan handlers run it row by row for a block that holds a large
    # operand (their own rows are the small-argument side alone).
    "sin": ("sinf(a[i])", 1),
    "cos": ("cosf(a[i])", 1),
    "tan": ("tanf(a[i])", 1),
}
MASKED = {"sin", "cos", "tan"}

# the sequences are compiler output of a third-party library: its notice travels with them
NOTICE = """THIRD-PARTY NOTICE.  The instruction sequences below are compiler output of AMD's ROCm-Device-Libs (the OCML math library:
powf, sinhf, coshf and the sequences the transcendental handlers of gen_tc_asm.py were transcribed from), distributed under
the University of Illinois/NCSA Open Source License.  The licence text that ships with ROCm
(/opt/rocm/share/doc/rocm-device-libs/LICENSE.TXT) is reproduced in gen/ROCM_DEVICE_LIBS_LICENSE.TXT next to this file:

    Copyright (c) 2014-2016, Advanced Micro Devices, Inc.  All rights reserved.
    Developed by: AMD Research and AMD HSA Software Development, Advanced Micro Devices, Inc., www.amd.com
"""

DROP = re.compile(r"^(s_load_|global_load_|global_store_|s_waitcnt|s_endpgm)")
REG = re.compile(r"\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b")


def compile_probe(name, expr, nargs):
    args = "const float* a, const float* b, float* o" if nargs == 2 else "const float* a, float* o"
    src = f'#include <hip/hip_runtime.h>\nextern "C" __global__ void k({args}) {{ int i = threadIdx.x; o[i] = {expr}; }}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.hip"), "w").write(src)
        subprocess.run([HIPCC] + FLAGS + ["p.hip", "-o", "p.s"], cwd=d, check=True, capture_output=True)
        text = open(os.path.join(d, "p.s")).read()
    body = text[text.index("\nk:"):]
    body = body[:body.index("s_endpgm")]
    return [ln.split(";")[0].strip() for ln in body.splitlines()[2:] if ln.strip() and not ln.strip().startswith(";")]


def transcribe(name, expr, nargs):
    lines = [ln for ln in compile_probe(name, expr, nargs) if ln]
    if name in MASKED:   # drop the skips (and their labels): exec masks alone decide who does what
        assert not any(ln.startswith(("s_branch", "s_cbranch_scc", "s_cbranch_vcc", "s_setpc")) for ln in lines), f"{name}: a real branch"
        lines = [ln for ln in lines if not ln.startswith("s_cbranch_exec") and not ln.endswith(":")]
    assert not any(ln.startswith((".", "s_cbranch", "s_branch")) or ln.endswith(":") for ln in lines), f"{name}: not straight-line"
    assert name in MASKED or not any("saveexec" in ln for ln in lines), f"{name}: exec masks"
    loads = [ln for ln in lines if ln.startswith("global_load_dword")]
    store = [ln for ln in lines if ln.startswith("global_store_dword")]
    assert len(loads) == nargs and len(store) == 1
    # operands of the probe: the load destinations in argument order (a is loaded from the first kernarg pointer), the stored value
    def first_reg(ln, k):
        return ln.split(None, 1)[1].split(",")[k].strip()
    ins = [first_reg(ln, 0) for ln in loads]
    addr = first_reg(loads[0], 1)           # byte offset register of the element (address arithmetic)
    if nargs == 2:  # order by the pointer pair each load uses: a = lower kernarg
        ptr = [first_reg(ln, 2) for ln in loads]
        ins = [r for _, r in sorted(zip(ptr, ins), key=lambda x: int(re.search(r"\d+", x[0]).group()))]
    out = first_reg(store[0], 1)
    keep = []
    for ln in lines:
        op = ln.split()[0]
        if DROP.match(op):
            continue
        if op.startswith("v_lshlrev_b32") and first_reg(ln, 0) == addr:
            continue  # element offset
        keep.append(ln)
    vpairs, spairs, vall, sall = set(), set(), set(), set()
    for ln in keep:
        rest = ln.split(None, 1)[1] if " " in ln else ""
        for m in REG.finditer(rest):
            if m.group(1):
                lo, hi = int(m.group(2)), int(m.group(3))
                assert hi == lo + 1 and lo % 2 == 0, ln
                (vpairs if m.group(1) == "v" else spairs).add(lo)
                (vall if m.group(1) == "v" else sall).update((lo, hi))
            else:
                (vall if m.group(4) == "v" else sall).add(int(m.group(5)))

    def sub(m):
        if m.group(1):
            return "{%s%d_%d}" % (m.group(1), int(m.group(2)), int(m.group(3)))
        return "{%s%d}" % (m.group(4), int(m.group(5)))

    tmpl = []
    for ln in keep:
        op, _, rest = ln.partition(" ")
        op = re.sub(r"_e32$", "", op)  # the assembler picks the short encoding itself
        tmpl.append((op + " " + REG.sub(sub, rest)).strip())
    reg = lambda r: int(re.search(r"\d+", r).group())
    return {
        "source": f"hipcc {' '.join(FLAGS[:4])}: o[i] = {expr}",
        "lines": tmpl,
        "vregs": sorted(vall), "vpairs": sorted(vpairs), "sregs": sorted(sall), "spairs": sorted(spairs),
        "inputs": [reg(r) for r in ins], "output": reg(out),
    }


def main():
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    out = ['"""GENERATED by gen/ocml_transcribe.py -- the device math library\'s instruction sequences with registers as',
           f'placeholders ({ver}).  Do not edit; re-run the script to refresh.', '', NOTICE + '"""', "BODIES = {"]
    for name, (expr, nargs) in PROBES.items():
        b = transcribe(name, expr, nargs)
        out.append(f"    {name!r}: {{")
        for k in ("source", "vregs", "vpairs", "sregs", "spairs", "inputs", "output"):
            out.append(f"        {k!r}: {b[k]!r},")
        out.append("        'lines': [")
        for ln in b["lines"]:
            out.append(f"            {ln!r},")
        out.append("        ],\n    },")
        print(name, len(b["lines"]), "instructions,", len(b["vregs"]), "VGPRs", b["vpairs"], len(b["sregs"]), "SGPRs", b["spairs"], "in", b["inputs"], "out", b["output"], file=sys.stderr)
    out.append("}")
    open(os.path.join(HERE, "ocml_bodies.py"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
