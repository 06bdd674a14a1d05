#!/usr/bin/env python3
"""Generate interp_asm_<name>.inc: the interpreter core of the SR fitness kernel in gfx950 assembly.

Why assembly.  Micro-benchmarks on MI355X (scripts/ubench/issue_latency.hip, profiles/) show that a single
wave issues one instruction every ~4.3 shader-clock ticks whatever the instruction, that a taken direct
branch costs ~16 extra ticks, a computed jump (s_setpc_b64) ~28-60, and a v_readlane -> SALU -> v_readlane
round trip ~28.  The interpreter is therefore bound by INSTRUCTIONS PER NODE and taken jumps per node, and
the compiler's output (compare trees of ~6 taken branches, ~31 instructions per node) is 2-3x away from
what the hardware needs.  The generated block is ONE `asm volatile` statement with these properties:

  * the program is not fetched at all: the pre-decoder (C++, one node per lane) hands over six 64-bit
    BALLOT MASKS (bit j set = instruction j is VAR / CONST / ADD / SUB / MUL / DIV); the dispatch at the end
    of every handler is a chain of `s_bitcmp1_b64 mask, j` + `s_cbranch_scc1 handler` in order of expected
    frequency — direct branches only, exactly one taken branch per tree node, no compare tree, no loop
    counter (an instruction index with no mask bit is the end of the program);
  * K = 4 rows per lane; the operand stack and the lane's variables live in fixed VGPRs indexed through
    M0 (s_set_gpr_idx_on): stack slot e of row k is v[S0 + 4e + k], variable v of row k is v[V0 + 4v + k];
    binary operators read BOTH operands and write the result through one index window
    (`v_add_f32 v[S0+k], v[S0+4+k], v[S0+k]` with src0, src1 and dst indexed by 4(h-2)): 4 VALU per node;
  * division is IEEE (the sequence hipcc emits for `b == 0 ? NaN : a / b`, forward.cu:183-187).

  inputs   %[mvar] %[mconst] %[madd] %[msub] %[mmul] %[mdiv]   SGPR pairs, the ballot masks
           %[pay]   VGPR, lane j = payload of instruction j (constant bits, or 4 * variable index)
           %[lds]   VGPR, byte address in LDS of this lane's float4 of variable 0; variable v at + v*1024
  outputs  %[r0..r3] the four results (stack slot 0 of the lane's four rows)

Fixed registers (all clobbered): VGPR block [base, top): DIV temporaries, NaN, operand copies, VARS, STK;
s36 = j (instruction index), s37 = 4 * stack height, s43 = payload, s[44:45] scratch.
"""
import sys


def gen(name, depth, vla, base=32):
    D = [base + i for i in range(5)]      # division temporaries
    NAN = base + 5
    TA = [base + 6 + i for i in range(4)]  # operand a of the four rows (also the VAR staging registers)
    TB = [base + 10 + i for i in range(4)]  # operand b
    V0 = base + 14
    assert V0 % 2 == 0
    S0 = V0 + 4 * vla
    top = S0 + 4 * depth
    L = []
    a = L.append
    uid = "%="
    H = {n: f".Lh_{n}_{uid}" for n in ("var", "const", "add", "sub", "mul", "div", "end")}

    def dispatch(inc=True):
        if inc:
            a("s_add_u32 s36, s36, 1")
        for m, n in (("mvar", "var"), ("mconst", "const"), ("madd", "add"), ("msub", "sub"), ("mmul", "mul"), ("mdiv", "div")):
            a(f"s_bitcmp1_b64 %[{m}], s36")
            a(f"s_cbranch_scc1 {H[n]}")
        a(f"s_branch {H['end']}")

    for v in range(vla):
        a(f"ds_read_b128 v[{V0 + 4 * v}:{V0 + 4 * v + 3}], %[lds] offset:{1024 * v}")
    a(f"v_mov_b32 v{NAN}, 0x7fc00000")
    a("s_mov_b32 s36, 0")
    a("s_mov_b32 s37, 0")
    a("s_waitcnt lgkmcnt(0)")
    dispatch(inc=False)

    # VAR: stack[h] = vars[var]   (two index windows: source by variable, destination by height)
    a(f"{H['var']}:")
    a("v_readlane_b32 s43, %[pay], s36")
    a("s_set_gpr_idx_on s43, gpr_idx(SRC0)")
    for k in range(4):
        a(f"v_mov_b32 v{TA[k]}, v{V0 + k}")
    a("s_set_gpr_idx_on s37, gpr_idx(DST)")
    for k in range(4):
        a(f"v_mov_b32 v{S0 + k}, v{TA[k]}")
    a("s_set_gpr_idx_off")
    a("s_add_u32 s37, s37, 4")
    dispatch()
    # CONST: stack[h] = constant
    a(f"{H['const']}:")
    a("v_readlane_b32 s43, %[pay], s36")
    a("s_add_u32 s36, s36, 1")
    a("s_set_gpr_idx_on s37, gpr_idx(DST)")
    for k in range(4):
        a(f"v_mov_b32 v{S0 + k}, s43")
    a("s_set_gpr_idx_off")
    a("s_add_u32 s37, s37, 4")
    dispatch(inc=False)
    # ADD SUB MUL: stack[h-2] = stack[h-1] (left operand) op stack[h-2] (right operand)
    for n, op in (("add", "v_add_f32"), ("sub", "v_sub_f32"), ("mul", "v_mul_f32")):
        a(f"{H[n]}:")
        a("s_sub_u32 s37, s37, 8")
        a("s_set_gpr_idx_on s37, gpr_idx(SRC0,SRC1,DST)")
        for k in range(4):
            a(f"{op} v{S0 + k}, v{S0 + 4 + k}, v{S0 + k}")
        a("s_set_gpr_idx_off")
        a("s_add_u32 s37, s37, 4")
        dispatch()
    # DIV
    a(f"{H['div']}:")
    a("s_sub_u32 s37, s37, 8")
    a("s_set_gpr_idx_on s37, gpr_idx(SRC0)")
    for k in range(4):
        a(f"v_mov_b32 v{TA[k]}, v{S0 + 4 + k}")
        a(f"v_mov_b32 v{TB[k]}, v{S0 + k}")
    a("s_set_gpr_idx_off")
    d3, d4, d6, d7, d8 = D
    for k in range(4):
        x, y = TA[k], TB[k]
        a(f"v_div_scale_f32 v{d3}, s[44:45], v{y}, v{y}, v{x}")
        a(f"v_rcp_f32 v{d4}, v{d3}")
        a(f"v_div_scale_f32 v{d6}, vcc, v{x}, v{y}, v{x}")
        a(f"v_fma_f32 v{d7}, -v{d3}, v{d4}, 1.0")
        a(f"v_fmac_f32 v{d4}, v{d7}, v{d4}")
        a(f"v_mul_f32 v{d7}, v{d6}, v{d4}")
        a(f"v_fma_f32 v{d8}, -v{d3}, v{d7}, v{d6}")
        a(f"v_fmac_f32 v{d7}, v{d8}, v{d4}")
        a(f"v_fma_f32 v{d3}, -v{d3}, v{d7}, v{d6}")
        a(f"v_div_fmas_f32 v{d3}, v{d3}, v{d4}, v{d7}")
        a(f"v_div_fixup_f32 v{x}, v{d3}, v{y}, v{x}")
        a(f"v_cmp_neq_f32 vcc, 0, v{y}")
        a("s_nop 1")
        a(f"v_cndmask_b32 v{x}, v{NAN}, v{x}, vcc")
    a("s_set_gpr_idx_on s37, gpr_idx(DST)")
    for k in range(4):
        a(f"v_mov_b32 v{S0 + k}, v{TA[k]}")
    a("s_set_gpr_idx_off")
    a("s_add_u32 s37, s37, 4")
    dispatch()
    # END
    a(f"{H['end']}:")
    for k in range(4):
        a(f"v_mov_b32 %[r{k}], v{S0 + k}")
    body = "\n".join(f'    "{line}\\n\\t"' for line in L)
    clob = ['"memory"', '"vcc"', '"scc"', '"s36"', '"s37"', '"s43"', '"s44"', '"s45"'] + [f'"v{i}"' for i in range(base, top)]
    clob_txt = ",\n      ".join(", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12))
    out = f'''// GENERATED by gen/gen_interp_asm.py {name} (stack depth {depth}, {vla} variables, VGPRs v{base}..v{top - 1}) — do not edit.
#define EVOGP_ASM_TOP_{name.upper()} {top}
#define EVOGP_INTERP_ASM_{name.upper()}(r0_, r1_, r2_, r3_, mvar_, mconst_, madd_, msub_, mmul_, mdiv_, pay_, lds_) \\
  asm volatile( \\
'''
    out += "\n".join(line + " \\" for line in body.split("\n"))
    out += f'''
    : [r0] "=v"(r0_), [r1] "=v"(r1_), [r2] "=v"(r2_), [r3] "=v"(r3_) \\
    : [mvar] "s"(mvar_), [mconst] "s"(mconst_), [madd] "s"(madd_), [msub] "s"(msub_), [mmul] "s"(mmul_), [mdiv] "s"(mdiv_), \\
      [pay] "v"(pay_), [lds] "v"(lds_) \\
    : {clob_txt.replace(chr(10), " " + chr(92) + chr(10))})
'''
    return out


if __name__ == "__main__":
    outdir = sys.argv[1] if len(sys.argv) > 1 else "."
    for name, depth, vla in (("d10", 10, 10), ("d16", 16, 12)):
        with open(f"{outdir}/interp_asm_{name}.inc", "w") as f:
            f.write(gen(name, depth, vla))
        print("wrote", f"{outdir}/interp_asm_{name}.inc")
