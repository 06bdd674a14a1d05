#!/usr/bin/env python3
"""Generate interp_asm_<name>.inc: the threaded-code interpreter core in gfx950 assembly.

Why assembly: the inner loop is bound by scalar/branch issue.  The AMDGPU back end
lowers `switch` and computed goto to compare-and-branch trees and cannot take `asm goto`, so a
jump-table dispatch has to be written by hand.  The generated block is ONE `asm volatile` statement:

  inputs   %[prog]  VGPR, lane j = byte offset of the handler of instruction j (handler index * 512),
                    lane n = offset of the END handler
           %[pay]   VGPR, lane j = payload of instruction j (constant bits, or 4 * variable index)
           %[lds]   VGPR, byte address in LDS of this lane's float4 of variable 0; variable v is at
                    + v * 1024 (64 lanes x 16 B)
  outputs  %[r0..r3] the four results (top of stack of the lane's four rows)

Register plan (fixed VGPRs, all listed as clobbers; K = 4 rows per lane):
  TMP  B0..B3 popped operand, D0..D4 division temporaries, NAN
  TOS  T0..T3 cached top of stack
  VARS [var][k]   variables of the lane's four rows, loaded from LDS at entry
  STK  [slot][k]  operand stack; element e (e < h-1) lives in slot e+1, a push at height h parks the
                  old top in slot h
Scalar: s36 = j (current instruction), s37 = 4 * stack height, s[38:39] = handler table base,
s[40:41] = jump target, s42 = next handler offset (prefetched), s43 = payload, s[44:45] scratch.

Every handler is a 512-byte slot: [prefetch next opcode] [body] [jump] — threaded code, one taken
jump per tree node, no compare tree, no loop counter (the program ends with an END instruction).
Handlers: 0 CONST, 1 VAR, 2 ADD, 3 SUB, 4 MUL, 5 DIV (IEEE division, NaN when the divisor is 0:
forward.cu:183-187), 6 END.  The division sequence is the one hipcc emits for `b == 0 ? NaN : a / b`.
"""
import sys

STRIDE = 512
H = {"CONST": 0, "VAR": 1, "ADD": 2, "SUB": 3, "MUL": 4, "DIV": 5, "END": 6}


def gen(name, depth, vla, base=64):
    B = [base + i for i in range(4)]
    D = [base + 4 + i for i in range(5)]
    NAN = base + 9
    T = [base + 10 + i for i in range(4)]
    V0 = base + 14
    S0 = V0 + 4 * vla
    top = S0 + 4 * depth  # first register above the block
    L = []
    a = L.append
    uid = "%="
    a("s_getpc_b64 s[38:39]")
    a(f".Lafter_{uid}:")
    a(f"s_add_u32 s38, s38, .Lbase_{uid}-.Lafter_{uid}")
    a("s_addc_u32 s39, s39, 0")
    for v in range(vla):
        a(f"ds_read_b128 v[{V0 + 4 * v}:{V0 + 4 * v + 3}], %[lds] offset:{1024 * v}")
    a(f"v_mov_b32 v{NAN}, 0x7fc00000")
    a("s_mov_b32 s36, 0")
    a("s_mov_b32 s37, 0")
    a("v_readlane_b32 s42, %[prog], s36")
    a("s_waitcnt lgkmcnt(0)")
    a("s_add_u32 s40, s38, s42")
    a("s_addc_u32 s41, s39, 0")
    a("s_setpc_b64 s[40:41]")
    a(f".p2align 9")
    a(f".Lbase_{uid}:")

    def prefetch(need_pay):
        if need_pay:
            a("v_readlane_b32 s43, %[pay], s36")
        a("s_add_u32 s36, s36, 1")
        a("v_readlane_b32 s42, %[prog], s36")

    def jump():
        a("s_add_u32 s40, s38, s42")
        a("s_addc_u32 s41, s39, 0")
        a("s_setpc_b64 s[40:41]")

    def spill_tos():
        a("s_set_gpr_idx_on s37, gpr_idx(DST)")
        for k in range(4):
            a(f"v_mov_b32 v{S0 + k}, v{T[k]}")
        a("s_set_gpr_idx_off")
        a("s_add_u32 s37, s37, 4")

    def pop_b():
        a("s_sub_u32 s37, s37, 4")
        a("s_set_gpr_idx_on s37, gpr_idx(SRC0)")
        for k in range(4):
            a(f"v_mov_b32 v{B[k]}, v{S0 + k}")
        a("s_set_gpr_idx_off")

    # 0 CONST
    prefetch(True)
    spill_tos()
    for k in range(4):
        a(f"v_mov_b32 v{T[k]}, s43")
    jump()
    # 1 VAR
    a(".p2align 9")
    prefetch(True)
    spill_tos()
    a("s_set_gpr_idx_on s43, gpr_idx(SRC0)")
    for k in range(4):
        a(f"v_mov_b32 v{T[k]}, v{V0 + k}")
    a("s_set_gpr_idx_off")
    jump()
    # 2..4 ADD SUB MUL  (a = top = left operand, b = popped = right operand)
    for op in ("v_add_f32", "v_sub_f32", "v_mul_f32"):
        a(".p2align 9")
        prefetch(False)
        pop_b()
        for k in range(4):
            a(f"{op} v{T[k]}, v{T[k]}, v{B[k]}")
        jump()
    # 5 DIV
    a(".p2align 9")
    prefetch(False)
    pop_b()
    d3, d4, d6, d7, d8 = D
    for k in range(4):
        x, y = T[k], B[k]
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
    jump()
    # 6 END
    a(".p2align 9")
    a(f"s_branch .Lend_{uid}")
    a(".p2align 9")
    a(f".Lend_{uid}:")
    for k in range(4):
        a(f"v_mov_b32 %[r{k}], v{T[k]}")
    body = "\n".join(f'    "{line}\\n\\t"' for line in L)
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(36, 46)] + [f'"v{i}"' for i in range(base, top)]
    clob_txt = ",\n      ".join(", ".join(clob[i:i + 12]) for i in range(0, len(clob), 12))
    out = f'''// GENERATED by gen/gen_interp_asm.py {name} (depth {depth}, {vla} variables, VGPRs v{base}..v{top - 1}) — do not edit.
#define EVOGP_ASM_TOP_{name.upper()} {top}
#define EVOGP_INTERP_ASM_{name.upper()}(r0_, r1_, r2_, r3_, prog_, pay_, lds_) \\
  asm volatile( \\
'''
    out += "\n".join(line + " \\" for line in body.split("\n"))
    out += f'''
    : [r0] "=v"(r0_), [r1] "=v"(r1_), [r2] "=v"(r2_), [r3] "=v"(r3_) \\
    : [prog] "v"(prog_), [pay] "v"(pay_), [lds] "v"(lds_) \\
    : {clob_txt.replace(chr(10), " " + chr(92) + chr(10))})
'''
    return out


if __name__ == "__main__":
    outdir = sys.argv[1] if len(sys.argv) > 1 else "."
    for name, depth, vla in (("d16", 16, 12), ("d12", 12, 10)):
        with open(f"{outdir}/interp_asm_{name}.inc", "w") as f:
            f.write(gen(name, depth, vla))
        print("wrote", f"{outdir}/interp_asm_{name}.inc")
