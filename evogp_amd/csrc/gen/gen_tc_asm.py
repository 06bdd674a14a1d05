#!/usr/bin/env python3
"""Generate tc_interp_k<K>.inc: the threaded-code SR-fitness interpreter for gfx950, v3.

What changed against the v2 core (gen_interp_asm.py) and why — all numbers from scripts/ubench/issue_model.hip
on MI355X (profiles/r01_issue_model_ubench.log):

  * one SIMD retires a wave64 fp32 VALU instruction every ~2.3 clocks but only ONE scalar instruction every
    4 clocks, a v_readlane costs 8-12 VALU clocks, and a single wave issues at most one instruction per
    ~4.3 clocks.  v2 spent ~11 SALU + 2 v_readlane per tree node for 4-8 VALU: scalar bound.
  * v3 therefore (1) interprets a COMPILED program: a separate kernel (sr_tc.hip: tc_compile_kernel) fuses every
    leaf into its parent operator, so the number of dispatches is the number of function nodes (12.6 instead
    of 26.3 per tree on configs[1]) and the operand stack only holds intermediate results;
    (2) fetches the program with scalar loads into a 64-SGPR window (16 instructions of 4 dwords:
    {handler address, -, operand a, operand b}); the dispatch is  s_movrels_b32 + s_setpc_b64  on an
    ABSOLUTE handler address — no v_readlane, no decode, no compare chain;
    (3) works on K = 8 rows per lane (one tree instruction = 8 VALU), so the scalar work per dispatch
    (5-7 SALU + 1 jump) hides behind the vector work of the other resident waves;
    (4) keeps the dataset in LDS, transposed so that a lane's rows of one variable are one ds_read_b128 per
    four rows: variable operands cost LDS bandwidth instead of VALU moves and the VGPR budget (128 = 4 waves
    per SIMD) goes to the operand stack;
    (5) lets every wave evaluate WHOLE trees (all datapoint tiles, one after the other): no barrier, no partial
    sums in LDS, no float atomics; trees are handed out in small batches from a global counter.

The block is ONE `asm volatile` statement that never returns (it ends the wave).  Register map (fixed):

  SGPR  s[16:17] jump target (lo from the program, hi constant)   s18 J = dword offset of the current instruction
        s19 H = K * stack height   s20 scatter M0 of the DIV body   s21 tile   s22 b (tree in batch)   s23 trees in batch
        s24 t0   s25 next t0   s[26:27] mask of evaluated trees   s[28:29] record address   s[30:31] operands a, b
        s32..s35 scratch   s[36:99] program window
  VGPR  v0 lane   v1 X base of the lane   v2 X base of the tile   v3 y address of the tile   v4,v5 scratch addresses
        v6 error accumulator   v7 batch results (lane b = tree b)   v8 NaN   v9 scratch   v[10:11] counter address
        v12 grabbed t0   v13 batch size   v14 store offset   v15 4*lane   v16..v20 division temporaries
        TA = v[24:24+K)  TB  Q  then the operand stack S0.. (slot e of row k = S0 + K*e + k)

VGPR indexing stays enabled while a program runs; handlers select the indexed operands by writing
M0 = (mode << 12) | index directly (s_add_u32 m0, H, imm), which replaces s_set_gpr_idx_on/off pairs.
"""
import sys

FORMS = ("SS", "SV", "VS", "SC", "CS", "VV", "VC", "CV")
OPS = ("add", "sub", "mul", "div")
SLOT = 256  # bytes per handler slot


def gen(K, DEPTH, stats=False):
    assert K % 4 == 0
    G = K // 4
    TA, TB, Q = 24, 24 + K, 24 + 2 * K
    S0 = 24 + 3 * K
    NV = S0 + K * DEPTH
    DT = [16, 17, 18, 19, 20]
    W = 36
    sPC, sJ, sH, sDST, sTILE, sB, sNB, sT0, sT0N, sOK, sREC, sA, sBop = 20, 22, 23, 24, 25, 26, 27, 28, 29, 30, 32, 34, 35
    T1, T2 = 100, 101
    T4 = sDST    # free outside the division stubs
    sBLK = sT0N  # the next grab is only live between two batches
    P1, P2, P3, P4 = 20, 21, 22, 23  # prologue scratch (control registers that are not live yet)
    uid = "%="
    L = []
    a = L.append

    def lab(n):
        return f".Ltc_{n}_{uid}"

    # cycle accounting (stats build only): v22 = ticks waiting for program records, v23 = ticks waiting for work,
    # v5 is not available (VV handlers), so the start tick of an interval is parked in the spare stack slot NV-1
    def tick_begin():
        if stats:
            a(f"s_memtime s[{T1}:{T2}]")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_mov_b32 v{NV - 1}, s{T1}")

    def tick_end(acc):
        if stats:
            a(f"s_memtime s[{T1}:{T2}]")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_sub_u32 v{NV - 1}, s{T1}, v{NV - 1}")
            a(f"v_add_u32 v{acc}, v{acc}, v{NV - 1}")

    hid = {}
    for o, op in enumerate(OPS):
        for f, form in enumerate(FORMS):
            hid[f"{op}_{form}"] = o * 8 + f
    hid["push_c"], hid["push_v"], hid["end"], hid["skip"], hid["next"] = 32, 33, 34, 35, 36
    NH = 37

    def epilogue():
        a(f"s_add_u32 s{sJ}, s{sJ}, 4")
        a(f"s_mov_b32 m0, s{sJ}")
        a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")

    def prefetch_pc():
        a(f"s_movrels_b32 s{sPC}, s{W + 4}")

    def load_var(bank, sreg, vaddr):
        a(f"v_add_u32 v{vaddr}, s{sreg}, v2")
        for g in range(G):
            a(f"ds_read_b128 v[{bank + 4 * g}:{bank + 4 * g + 3}], v{vaddr}" + (f" offset:{1024 * g}" if g else ""))

    # ------------------------------------------------------------------ prologue
    a("v_mbcnt_lo_u32_b32 v0, -1, 0")
    a("v_mbcnt_hi_u32_b32 v0, -1, v0")
    a("v_lshlrev_b32 v1, 4, v0")
    a("v_add_u32 v1, %[ldsx], v1")
    a("v_lshlrev_b32 v15, 2, v0")
    a("v_mov_b32 v8, 0x7fc00000")
    a("s_load_dwordx4 s[8:11], %[karg], 0x0")    # program records, fitness
    a(f"s_load_dwordx2 s[{P3}:{P4}], %[karg], 0x10")  # work counter
    a("s_load_dwordx8 s[12:19], %[karg], 0x28")  # pop, D, var_len, tiles, batch, flags, query, record stride
    a("s_waitcnt lgkmcnt(0)")
    a(f"v_mov_b32 v10, s{P3}")
    a(f"v_mov_b32 v11, s{P4}")
    a("v_mov_b32 v13, s16")
    a("s_mul_i32 s14, s14, s15")
    a(f"s_mul_i32 s14, s14, {G * 1024}")  # s14 = LDS distance from X to y
    a(f"s_load_dwordx4 s[{P1}:{P4}], %[karg], 0x48")  # static trees per workgroup, first dynamic tree, LDS offset of the queue head
    a("s_waitcnt lgkmcnt(0)")
    a(f"v_mov_b32 v21, s{P3}")                  # LDS address of the workgroup's queue head
    a("v_add_u32 v21, %[ldsx], v21")
    a(f"s_mul_i32 %[wgid], %[wgid], s{P1}")     # first tree of the workgroup's static share
    a(f"s_mov_b32 %[ldsx], s{P1}")              # from here on: size of the static share
    a(f"s_mov_b32 %[dyn], s{P2}")               # first tree of the dynamic region
    a(f"s_getpc_b64 s[{T1}:{T2}]")
    a(f"{lab('pc')}:")
    a(f"s_add_u32 s{T1}, s{T1}, {lab('hbase')}-{lab('pc')}")
    a(f"s_addc_u32 s{T2}, s{T2}, 0")
    a(f"s_mov_b32 s{sPC + 1}, s{T2}")
    # query mode: report the handler base address and leave
    a("s_cmp_eq_u32 s18, 0")
    a(f"s_cbranch_scc1 {lab('run')}")
    a(f"v_mov_b32 v4, s{T1}")
    a(f"v_mov_b32 v5, s{T2}")
    a("v_mov_b32 v9, 0")
    a("global_store_dwordx2 v9, v[4:5], s[10:11]")
    a("s_waitcnt vmcnt(0)")
    a("s_endpgm")
    a(f"{lab('run')}:")
    if stats:
        a("v_mov_b32 v22, 0")
        a("v_mov_b32 v23, 0")
        a(f"v_mov_b32 v{NV - 2}, 0")  # trees
        a(f"v_mov_b32 v{NV - 3}, 0")  # dispatches
        a(f"s_memtime s[{T1}:{T2}]")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_mov_b32 v{NV - 4}, s{T1}")  # start tick
    a("s_mov_b32 s18, 1")  # 1 while the workgroup's static share lasts
    # ------------------------------------------------------------------ batch loop
    # Work distribution: every workgroup owns a contiguous static share that its waves split through an LDS
    # counter (cheap); the rest of the population is handed out from one global counter (atomics on one
    # address serialise at ~11 ns each, so only the load-balancing tail goes through them).
    a(f"{lab('batch')}:")
    a("s_cmp_eq_u32 s18, 0")
    a(f"s_cbranch_scc1 {lab('dyn')}")
    tick_begin()
    a("s_mov_b64 exec, 1")
    a("ds_add_rtn_u32 v12, v21, v13")
    a("s_mov_b64 exec, -1")
    a("s_waitcnt lgkmcnt(0)")
    tick_end(23)
    a(f"v_readfirstlane_b32 s{sT0}, v12")
    a(f"s_cmp_lt_u32 s{sT0}, %[ldsx]")
    a(f"s_cbranch_scc0 {lab('to_dyn')}")
    a(f"s_sub_u32 s{sNB}, %[ldsx], s{sT0}")
    a(f"s_min_u32 s{sNB}, s{sNB}, s16")
    a(f"s_add_u32 s{sT0}, s{sT0}, %[wgid]")
    a(f"s_branch {lab('have_batch')}")
    a(f"{lab('to_dyn')}:")
    a("s_mov_b32 s18, 0")
    tick_begin()
    a("s_mov_b64 exec, 1")
    a("global_atomic_add v12, v[10:11], v13, off sc0")
    a("s_mov_b64 exec, -1")
    a("s_waitcnt vmcnt(0)")
    tick_end(23)
    a(f"v_readfirstlane_b32 s{sT0N}, v12")
    a(f"s_add_u32 s{sT0N}, s{sT0N}, %[dyn]")
    a(f"{lab('dyn')}:")
    a(f"s_mov_b32 s{sT0}, s{sT0N}")
    a(f"s_cmp_ge_u32 s{sT0}, s12")
    a(f"s_cbranch_scc1 {lab('exit')}")
    a("s_mov_b64 exec, 1")
    a("global_atomic_add v12, v[10:11], v13, off sc0")  # the next batch, consumed at the end of this one
    a("s_mov_b64 exec, -1")
    a(f"s_sub_u32 s{sNB}, s12, s{sT0}")
    a(f"s_min_u32 s{sNB}, s{sNB}, s16")
    a(f"{lab('have_batch')}:")
    a(f"s_mov_b32 s{sB}, 0")
    a(f"s_mov_b64 s[{sOK}:{sOK + 1}], 0")
    a("v_mov_b32 v7, 0")
    # ------------------------------------------------------------------ tree loop
    a(f"{lab('tree')}:")
    a(f"s_add_u32 s{T1}, s{sT0}, s{sB}")
    a(f"s_mul_hi_u32 s{sREC + 1}, s{T1}, s19")
    a(f"s_mul_i32 s{sREC}, s{T1}, s19")
    a(f"s_add_u32 s{sREC}, s{sREC}, s8")
    a(f"s_addc_u32 s{sREC + 1}, s{sREC + 1}, s9")
    tick_begin()
    for i in range(4):
        a(f"s_load_dwordx16 s[{W + 16 * i}:{W + 16 * i + 15}], s[{sREC}:{sREC + 1}], {hex(64 * i)}")
    a("v_mov_b32 v6, 0")
    a(f"s_mov_b32 s{sTILE}, 0")
    a(f"s_mov_b32 s{sBLK}, 0")
    a("s_waitcnt lgkmcnt(0)")
    tick_end(22)
    if stats:
        a(f"v_add_u32 v{NV - 2}, 1, v{NV - 2}")
    # ------------------------------------------------------------------ tile loop (one pass of the program)
    a(f"{lab('tile')}:")
    a(f"s_mul_i32 s{T1}, s{sTILE}, {G * 1024}")
    a(f"v_add_u32 v2, s{T1}, v1")
    a("v_add_u32 v3, s14, v2")
    a(f"s_mov_b32 s{sH}, 0")
    a(f"s_mov_b32 s{sJ}, 0")
    a(f"s_cmp_eq_u32 s{sBLK}, 0")
    a(f"s_cbranch_scc1 {lab('tile_go')}")
    for i in range(4):  # a program longer than one block: its first block has to come back
        a(f"s_load_dwordx16 s[{W + 16 * i}:{W + 16 * i + 15}], s[{sREC}:{sREC + 1}], {hex(64 * i)}")
    a(f"s_mov_b32 s{sBLK}, 0")
    a("s_waitcnt lgkmcnt(0)")
    a(f"{lab('tile_go')}:")
    a(f"s_set_gpr_idx_on s{sJ}, 0")  # J == 0: enables indexing with no operand selected, M0 = 0
    a(f"s_mov_b32 s{sPC}, s{W}")
    a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")

    # ------------------------------------------------------------------ handlers
    a(".p2align 8")
    a(f"{lab('hbase')}:")

    def begin(name):
        a(f".org {lab('hbase')}+{SLOT * hid[name]}")  # fails to assemble if the previous handler overflowed its slot
        a(f"{lab('h_' + name)}:")

    MODE = {"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}

    def m0_stack(mode_bits, off):
        """M0 = (mode << 12) + H + off   (off in registers, may be negative)"""
        imm = (mode_bits << 12) + off
        a(f"s_add_u32 m0, s{sH}, {hex(imm & 0xFFFFFFFF)}")

    def arith(op, form):
        ins = {"add": "v_add_f32", "sub": "v_sub_f32", "mul": "v_mul_f32"}[op]
        rev = {"add": "v_add_f32", "sub": "v_subrev_f32", "mul": "v_mul_f32"}[op]
        begin(f"{op}_{form}")
        prefetch_pc()
        if form == "SS":
            m0_stack(MODE["SRC0"] | MODE["SRC1"] | MODE["DST"], -2 * K)
            for k in range(K):
                a(f"{ins} v{S0 + k}, v{S0 + K + k}, v{S0 + k}")
            a(f"s_sub_u32 s{sH}, s{sH}, {K}")
        elif form == "SV":  # stack top (left) op variable
            a(f"s_movrels_b32 s{sBop}, s{W + 3}")
            load_var(TB, sBop, 4)
            m0_stack(MODE["SRC0"] | MODE["DST"], -K)
            a("s_waitcnt lgkmcnt(0)")
            for k in range(K):
                a(f"{ins} v{S0 + k}, v{S0 + k}, v{TB + k}")
        elif form == "VS":  # variable (left) op stack top
            a(f"s_movrels_b32 s{sA}, s{W + 2}")
            load_var(TA, sA, 4)
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            a("s_waitcnt lgkmcnt(0)")
            for k in range(K):
                a(f"{ins} v{S0 + k}, v{TA + k}, v{S0 + k}")
        elif form == "SC":  # stack top op constant
            a(f"s_movrels_b32 s{sBop}, s{W + 3}")
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            for k in range(K):
                a(f"{rev} v{S0 + k}, s{sBop}, v{S0 + k}")
        elif form == "CS":  # constant op stack top
            a(f"s_movrels_b32 s{sA}, s{W + 2}")
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            for k in range(K):
                a(f"{ins} v{S0 + k}, s{sA}, v{S0 + k}")
        elif form == "VV":
            a(f"s_movrels_b64 s[{sA}:{sBop}], s[{W + 2}:{W + 3}]")
            load_var(TA, sA, 4)
            load_var(TB, sBop, 5)
            m0_stack(MODE["DST"], 0)
            a("s_waitcnt lgkmcnt(0)")
            for k in range(K):
                a(f"{ins} v{S0 + k}, v{TA + k}, v{TB + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        elif form == "VC":  # variable op constant
            a(f"s_movrels_b64 s[{sA}:{sBop}], s[{W + 2}:{W + 3}]")
            load_var(TA, sA, 4)
            m0_stack(MODE["DST"], 0)
            a("s_waitcnt lgkmcnt(0)")
            for k in range(K):
                a(f"{rev} v{S0 + k}, s{sBop}, v{TA + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        elif form == "CV":  # constant op variable
            a(f"s_movrels_b64 s[{sA}:{sBop}], s[{W + 2}:{W + 3}]")
            load_var(TB, sBop, 4)
            m0_stack(MODE["DST"], 0)
            a("s_waitcnt lgkmcnt(0)")
            for k in range(K):
                a(f"{ins} v{S0 + k}, s{sA}, v{TB + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        epilogue()

    def div_stub(form):
        """gather a -> TA, b -> TB, set the scatter index of the result, adjust H, go to the shared body"""
        begin(f"div_{form}")
        prefetch_pc()
        la, rb = form[0], form[1]
        if form == "SS":
            m0_stack(MODE["SRC0"], -2 * K)
            for k in range(K):
                a(f"v_mov_b32 v{TA + k}, v{S0 + K + k}")
                a(f"v_mov_b32 v{TB + k}, v{S0 + k}")
            a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - 2 * K)}")
            a(f"s_sub_u32 s{sH}, s{sH}, {K}")
            a("s_mov_b32 m0, 0")
        else:
            need_a = la != "S"
            need_b = rb != "S"
            if need_a and need_b:
                a(f"s_movrels_b64 s[{sA}:{sBop}], s[{W + 2}:{W + 3}]")
            elif need_a:
                a(f"s_movrels_b32 s{sA}, s{W + 2}")
            else:
                a(f"s_movrels_b32 s{sBop}, s{W + 3}")
            if la == "V":
                load_var(TA, sA, 4)
            if rb == "V":
                load_var(TB, sBop, 5)
            if la == "S" or rb == "S":
                m0_stack(MODE["SRC0"], -K)
                bank = TA if la == "S" else TB
                for k in range(K):
                    a(f"v_mov_b32 v{bank + k}, v{S0 + k}")
                a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
            else:
                a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
                a(f"s_add_u32 s{sH}, s{sH}, {K}")
            a("s_mov_b32 m0, 0")
            if la == "C":
                for k in range(K):
                    a(f"v_mov_b32 v{TA + k}, s{sA}")
            if rb == "C":
                for k in range(K):
                    a(f"v_mov_b32 v{TB + k}, s{sBop}")
            if la == "V" or rb == "V":
                a("s_waitcnt lgkmcnt(0)")
        a(f"s_branch {lab('divbody')}")

    def div_rows(xs, ys, qs):
        """IEEE division rows: q = (y == 0) ? NaN : x / y  up to (not including) v_div_fixup (forward.cu:183-187)"""
        d3, d4, d6, d7, d8 = DT
        for x, y, q in zip(xs, ys, qs):
            a(f"v_cmp_neq_f32 vcc, 0, v{y}")
            a(f"v_cndmask_b32 v{x}, v8, v{x}, vcc")  # a NaN numerator makes the quotient NaN
            a(f"v_div_scale_f32 v{d3}, s[{T1}:{T2}], v{y}, v{y}, v{x}")
            a(f"v_rcp_f32 v{d4}, v{d3}")
            a(f"v_div_scale_f32 v{d6}, vcc, v{x}, v{y}, v{x}")
            a(f"v_fma_f32 v{d7}, -v{d3}, v{d4}, 1.0")
            a(f"v_fmac_f32 v{d4}, v{d7}, v{d4}")
            a(f"v_mul_f32 v{d7}, v{d6}, v{d4}")
            a(f"v_fma_f32 v{d8}, -v{d3}, v{d7}, v{d6}")
            a(f"v_fmac_f32 v{d7}, v{d8}, v{d4}")
            a(f"v_fma_f32 v{d3}, -v{d3}, v{d7}, v{d6}")
            a(f"v_div_fmas_f32 v{q}, v{d3}, v{d4}, v{d7}")

    for op in ("add", "sub", "mul"):
        for form in FORMS:
            arith(op, form)
    for form in FORMS:
        div_stub(form)

    # push constant (folded constant subtree, or a tree that is a single constant)
    begin("push_c")
    prefetch_pc()
    a(f"s_movrels_b32 s{sA}, s{W + 2}")
    m0_stack(MODE["DST"], 0)
    for k in range(K):
        a(f"v_mov_b32 v{S0 + k}, s{sA}")
    a(f"s_add_u32 s{sH}, s{sH}, {K}")
    epilogue()
    # push variable (a tree that is a single variable; H == 0)
    begin("push_v")
    prefetch_pc()
    a(f"s_movrels_b32 s{sA}, s{W + 2}")
    load_var(S0, sA, 4)
    a(f"s_add_u32 s{sH}, s{sH}, {K}")
    a("s_waitcnt lgkmcnt(0)")
    epilogue()

    # end of the program: fold this tile's errors into the accumulator
    begin("end")
    a(f"s_branch {lab('endbody')}")
    # a tree the compiler could not take: leave its (marked) fitness word alone
    begin("skip")
    a("s_set_gpr_idx_off")
    a(f"s_branch {lab('next_tree')}")
    # continuation: the program goes on in the next 256-byte block of the record
    begin("next")
    a(f"s_add_u32 s{sBLK}, s{sBLK}, 1")
    a(f"s_lshl_b32 s{T1}, s{sBLK}, 8")
    a(f"s_add_u32 s{T1}, s{sREC}, s{T1}")
    a(f"s_addc_u32 s{T2}, s{sREC + 1}, 0")
    for i in range(4):
        a(f"s_load_dwordx16 s[{W + 16 * i}:{W + 16 * i + 15}], s[{T1}:{T2}], {hex(64 * i)}")
    a(f"s_mov_b32 s{sJ}, 0")
    a("s_mov_b32 m0, 0")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_mov_b32 s{sPC}, s{W}")
    a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")
    a(f".org {lab('hbase')}+{SLOT * NH}")

    # end of the program: fold this tile's errors into the accumulator
    a(f"{lab('endbody')}:")
    a("s_set_gpr_idx_off")
    if stats:
        a(f"v_add_u32 v{NV - 3}, s{sJ}, v{NV - 3}")
    for g in range(G):
        a(f"ds_read_b128 v[{TB + 4 * g}:{TB + 4 * g + 3}], v3" + (f" offset:{1024 * g}" if g else ""))
    a(f"s_add_u32 s{T1}, s{sTILE}, 1")
    a(f"s_cmp_lt_u32 s{T1}, s15")
    a(f"s_cselect_b32 s{T2}, 0, s17")  # flag bit 1 (ragged) survives only on the last tile
    a(f"s_and_b32 s{T2}, s{T2}, 2")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_cmp_eq_u32 s{T2}, 0")
    a(f"s_cbranch_scc1 {lab('end_full')}")
    # ragged tile: rows >= D contribute nothing
    a(f"s_mul_i32 s{T2}, s{sTILE}, {256 * G}")
    for k in range(K):
        g, q = divmod(k, 4)
        a(f"s_add_u32 s{T4}, s{T2}, {g * 256 + q}")
        a(f"v_add_u32 v4, s{T4}, v15")
        a(f"v_sub_f32 v9, v{TB + k}, v{S0 + k}")
        a("v_cmp_gt_u32 vcc, s13, v4")
        a("s_bitcmp0_b32 s17, 0")
        a(f"s_cbranch_scc1 {lab(f'rag_abs{k}')}")
        a("v_mul_f32 v9, v9, v9")
        a(f"{lab(f'rag_abs{k}')}:")
        a("v_and_b32 v9, 0x7fffffff, v9")
        a("v_cndmask_b32 v9, 0, v9, vcc")
        a("v_add_f32 v6, v6, v9")
    a(f"s_branch {lab('end_acc')}")
    a(f"{lab('end_full')}:")
    a("s_bitcmp0_b32 s17, 0")
    a(f"s_cbranch_scc1 {lab('end_abs')}")
    for k in range(K):
        a(f"v_sub_f32 v9, v{TB + k}, v{S0 + k}")
        a("v_mul_f32 v9, v9, v9")
        a("v_add_f32 v6, v6, v9")
    a(f"s_branch {lab('end_acc')}")
    a(f"{lab('end_abs')}:")
    for k in range(K):
        a(f"v_sub_f32 v9, v{TB + k}, v{S0 + k}")
        a("v_add_f32_e64 v6, v6, |v9|")
    a(f"{lab('end_acc')}:")
    a(f"s_mov_b32 s{sTILE}, s{T1}")
    a(f"s_cmp_lt_u32 s{sTILE}, s15")
    a(f"s_cbranch_scc1 {lab('tile')}")
    # the tree is finished: fixed-order sum of the 64 lanes, lane b of v7 receives it
    for ctl in ("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0",
                "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0",
                "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"):
        a("s_nop 1")
        a(f"v_add_f32_dpp v6, v6, v6 {ctl}")
    a("s_nop 1")
    a(f"v_readlane_b32 s{T1}, v6, 63")
    a(f"s_mov_b32 m0, s{sB}")
    a(f"s_bitset1_b64 s[{sOK}:{sOK + 1}], s{sB}")
    a(f"v_writelane_b32 v7, s{T1}, m0")
    a(f"{lab('next_tree')}:")
    a(f"s_add_u32 s{sB}, s{sB}, 1")
    a(f"s_cmp_lt_u32 s{sB}, s{sNB}")
    a(f"s_cbranch_scc1 {lab('tree')}")
    # batch finished: take the prefetched grab first (the store below then never sits in front of a wait),
    # then mean = sum / D and one coalesced store for the evaluated trees
    a("s_cmp_eq_u32 s18, 0")
    a(f"s_cbranch_scc0 {lab('no_grab')}")
    tick_begin()
    a("s_waitcnt vmcnt(0)")
    tick_end(23)
    a(f"v_readfirstlane_b32 s{sT0N}, v12")
    a(f"s_add_u32 s{sT0N}, s{sT0N}, %[dyn]")
    a(f"{lab('no_grab')}:")
    a(f"s_cmp_eq_u64 s[{sOK}:{sOK + 1}], 0")
    a(f"s_cbranch_scc1 {lab('batch')}")
    a(f"s_mov_b64 exec, s[{sOK}:{sOK + 1}]")
    a("v_cvt_f32_u32 v9, s13")
    d3, d4, d6, d7, d8 = DT
    a(f"v_div_scale_f32 v{d3}, s[{T1}:{T2}], v9, v9, v7")
    a(f"v_rcp_f32 v{d4}, v{d3}")
    a(f"v_div_scale_f32 v{d6}, vcc, v7, v9, v7")
    a(f"v_fma_f32 v{d7}, -v{d3}, v{d4}, 1.0")
    a(f"v_fmac_f32 v{d4}, v{d7}, v{d4}")
    a(f"v_mul_f32 v{d7}, v{d6}, v{d4}")
    a(f"v_fma_f32 v{d8}, -v{d3}, v{d7}, v{d6}")
    a(f"v_fmac_f32 v{d7}, v{d8}, v{d4}")
    a(f"v_fma_f32 v{d3}, -v{d3}, v{d7}, v{d6}")
    a(f"v_div_fmas_f32 v{d3}, v{d3}, v{d4}, v{d7}")
    a(f"v_div_fixup_f32 v{d3}, v{d3}, v9, v7")
    a(f"v_add_u32 v14, s{sT0}, v0")
    a("v_lshlrev_b32 v14, 2, v14")
    a(f"global_store_dword v14, v{d3}, s[10:11]")
    a("s_mov_b64 exec, -1")
    a(f"s_branch {lab('batch')}")

    # shared division body: K rows, then the scatter through v_div_fixup with an indexed destination
    a(f"{lab('divbody')}:")
    div_rows([TA + k for k in range(K)], [TB + k for k in range(K)], [Q + k for k in range(K)])
    a(f"s_mov_b32 m0, s{sDST}")
    for k in range(K):
        a(f"v_div_fixup_f32 v{S0 + k}, v{Q + k}, v{TB + k}, v{TA + k}")
    epilogue()

    a(f"{lab('exit')}:")
    if stats:  # {record wait, work wait, trees, 4 * dispatches, wave ticks, waves} += this wave's counters
        a(f"s_memtime s[{T1}:{T2}]")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_sub_u32 v{NV - 4}, s{T1}, v{NV - 4}")
        a(f"s_load_dwordx2 s[{T1}:{T2}], %[karg], 0x58")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_mov_b32 v10, s{T1}")
        a(f"v_mov_b32 v11, s{T2}")
        a("v_mov_b32 v13, 0")
        a("s_mov_b64 exec, 1")
        for i, src in enumerate((22, 23, NV - 2, NV - 3, NV - 4, None)):
            if src is None:
                a("v_mov_b32 v12, 1")
            else:
                a(f"v_mov_b32 v12, v{src}")
            a(f"global_atomic_add_x2 v[10:11], v[12:13], off offset:{8 * i}")
        a("s_waitcnt vmcnt(0)")
    a("s_endpgm")

    body = "\n".join(f'    "{line}\\n\\t"' for line in L)
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(8, 102)] + [f'"v{i}"' for i in range(0, NV)]
    clob_txt = ", ".join(clob)
    name = f"K{K}" + ("S" if stats else "")
    out = f"// GENERATED by gen/gen_tc_asm.py (K = {K} rows per lane, {DEPTH}-entry operand stack, VGPRs v0..v{NV - 1}) — do not edit.\n"
    out += f"#define EVOGP_TC_{name}_DEPTH {DEPTH}\n#define EVOGP_TC_{name}_VGPRS {NV}\n"
    if K == 8 and not stats:
        out += f"#define EVOGP_TC_SLOT {SLOT}\n#define EVOGP_TC_NHANDLERS {NH}\n"
        for n, i in sorted(hid.items(), key=lambda kv: kv[1]):
            out += f"#define EVOGP_TC_H_{n.upper()} {i}\n"
    out += f"#define EVOGP_TC_ASM_{name}(karg_, ldsx_, wgid_, dyn_) \\\n  asm volatile( \\\n"
    out += "\n".join(line + " \\" for line in body.split("\n"))
    out += f'''
    : [ldsx] "+s"(ldsx_), [wgid] "+s"(wgid_), [dyn] "+s"(dyn_) \\
    : [karg] "s"(karg_) \\
    : {clob_txt})
'''
    return out


if __name__ == "__main__":
    outdir = sys.argv[1] if len(sys.argv) > 1 else "."
    for K, depth in ((8, 10), (4, 15)):
        with open(f"{outdir}/tc_interp_k{K}.inc", "w") as f:
            f.write(gen(K, depth))
            if K == 8:
                f.write(gen(K, depth, stats=True))  # cycle-accounting build (its top stack slot holds the counters)
        print("wrote", f"{outdir}/tc_interp_k{K}.inc")
