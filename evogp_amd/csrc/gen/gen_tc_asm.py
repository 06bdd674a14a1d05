#!/usr/bin/env python3
"""Generate tc_interp_k<K>.inc: the threaded-code SR-fitness interpreter for gfx950.

Why it looks the way it does — measured on MI355X (scripts/ubench/*.hip, profiles/*ubench*.log; PMC counters in profiles/;
DESIGN.md section 3.1 and docs/DESIGN_history_r01_r03.md have the numbers):

  * a tree interpreter on this chip ends up bound by VALU issue (the division's dependent operations above all) once the
    per-instruction overheads are gone: a computed jump costs a wave ~50 clocks, a v_readlane 8-12 VALU clocks, one
    scalar instruction issues per ~4 clocks and SIMD, a cold 256-byte scalar load takes >1000 clocks.  So: as few
    dispatches as possible, as much vector work per dispatch as the registers allow, no exposed waits;
  * it interprets a COMPILED program: tc_compile_kernel (sr_tc.hip) fuses every leaf into its parent operator, so there
    is one dispatch per FUNCTION node (12.9 instead of 26.3 per tree on configs[1]) and the operand stack only holds
    intermediate results;
  * a program is 8-byte words {(LDS offset of the instruction's variable operand / 1024) << 24 | aux << 16 | offset of its
    handler in the 64-KiB-aligned handler table, constant or second variable}; a 256-byte record (32 words) is one fill of
    the 64-SGPR window (four s_load_dwordx16); dispatch = s_movrels_b32 (next word) + s_pack_lh_b32_b16 (handler address) +
    s_setpc_b64 — no v_readlane, no decode, no compare chain.  Longer programs chain blocks: word 31 is then NEXT (refill
    from the tree's record in the next array of records);
  * K rows per lane (8; 4 and 1 for small datasets): one tree instruction = K VALU (a division ~8 K, sin / cos ~25 K);
  * handlers: + - * / in the eight operand forms {S stack, V variable, C constant}^2 minus CC (folded by the compiler);
    unary neg abs sin cos tan sqrt loose-sqrt exp log loose-log in the forms S (in place) and V (push); the
    transcendental ones are the device math library's instruction sequences (taken from hipcc's output for sinf, ...),
    sin / cos / tan only their small-argument path: a block with an operand of 2^17 or more BAILS OUT at run time (the
    tree gets the register kernels' sentinel and their pending flag is raised);
  * the rarer functions share GENERIC STUBS (eight binary forms, two unary ones) that gather the operands into fixed banks
    and jump to the body the word's aux field names: loose division, max min, < > <= >=, and — run row by row through the
    library's transcribed sequences (gen/ocml_transcribe.py -> ocml_bodies.py, 120-190 instructions each) — pow, loose pow,
    sinh, cosh; tanh inline; IF with its three operands on the stack;
  * multi-output programs (forward.cu:237-243): the first out_len stack entries are the output accumulators; mo_begin clears
    them, acc_s adds the top of the stack to one of them, end_mo folds the errors of all outputs; end_cls instead takes the
    arg-max over them per row and counts the rows whose arg-max is their class label (the Classification problem's fitness);
  * divisions of S / S, S / c, c / S have in-place handlers (operands read where they are -- every operand position M0-relative --,
    temporaries in the registers above the stack top), used where the compiler finds that entry free;
  * the division comes in three selectable row sequences (ieee / short / fast, evogp_hip_set_sr_division); the
    reference's "b == 0 -> NaN" is tested once per K x 64 block (min |b|), not per row; under short / fast a block whose
    operands all lie in [2^-46, 2^46] skips the range scaling (rcp, mul, two packed fmas per row pair, the quotient written in
    place), a numerator or denominator that is +-0 in the whole block is one instruction per row, and dataset variables whose
    whole column is in range (flags bits 17-30, set by sr_tc_kernel's prologue) need no test (DIVRANGE / TRUST below);
  * round 5: a division BY A VARIABLE reads the variable's reciprocal from a second copy of the dataset's columns in LDS (divr_*: the
    trusted-variable division without its v_rcp_f32; flags bit 11, aux = KiB index of the reciprocal column); the hot handlers have TWINS
    that do not prefetch (NOPF_TWINS: the compiler names them where the next word has no variable operand); end_cls judges segments of
    lanes against a scalar label (the launch stages the rows in label order); a wide-stack build of the 8-row variant (13 entries);
    pow / sinh / cosh / exp / log run over row pairs (gen/pair_rows.py packs the transcribed bodies and proves the packed body equal);
  * constants enter + - * and PUSH_C through packed instructions (one SGPR source for two rows per issue slot, PKCONST);
  * the dataset lives in LDS, transposed so that a lane's rows of one variable are one ds_read_b128 per four rows.
    Variable operands are PREFETCHED one instruction ahead: every handler starts by issuing the LDS reads for the NEXT
    instruction's variable into the other of two operand banks.  Handlers therefore come in two flavours (current bank
    0 / 1), chosen by the compiler from the parity of the instruction index;
  * every wave evaluates WHOLE trees (all datapoint tiles, one after the other): no barrier, no partial sums in LDS, no
    float atomics.  Work distribution: each workgroup owns a static share that its waves walk round-robin in batches (the
    next batch is known: its records are pulled into L2 / the scalar cache while the current one runs); the
    load-balancing tail is one dynamic region and one counter line per XCD.

The block is ONE `asm volatile` statement that never returns (it ends the wave).  Register map (fixed):

  SGPR  s[8:9] records  s[10:11] fitness  s12 pop, then end of this XCD's dynamic region  s13 D  s14 LDS distance X->y  s15 tiles  s16 batch  s17 flags (bits 5-7, 17-30: trusted variables)
        s18 static phase  s19 record stride  s[20:21] jump target  s22 J = dword offset of the current instruction
        s23 H = K * stack height  s24 scatter M0 of the division / scratch  s25 tile  s26 b (tree in batch)
        s27 trees in batch  s28 t0  s29 next dynamic t0 / program block  s[30:31] mask of evaluated trees
        s[32:33] record address  s[34:35] operands a, b  s[36:99] program window  s[100:101] scratch
        in/out operands of the statement: static cursor, static end, static stride, first dynamic tree, prefetch offset
  VGPR  v0 lane  v1 X base of the lane  v2 X base of the tile  v3 y address of the tile  v4,v5 addresses
        v6 error accumulator  v7 batch results (lane b = tree b)  v8 NaN  v9 scratch  v[10:11] counter address
        v12 grabbed t0  v13 warm-up offset  v[14:17] warm-up sink  v18..v23 division temporaries
        P0 P1 (variable-operand banks)  T (second operand / labels)  Q (quotients)  then the operand stack
        (slot e of row k = S0 + K*e + k)

VGPR indexing stays enabled while a program runs; handlers select the indexed operands by writing
M0 = (mode << 12) | index directly (s_add_u32 m0, H, imm), which replaces s_set_gpr_idx_on/off pairs.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ocml_bodies import BODIES  # noqa: E402  (the library's sequences for pow / sinh / cosh, gen/ocml_transcribe.py)
from pair_rows import Pairing, check as check_pairing  # noqa: E402

FORMS = ("SS", "SV", "VS", "SC", "CS", "VV", "VC", "CV")
SENTINEL_HEAVY = 0x7FC0FEED  # sr_params.hpp kSentinelHeavy: "evaluate me in the FULL register kernel"
OPS = ("add", "sub", "mul", "div")
UNARY = ("neg", "abs", "sin", "cos", "tan", "sqrt", "lsqrt", "exp", "log", "llog")  # unary handlers, in the compiler kernel's numbering (sr_tc.hip)
GBIN = ("ldiv", "max", "min", "lt", "gt", "le", "ge", "pow", "lpow")  # bodies behind the generic binary stubs, selected by the word's aux field
GUN = ("zero", "sinh", "cosh", "tanh", "one", "rcp")                  # bodies behind the generic unary stubs (sr_tc.hip GU_*)
HEAVY_REGS = 24  # VGPRs the transcribed library sequences borrow from the top of the operand stack (the compiler keeps it free)
SLOT = 256  # bytes per handler slot
DIVIP_REGS = 12  # VGPRs an in-place division uses above its operands (the compiler reserves ceil(DIVIP_REGS / K) stack entries)
DIVIP = ("SS", "SC", "CS")  # division forms with an in-place handler (operands read where they are, temporaries above the stack)
DIVR = ("SV", "VV", "CV")   # divisions by a variable through the launch's reciprocal columns (divr_*: the product with 1 / v)
# Handlers with a twin that does NOT prefetch (name + "_np"): the compiler names the twin where the NEXT word has no variable operand (about
# half of the headline's words) -- four instructions less of the fifteen of an addition: the shift of the next word's offset, the
# address, two LDS reads
NOPF_TWINS = tuple(f"{op}_{form}" for op in ("add", "sub", "mul") for form in FORMS) + ("push_c", "push_v") \
    + tuple(f"divip_{f}" for f in DIVIP) + tuple(f"divr_{f}" for f in DIVR)
NHF = 37 + 2 * len(UNARY) + 8 + 2 + 4 + len(DIVIP) + 1 + 1 + len(DIVR) + len(NOPF_TWINS)  # handlers per flavour: ... + generic binary forms + generic unary S/V + if, acc, mo_begin, end_mo + in-place divisions + end_cls + swap + divisions by reciprocal columns


NOPF = False  # EVOGP_TC_GEN_NOPF=1: drop the operand prefetch (timing experiment, wrong results)
SPLAT = False    # EVOGP_TC_GEN_SPLAT=1: copy a constant operand into a VGPR before the row loop (experiment: 1.5 % SLOWER at 1 M trees)
FMA_LOSS = False  # EVOGP_TC_GEN_FMA_LOSS=1: accumulate squared errors with one fused multiply-add (timing experiment)
PKARITH = True   # register-register + - *, moves and the division's quotient estimates as packed instructions over row pairs (EVOGP_TC_GEN_PKARITH=0: VOP2)
DIVABREAST = True  # the range-tested division rows run two row pairs abreast (twelve temporaries; EVOGP_TC_GEN_DIVABREAST=0: one pair at a time)
PKCONST = True   # + - * with a constant operand, and the push of a constant, as packed instructions over row pairs (EVOGP_TC_GEN_PKCONST=0: off)
TRUST = True     # divisions by / of a dataset variable whose whole column is in range skip the range test (EVOGP_TC_GEN_TRUST=0: off)
DIVFIX = False   # the range-tested rows end in v_div_fixup (EVOGP_TC_GEN_DIVFIX=1: an experiment; it changes nothing but NaN payloads)
DIVRANGE = True  # short division: blocks whose operands all lie in [2^-46, 2^46] take rows without range scaling, residuals as v_pk_fma over row pairs (EVOGP_TC_GEN_DIVRANGE=0: off)
DIV_LO, DIV_HI = 0x28800000, 0x56800000  # 2^-46, 2^46: v_div_scale leaves such operands alone (|exponent difference| < 96, no denormal in sight)
EARLYREC = True  # END of a tree's last tile sends for the record of the batch's next tree (EVOGP_TC_GEN_EARLYREC=0: the tree loop does)
L2WARM = True    # vector loads that pull the next batch's records into L2 (EVOGP_TC_GEN_L2WARM=0: off)
KWARM_LINES = 4  # 64-byte lines of the next record the warm-up touches (EVOGP_TC_GEN_KWARM_LINES)
FUSED_SIZE_OFF, FUSED_LEN_OFF = 96 + 16, 96 + 60   # FusedParams (sr_tc.hip): c.size and c.gp_len behind the 96 bytes of TcParams (static_asserts there)
TOUCH = True     # fused build: the block's entry touches the rows of the wave's next batch (EVOGP_TC_GEN_TOUCH=0: off)
TRIGPK = True    # sin / cos / tan over row pairs with packed multiplications and fused multiply-adds (EVOGP_TC_GEN_TRIGPK=0: row by row)
LIBPK = True     # pow / sinh / cosh: the library's sequences over row PAIRS (gen/pair_rows.py; EVOGP_TC_GEN_LIBPK=0: row by row)
RECGLC = False   # fused build: the record loads carry glc (EVOGP_TC_GEN_RECGLC=1) instead of one s_dcache_inv per batch
CODEWARM = False  # EVOGP_TC_GEN_CODEWARM=1: every wave pulls the handler table and the bodies behind it into its XCD's L2 before it starts (experiment: 4-5 us SLOWER at 250 k - 1 M trees, no change at 125 k: profiles/r05Z_codewarm_ab.log)
CNDE64 = False  # EVOGP_TC_GEN_CNDE64=1: every VOP2 v_cndmask_b32 ..., vcc in its 64-bit encoding (scripts/ubench/valu_rates.hip: the VOP2 form costs a SIMD ~4 issue slots of the VOP3 form)
KWARM = False  # scalar-cache warm-up of the next record (EVOGP_TC_GEN_KWARM=1 at generation time enables it): +1.5 % in round 2, -0.5 % since the division was rebuilt (profiles/r03E_div_range_ab.log)


def count_path(L, start, taken, idx_on=True):
    """Instruction counts along the usual path of one handler: from its label to the jump to the next handler, following
    unconditional branches; of the conditional ones only those to a label in `taken` (END: a full tile, then the next tile
    of the same tree).  bench.py multiplies the counts by the handler histogram of a population
    (evogp_hip_debug_tc_histogram) to state how much of the VALU issue rate the interpreter uses."""
    idx = {line[:-1]: i for i, line in enumerate(L) if line.endswith(":")}
    c = {"valu": 0, "trans": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "valu_clk": 0}
    i, steps = idx[start], 0
    while steps < 20000:
        steps += 1
        line = L[i]
        i += 1
        if line.endswith(":") or line.startswith("."):
            continue
        w = line.split()
        op = w[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if op.split("_")[1] in ("rcp", "sqrt", "exp", "log", "rsq", "sin", "cos"):
                c["trans"] += 1
                c["valu_clk"] += 8
            else:
                # issue clocks by class (scripts/ubench/valu_rates.hip, kernel wall time per 64-lane instruction): a VOP2 add / sub /
                # mul / mov on VGPR sources 2 -- but only while VGPR indexing is off (k_add_idx_* in the same table: 4 with it on, and it
                # is on from a program's first instruction to its END) --, everything else (VOP3, three sources, an SGPR or literal
                # source, packed) 4, transcendentals 8
                srcs = " ".join(w[2:]) if len(w) > 2 else ""
                fast = not idx_on and op in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32") and not any(
                    t in srcs for t in ("s", "0x", "|", "%"))
                c["valu_clk"] += 2 if fast else 4
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_"):
            c["vmem"] += 1
        elif op.startswith("s_load") or op.startswith("s_memtime"):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op in ("s_setpc_b64", "s_endpgm"):
                break
            if op == "s_branch" or (op.startswith("s_cbranch") and w[1] in taken):
                i = idx[w[1]]
    return c


def gen(K, DEPTH, stats=False, fast=0, info=None, fused=False, wide=False):
    assert K % 4 == 0 or K == 1
    assert 24 + 4 * K + K * DEPTH <= 256
    G = max(K // 4, 1)   # 1-KiB groups of a variable's tile: 64 lanes x 16 bytes (K = 1 uses the first row of every lane's four)
    RPL = min(K, 4)      # rows per lane and group
    P = [24, 24 + K]
    T, Q = 24 + 2 * K, 24 + 3 * K
    S0 = 24 + 4 * K
    NV = S0 + K * DEPTH
    DT = [18, 19, 20, 21, 22]
    W = 36
    sPC, sJ, sH, sDST, sTILE, sB, sNB, sT0, sT0N, sOK, sREC, sA, sX = 20, 22, 23, 24, 25, 26, 27, 28, 29, 30, 32, 34, 35
    sBop = sA    # an instruction carries ONE 32-bit operand (a constant, or the 16-byte-unit LDS offset of its second variable)
    T1, T2 = 100, 101
    T4 = sDST    # free outside the division stubs
    P1_, P2_, P3_, P4_ = 20, 21, 22, 23  # prologue scratch (control registers that are not live yet)
    CUR, END_, STRIDE, DYN, PF, BASE = "%[cur]", "%[lim]", "%[stride]", "%[dyn]", "%[pf]", "%[base]"
    if fused:
        # FUSED build (sr_fused_kernel, sr_tc.hip): the statement runs ONE batch of compiled trees and RETURNS.  The wave's C++ loop
        # owns the work distribution and compiles the batch into the wave's own ring of records (hot in L2) right before it
        # enters; nothing here touches s12 / s16 / s18 / s19 / s29 as batch bookkeeping or v12-v17 (the compiler keeps what it
        # needs across the statement there), so the prefetch scratch and the table base live in two of those scalars.
        assert not stats
        PF, BASE = "s16", "s18"
    GLC = " glc" if (fused and RECGLC) else ""

    def rec_addr(lo, hi, index, tmp):
        """address of the first block of the record of tree `index` (an SGPR name: tree number, or slot of the wave's ring)"""
        if fused:
            a(f"s_lshl_b32 s{tmp}, {index}, 8")
            a(f"s_add_u32 s{lo}, s8, s{tmp}")
            a(f"s_addc_u32 s{hi}, s9, 0")
        else:
            a(f"s_mul_hi_u32 s{hi}, {index}, s19")
            a(f"s_mul_i32 s{lo}, {index}, s19")
            a(f"s_add_u32 s{lo}, s{lo}, s8")
            a(f"s_addc_u32 s{hi}, s{hi}, s9")

    def load_window():
        for i in range(4):
            a(f"s_load_dwordx16 s[{W + 16 * i}:{W + 16 * i + 15}], s[{sREC}:{sREC + 1}], {hex(64 * i)}{GLC}")
    uid = "%="
    L = []
    sect = [L]   # the list instructions are appended to (the program's tail is generated late but placed early, see below)

    def a(line):
        sect[0].append(line)

    def lab(n):
        return f".Ltc_{n}_{uid}"

    hid = {}
    for o, op in enumerate(OPS):
        for f, form in enumerate(FORMS):
            hid[f"{op}_{form}"] = o * 8 + f
    hid["push_c"], hid["push_v"], hid["end"], hid["skip"] = 32, 33, 34, 35
    for u, uop in enumerate(UNARY):
        hid[f"{uop}_S"], hid[f"{uop}_V"] = 36 + 2 * u, 37 + 2 * u
    hid["next"] = 36 + 2 * len(UNARY)
    nh = hid["next"] + 1
    for f, form in enumerate(FORMS):
        hid[f"gbin_{form}"] = nh + f
    hid["gun_S"], hid["gun_V"], hid["if_sss"], hid["acc_s"], hid["mo_begin"], hid["end_mo"] = (nh + 8 + i for i in range(6))
    for i, form in enumerate(DIVIP):
        hid[f"divip_{form}"] = nh + 14 + i
    hid["end_cls"] = nh + 14 + len(DIVIP)
    hid["swap"] = nh + 14 + len(DIVIP) + 1
    for i, form in enumerate(DIVR):
        hid[f"divr_{form}"] = nh + 14 + len(DIVIP) + 2 + i
    for i, name in enumerate(NOPF_TWINS):
        hid[name + "_np"] = nh + 14 + len(DIVIP) + 2 + len(DIVR) + i
    assert NHF == nh + 14 + len(DIVIP) + 2 + len(DIVR) + len(NOPF_TWINS) and 2 * NHF * SLOT <= 65536
    twin = [False]   # True while a no-prefetch twin is being generated (begin, entry, prefetch, wait_cur look at it)

    # cycle accounting (stats build only); counters live in the top operand-stack slot
    A_REC, A_WORK, A_TREES, A_DISP, A_START, A_TICK = NV - 1, NV - 2, NV - 3, NV - 4, NV - 5, NV - 6

    def tick_begin():
        if stats:
            a(f"s_memtime s[{T1}:{T2}]")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_mov_b32 v{A_TICK}, s{T1}")

    def tick_end(acc):
        if stats:
            a(f"s_memtime s[{T1}:{T2}]")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_sub_u32 v{A_TICK}, s{T1}, v{A_TICK}")
            a(f"v_add_u32 v{acc}, v{acc}, v{A_TICK}")

    def epilogue():
        a(f"s_add_u32 s{sJ}, s{sJ}, 2")
        a(f"s_mov_b32 m0, s{sJ}")
        a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")

    def read_bank(bank, vaddr):
        if K == 1:
            a(f"ds_read_b32 v{bank}, v{vaddr}")
            return
        for g in range(G):
            a(f"ds_read_b128 v[{bank + 4 * g}:{bank + 4 * g + 3}], v{vaddr}" + (f" offset:{1024 * g}" if g else ""))

    def entry():
        """common head of a handler: address of the next handler and LDS offset of the next instruction's variable"""
        a(f"s_movrels_b32 s{sX}, s{W + 2}")                  # next word: {LDS offset / 1024 of its variable, aux, handler offset}
        a(f"s_pack_lh_b32_b16 s{sPC}, s{sX}, {BASE}")        # handler table is 64 KiB aligned: address = {base.hi16, offset}
        if not twin[0]:
            a(f"s_lshr_b32 {PF}, s{sX}, 24")                # (in 1-KiB units: the shift rides in prefetch's v_lshl_add)

    def read_aux(dst):
        """aux field (bits 23:16) of the CURRENT instruction's word; M0 must still be J (i.e. before any m0_stack)"""
        a(f"s_movrels_b32 s{dst}, s{W}")
        a(f"s_bfe_u32 s{dst}, s{dst}, 0x80010")

    def prefetch(nxt):
        if NOPF or twin[0]:  # (NOPF: timing experiment only, wrong results)
            return
        a(f"v_lshl_add_u32 v4, {PF}, 10, v2")
        read_bank(nxt, 4)

    def warm(sreg):
        """pull the eight records starting at tree `sreg` into L2 (results discarded)"""
        if not L2WARM:
            return
        a(f"s_mul_hi_u32 s{T2}, {sreg}, s19")
        a(f"s_mul_i32 s{T1}, {sreg}, s19")
        a(f"s_add_u32 s{T1}, s{T1}, s8")
        a(f"s_addc_u32 s{T2}, s{T2}, s9")
        a(f"global_load_dwordx4 v[14:17], v13, s[{T1}:{T2}]")
        a(f"global_load_dwordx4 v[14:17], v13, s[{T1}:{T2}] offset:1024")

    def dyn_batch(dst):
        """trees per DYNAMIC batch = batch >> flags[9:8] (at least 1): the tail is handed out in smaller pieces"""
        a(f"s_bfe_u32 s{T2}, s17, 0x20008")
        a(f"s_lshr_b32 s{dst}, s16, s{T2}")
        a(f"s_max_u32 s{dst}, s{dst}, 1")

    def grab():
        dyn_batch(T1)
        a("s_mov_b64 exec, 1")
        a(f"v_mov_b32 v9, s{T1}")
        a("global_atomic_add v12, v[10:11], v9, off sc0")
        a("s_mov_b64 exec, -1")

    def pool_grab():
        """the next batch of the WORKGROUP's pool: a counter in LDS (STRIDE holds its address), a few hundred clocks, no traffic beyond
        the CU.  LDS operations complete in order, so the handlers' counted waits (s_waitcnt lgkmcnt(n)) stay right with this one in
        flight ahead of their reads."""
        dyn_batch(T1)
        a("s_mov_b64 exec, 1")
        a(f"v_mov_b32 v9, s{T1}")
        a(f"v_mov_b32 v4, {STRIDE}")
        a("ds_add_rtn_u32 v12, v4, v9")
        a("s_mov_b64 exec, -1")

    # ------------------------------------------------------------------ prologue
    if fused:
        # entry of a batch: the records were written by this wave's vector stores a moment ago -- wait for them, and drop what the
        # scalar cache may still hold of the ring's previous contents
        a("s_waitcnt vmcnt(0)")
        if not RECGLC:
            a("s_dcache_inv")
        if TOUCH:
            # one line per lane of the rows of the wave's NEXT batch: they arrive in L2 while this batch is interpreted (v12 is the
            # sink; the batch's end waits for it with everything else)
            a("global_load_ushort v12, %[taddr], off")
        a("v_mbcnt_lo_u32_b32 v0, -1, 0")
        a("v_mbcnt_hi_u32_b32 v0, -1, v0")
        a("v_lshlrev_b32 v1, 4, v0")
        a("v_add_u32 v1, %[ldsx], v1")
        a("v_mov_b32 v8, 0x7fc00000")
        a("s_load_dwordx4 s[8:11], %[karg], 0x0")    # base of the record rings, fitness
        a(f"s_load_dwordx2 s[{P3_}:{P4_}], %[karg], 0x10")  # word 0 of the call's scratch block (the register kernels' pending flag)
        a("s_load_dwordx8 s[12:19], %[karg], 0x28")  # pop, D, var_len, tiles, batch, flags, -, -
        a("s_waitcnt lgkmcnt(0)")
        a("s_or_b32 s17, s17, %[trust]")              # the trusted-variable bits (17-30, and 5-7)
        # the lengths of the NEXT batch's trees (size[t0n + lane][0], lanes below its tree count) for the wave's compiler: on their way
        # while this batch runs (FUSED_SIZE_OFF / FUSED_LEN_OFF: where the kernel's argument block keeps the size array and gp_len)
        a(f"s_load_dwordx2 s[{P1_}:{P2_}], %[karg], {hex(FUSED_SIZE_OFF)}")
        a(f"s_load_dword s{T1}, %[karg], {hex(FUSED_LEN_OFF)}")
        a(f"s_bfe_u32 s{T2}, %[nb], 0x80008")
        a("v_add_u32 v4, %[t0n], v0")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_mul_lo_u32 v4, v4, s{T1}")
        a("v_lshlrev_b32 v4, 1, v4")
        a(f"s_bfm_b64 exec, s{T2}, 0")
        a(f"s_cbranch_execz {lab('no_lens')}")
        a(f"global_load_sshort %[lensn], v4, s[{P1_}:{P2_}]")
        a(f"{lab('no_lens')}:")
        a("s_mov_b64 exec, -1")
        a(f"v_mov_b32 v10, s{P3_}")
        a(f"v_mov_b32 v11, s{P4_}")
        a("s_mul_i32 s14, s14, s15")
        a(f"s_mul_i32 s14, s14, {G * 1024}")          # s14 = LDS distance from X to y
        a("s_add_u32 s8, s8, %[roff]")                # this wave's ring of records
        a("s_addc_u32 s9, s9, 0")
        a(f"s_getpc_b64 s[{T1}:{T2}]")
        a(f"{lab('pc')}:")
        a(f"s_add_u32 s{T1}, s{T1}, {lab('hbase')}-{lab('pc')}")
        a(f"s_addc_u32 s{T2}, s{T2}, 0")
        a(f"s_mov_b32 {BASE}, s{T1}")                 # 64 KiB aligned: its low half is zero
        a(f"s_mov_b32 s{sPC + 1}, s{T2}")
        a(f"s_mov_b32 s{sT0}, %[t0]")
        a(f"s_and_b32 s{sNB}, %[nb], 0xff")
    else:
        a("v_mbcnt_lo_u32_b32 v0, -1, 0")
        a("v_mbcnt_hi_u32_b32 v0, -1, v0")
        a("v_lshlrev_b32 v1, 4, v0")
        a(f"v_add_u32 v1, {END_}, v1")               # END_ carries the LDS base of X on entry
        a("v_mov_b32 v8, 0x7fc00000")
        a("s_load_dwordx4 s[8:11], %[karg], 0x0")    # program records, fitness
        a(f"s_load_dwordx2 s[{P3_}:{P4_}], %[karg], 0x10")  # work counter
        a("s_load_dwordx8 s[12:19], %[karg], 0x28")  # pop, D, var_len, tiles, batch, flags, query, record stride
        a("s_waitcnt lgkmcnt(0)")
        a(f"s_or_b32 s17, s17, {BASE}")               # BASE carries the trusted-variable bits (17-30, and 5-7) on entry
        a(f"v_mov_b32 v10, s{P3_}")
        a(f"v_mov_b32 v11, s{P4_}")
        a("s_mul_i32 s14, s14, s15")
        a(f"s_mul_i32 s14, s14, {G * 1024}")          # s14 = LDS distance from X to y
        a("v_lshrrev_b32 v13, 4, v0")                 # warm-up offset: 16 lanes per record, 16 bytes per lane
        a("v_mul_lo_u32 v13, v13, s19")
        a("v_and_b32 v9, 15, v0")
        a("v_lshl_add_u32 v13, v9, 4, v13")
        a(f"s_load_dwordx4 s[{P1_}:{P4_}], %[karg], 0x48")  # static trees per workgroup, first dynamic tree, waves per workgroup
        a("s_waitcnt lgkmcnt(0)")
        # Round 6: a workgroup's share of the population is a POOL its waves draw small batches from through a counter in LDS (STRIDE
        # carries its address on entry, zeroed by the kernel's prologue) instead of a fixed round-robin split between the waves: the waves
        # of a CU finish their share together whatever their trees cost, so the share can be most of the population and the counters in
        # global memory -- one line per XCD, saturated at ~90 grabs per microsecond: 17 % of a wave's clocks at 100 k trees, round 5 --
        # only hand out the rest, which evens out the workgroups.
        a(f"s_mul_i32 {CUR}, {CUR}, s{P1_}")          # CUR carries the workgroup id on entry: first tree of its pool
        a(f"s_add_u32 {END_}, {CUR}, s{P1_}")         # end of the pool
        a(f"s_mov_b32 {DYN}, s{P2_}")                 # first tree of the dynamic region
        # one dynamic region and one counter line per XCD: region = XCC_ID & flags[15:12], s18 still holds trees per region
        a(f"s_getreg_b32 s{P1_}, hwreg(HW_REG_XCC_ID, 0, 4)")
        a(f"s_bfe_u32 s{P2_}, s17, 0x4000c")
        a(f"s_and_b32 s{P1_}, s{P1_}, s{P2_}")
        a(f"s_mul_i32 s{P2_}, s{P1_}, s18")
        a(f"s_add_u32 {DYN}, {DYN}, s{P2_}")          # first tree of this XCD's region
        a(f"s_add_u32 s{P2_}, {DYN}, s18")
        a(f"s_min_u32 s12, s12, s{P2_}")              # s12: end of this XCD's region (the population size is not needed again)
        a(f"s_lshl_b32 s{P1_}, s{P1_}, 7")            # this XCD's counter: 128 bytes per region
        a(f"v_add_co_u32 v10, vcc, s{P1_}, v10")
        a("v_addc_co_u32 v11, vcc, 0, v11, vcc")
        a(f"s_getpc_b64 s[{T1}:{T2}]")
        a(f"{lab('pc')}:")
        a(f"s_add_u32 s{T1}, s{T1}, {lab('hbase')}-{lab('pc')}")
        a(f"s_addc_u32 s{T2}, s{T2}, 0")
        a(f"s_mov_b32 {BASE}, s{T1}")                 # 64 KiB aligned: its low half is zero
        a(f"s_mov_b32 s{sPC + 1}, s{T2}")
        if CODEWARM:
            # The handler table and the bodies behind it (~100 KiB) have left the L2 of this XCD since the last launch (a call moves
            # 900 MB): every wave asks for all of it, one line per lane, before its first instruction fetch misses there -- the
            # instruction cache then finds its lines in L2.  Experiment, off: see CODEWARM
            a("v_lshlrev_b32 v4, 6, v0")
            a(f"s_mov_b32 s{P3_}, ({lab('code_end')}-{lab('hbase')}+4095)/4096")
            a(f"{lab('codewarm')}:")
            a(f"global_load_dword v14, v4, s[{T1}:{T2}]")
            a(f"s_add_u32 s{T1}, s{T1}, 4096")
            a(f"s_addc_u32 s{T2}, s{T2}, 0")
            a(f"s_sub_u32 s{P3_}, s{P3_}, 1")
            a(f"s_cmp_lg_u32 s{P3_}, 0")
            a(f"s_cbranch_scc1 {lab('codewarm')}")
        a(f"{lab('run')}:")
        if stats:
            for r in (A_REC, A_WORK, A_TREES, A_DISP):
                a(f"v_mov_b32 v{r}, 0")
            a(f"s_memtime s[{T1}:{T2}]")
            a("s_waitcnt lgkmcnt(0)")
            a(f"v_mov_b32 v{A_START}, s{T1}")
        a("s_mov_b32 s18, 1")  # 1 while the workgroup's pool lasts
        pool_grab()
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_readfirstlane_b32 s{sT0N}, v12")
        a(f"s_add_u32 s{sT0N}, s{sT0N}, {CUR}")
        # ------------------------------------------------------------------ batch loop
        a(f"{lab('batch')}:")
        a("s_cmp_eq_u32 s18, 0")
        a(f"s_cbranch_scc1 {lab('dyn')}")
        a(f"s_mov_b32 s{sT0}, s{sT0N}")
        a(f"s_cmp_lt_u32 s{sT0}, {END_}")
        a(f"s_cbranch_scc0 {lab('to_dyn')}")
        pool_grab()                                    # the next batch, one batch ahead: consumed behind this batch's second tree
        warm(f"s{sT0}")
        a(f"s_sub_u32 s{sNB}, {END_}, s{sT0}")
        dyn_batch(T1)
        a(f"s_min_u32 s{sNB}, s{sNB}, s{T1}")
        a(f"s_branch {lab('have_batch')}")
        # the pool is used up (every wave of the workgroup finds that out by itself, a batch after its last one): the rest of the launch
        # comes from this XCD's region.  The one grab whose latency nothing hides.
        a(f"{lab('to_dyn')}:")
        a("s_mov_b32 s18, 0")
        tick_begin()
        grab()
        a("s_waitcnt vmcnt(0)")
        tick_end(A_WORK)
        a(f"v_readfirstlane_b32 s{sT0N}, v12")
        a(f"s_add_u32 s{sT0N}, s{sT0N}, {DYN}")
        a(f"{lab('dyn')}:")
        a(f"s_mov_b32 s{sT0}, s{sT0N}")
        a(f"s_cmp_ge_u32 s{sT0}, s12")
        a(f"s_cbranch_scc1 {lab('exit')}")
        grab()                                         # the next batch, consumed at the end of this one
        warm(f"s{sT0}")
        a(f"s_sub_u32 s{sNB}, s12, s{sT0}")
        dyn_batch(T1)
        a(f"s_min_u32 s{sNB}, s{sNB}, s{T1}")
    a(f"{lab('have_batch')}:")
    a(f"s_mov_b32 s{sB}, 0")
    a(f"s_mov_b64 s[{sOK}:{sOK + 1}], 0")
    a("v_mov_b32 v7, 0")
    # ------------------------------------------------------------------ tree loop
    a(f"{lab('tree')}:")
    if fused:
        rec_addr(sREC, sREC + 1, f"s{sB}", T1)   # slot b of the wave's ring
    else:
        a(f"s_add_u32 s{T1}, s{sT0}, s{sB}")
        rec_addr(sREC, sREC + 1, f"s{T1}", T1)
    tick_begin()
    load_window()
    a(f"{lab('tree_loaded')}:")   # (END of the batch's previous tree arrives here with this record already on its way)
    a("v_mov_b32 v6, 0")
    a(f"s_mov_b32 s{sTILE}, 0")
    a("s_waitcnt lgkmcnt(0)")
    tick_end(A_REC)
    if not fused:
        # the second tree of a pool batch: the grab made at the batch's start has landed (nothing is in flight behind the wait above),
        # so the NEXT batch is known -- its records are pulled into L2 while the rest of this one runs
        a(f"s_cmp_eq_u32 s{sB}, 1")
        a(f"s_cbranch_scc0 {lab('no_prewarm')}")
        a("s_cmp_eq_u32 s18, 1")
        a(f"s_cbranch_scc0 {lab('no_prewarm')}")
        a(f"v_readfirstlane_b32 s{sT0N}, v12")
        a(f"s_add_u32 s{sT0N}, s{sT0N}, {CUR}")
        a(f"s_cmp_lt_u32 s{sT0N}, {END_}")
        a(f"s_cbranch_scc0 {lab('pool_dry')}")
        warm(f"s{sT0N}")
        a(f"s_branch {lab('no_prewarm')}")
        # ... or the pool has run dry: this wave's next batch comes from its XCD's region, and the grab goes out now -- a tree ahead of
        # its use, and at a moment of this wave's own (left to the batch's end, all waves of a workgroup -- of most workgroups -- would
        # queue at the counters together, with nothing to do meanwhile)
        a(f"{lab('pool_dry')}:")
        grab()
        a("s_mov_b32 s18, 2")
        a(f"{lab('no_prewarm')}:")
    if stats:
        a(f"v_add_u32 v{A_TREES}, 1, v{A_TREES}")
    elif KWARM and not fused:
        # pull the NEXT tree's record into the scalar cache while this tree runs: four one-line loads whose result nobody
        # reads (the kernarg pointer pair is dead after the prologue and serves as the sink).  A record is 256 bytes
        # (asserted on the host), the next tree of the batch is the next record; the slack behind the last record
        # covers the last tree.  Handlers count LDS completions with s_waitcnt lgkmcnt(n): scalar loads in flight can only
        # make such a wait longer, never shorter, because LDS operations complete in order.
        for i in range(KWARM_LINES):
            a(f"s_load_dwordx2 %[karg], s[{sREC}:{sREC + 1}], {hex(256 + 64 * i)}")
        # ... and, on the last tree of a STATIC batch, the first record of the wave's next batch (CUR already points at it)
        a(f"s_add_u32 s{T1}, s{sB}, 1")
        a(f"s_cmp_lt_u32 s{T1}, s{sNB}")
        a(f"s_cbranch_scc1 {lab('kwarm_done')}")
        a("s_cmp_eq_u32 s18, 1")
        a(f"s_cbranch_scc0 {lab('kwarm_done')}")
        a(f"s_cmp_lt_u32 {CUR}, {END_}")
        a(f"s_cbranch_scc0 {lab('kwarm_done')}")
        a(f"s_mul_hi_u32 s{T2}, {CUR}, s19")
        a(f"s_mul_i32 s{T1}, {CUR}, s19")
        a(f"s_add_u32 s{T1}, s{T1}, s8")
        a(f"s_addc_u32 s{T2}, s{T2}, s9")
        for i in range(KWARM_LINES):
            a(f"s_load_dwordx2 %[karg], s[{T1}:{T2}], {hex(64 * i)}")
        a(f"{lab('kwarm_done')}:")
    # ------------------------------------------------------------------ tile loop (one pass of the program)
    a(f"{lab('tile')}:")
    a(f"s_mul_i32 s{T1}, s{sTILE}, {G * 1024}")
    a(f"v_add_u32 v2, s{T1}, v1")
    a("v_add_u32 v3, s14, v2")
    a(f"s_mov_b32 s{sH}, 0")
    a(f"s_mov_b32 s{sJ}, 0")
    a(f"{lab('tile_go')}:")
    a(f"s_lshr_b32 {PF}, s{W}, 24")                # variable operand of the first instruction -> bank 0
    a(f"v_lshl_add_u32 v4, {PF}, 10, v2")
    read_bank(P[0], 4)
    a(f"s_set_gpr_idx_on s{sJ}, 0")  # J == 0: enables indexing with no operand selected, M0 = 0
    a(f"s_pack_lh_b32_b16 s{sPC}, s{W}, {BASE}")
    a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")

    # ------------------------------------------------------------------ handlers
    a(".p2align 16")  # 64 KiB: a handler address is {high half of the table address, 16-bit offset from the program word}
    a(f"{lab('hbase')}:")

    def begin(name, fl):
        if twin[0]:
            name += "_np"
        a(f".org {lab('hbase')}+{SLOT * (fl * NHF + hid[name])}")  # fails to assemble if the previous handler overflowed its slot
        a(f"{lab(f'h{fl}_' + name)}:")

    MODE = {"SRC0": 1, "SRC1": 2, "SRC2": 4, "DST": 8}

    def m0_stack(mode_bits, off):
        """M0 = (mode << 12) + H + off   (off in registers, may be negative)"""
        imm = (mode_bits << 12) + off
        a(f"s_add_u32 m0, s{sH}, {hex(imm & 0xFFFFFFFF)}")

    def wait_cur():
        a(f"s_waitcnt lgkmcnt({0 if twin[0] else G})")  # everything but the prefetch just issued has landed (LDS returns in order)

    def splat(sreg):
        """the operand a constant takes in a K-row loop: its SGPR (default), or -- an experiment, the kernel-level ubench had an
        SGPR source at half the issue rate -- v9 after one copy (a source-0 position that no handler addresses relative to
        M0).  In the interpreter the copy costs more than it saves: 1.066 -> 1.081 ms at 1 M trees."""
        if not SPLAT or K < 3:
            return f"s{sreg}"
        a(f"v_mov_b32 v9, s{sreg}")
        return "v9"

    def rows_op(op, dst, x, y):
        """K rows of dst = x op y on registers.  While VGPR indexing is enabled a VOP2 add / sub / mul issues at the 4-clock rate like
        everything else (scripts/ubench/valu_rates.hip: k_add_idx_* against k_add_vop2; the interpreter's SQ_ACTIVE_INST_VALU /
        SQ_INSTS_VALU is 4.3), so a packed instruction -- two rows in those 4 clocks -- halves the cost.  a - b is a + (-b)."""
        if PKARITH and K >= 2:
            pkins = "v_pk_mul_f32" if op == "mul" else "v_pk_add_f32"
            neg = " neg_lo:[0,1] neg_hi:[0,1]" if op == "sub" else ""
            for k in range(0, K, 2):
                a(f"{pkins} v[{dst + k}:{dst + k + 1}], v[{x + k}:{x + k + 1}], v[{y + k}:{y + k + 1}]{neg}")
        else:
            ins = {"add": "v_add_f32", "sub": "v_sub_f32", "mul": "v_mul_f32"}[op]
            for k in range(K):
                a(f"{ins} v{dst + k}, v{x + k}, v{y + k}")

    def rows_mov(dst, src):
        """K rows of dst = src (registers), packed where it pays (see rows_op)"""
        if PKARITH and K >= 2:
            for k in range(0, K, 2):
                a(f"v_pk_mov_b32 v[{dst + k}:{dst + k + 1}], v[{src + k}:{src + k + 1}], v[{src + k}:{src + k + 1}] op_sel:[0,1]")
        else:
            for k in range(K):
                a(f"v_mov_b32 v{dst + k}, v{src + k}")

    def arith(op, form, fl, name=None, guard=None):
        cur, nxt = P[fl], P[1 - fl]
        ins = {"add": "v_add_f32", "sub": "v_sub_f32", "mul": "v_mul_f32"}[op]
        rev = {"add": "v_add_f32", "sub": "v_subrev_f32", "mul": "v_mul_f32"}[op]
        begin(name or f"{op}_{form}", fl)
        if guard:
            guard()
        entry()
        if form == "SS":
            prefetch(nxt)
            m0_stack(MODE["SRC0"] | MODE["SRC1"] | MODE["DST"], -2 * K)
            rows_op(op, S0, S0 + K, S0)
            a(f"s_sub_u32 s{sH}, s{sH}, {K}")
        elif form == "SV":  # stack top (left) op variable
            prefetch(nxt)
            m0_stack(MODE["SRC0"] | MODE["DST"], -K)
            wait_cur()
            rows_op(op, S0, S0, cur)
        elif form == "VS":  # variable (left) op stack top
            prefetch(nxt)
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            wait_cur()
            rows_op(op, S0, cur, S0)
        elif form in ("SC", "CS") and PKCONST and K >= 2:
            # an SGPR source makes a VOP2 the slow class (4 clocks); a packed instruction takes the constant for two rows in
            # the same 4 clocks (op_sel_hi 0: both halves read the SGPR).  a - b is a + (-b) bit for bit
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["SRC0"] | MODE["DST"], -K)
            pkins = "v_pk_mul_f32" if op == "mul" else "v_pk_add_f32"
            neg = "" if op != "sub" else (" neg_lo:[0,1] neg_hi:[0,1]" if form == "SC" else " neg_lo:[1,0] neg_hi:[1,0]")
            for k in range(0, K, 2):
                a(f"{pkins} v[{S0 + k}:{S0 + k + 1}], v[{S0 + k}:{S0 + k + 1}], s[{sA}:{sA + 1}] op_sel_hi:[1,0]{neg}")
        elif form == "SC":  # stack top op constant
            a(f"s_movrels_b32 s{sBop}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            c = splat(sBop)
            for k in range(K):
                a(f"{rev} v{S0 + k}, {c}, v{S0 + k}")
        elif form == "CS":  # constant op stack top
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            c = splat(sA)
            for k in range(K):
                a(f"{ins} v{S0 + k}, {c}, v{S0 + k}")
        elif form == "VV":  # a was prefetched, b is read here
            a(f"s_movrels_b32 s{sBop}, s{W + 1}")
            a(f"v_lshl_add_u32 v5, s{sBop}, 4, v2")
            read_bank(T, 5)
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            wait_cur()
            rows_op(op, S0, cur, T)
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        elif form in ("VC", "CV") and PKCONST and K >= 2:
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            pkins = "v_pk_mul_f32" if op == "mul" else "v_pk_add_f32"
            neg = "" if op != "sub" else (" neg_lo:[0,1] neg_hi:[0,1]" if form == "VC" else " neg_lo:[1,0] neg_hi:[1,0]")
            wait_cur()
            for k in range(0, K, 2):
                a(f"{pkins} v[{S0 + k}:{S0 + k + 1}], v[{cur + k}:{cur + k + 1}], s[{sA}:{sA + 1}] op_sel_hi:[1,0]{neg}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        elif form == "VC":  # variable op constant
            a(f"s_movrels_b32 s{sBop}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            c = splat(sBop)
            wait_cur()
            for k in range(K):
                a(f"{rev} v{S0 + k}, {c}, v{cur + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        elif form == "CV":  # constant op variable
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            c = splat(sA)
            wait_cur()
            for k in range(K):
                a(f"{ins} v{S0 + k}, {c}, v{cur + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        epilogue()

    DIV_KIND = {"SS": "Tc", "SV": "Tc", "VS": "cT", "SC": "Tc", "CS": "Tc", "VV": "cT", "VC": "cT", "CV": "Tc"}

    def div_stub(form, fl):
        """put a into x and b into y (one of them the current operand bank, the other T), set the scatter index of the
        result, adjust H, go to the division body for this bank assignment"""
        cur, nxt = P[fl], P[1 - fl]
        kind = DIV_KIND[form]
        x, y = (T, cur) if kind == "Tc" else (cur, T)
        begin(f"div_{form}", fl)
        entry()
        la, rb = form[0], form[1]
        if la == "C":
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
        if rb == "C" or form == "VV":
            a(f"s_movrels_b32 s{sBop}, s{W + 1}")
        if form == "VV":
            a(f"v_lshl_add_u32 v5, s{sBop}, 4, v2")
            read_bank(T, 5)
        vtrust = use_range and TRUST and form in ("CV", "VV", "SV", "VS")
        if vtrust:
            a(f"s_movrels_b32 s{T1}, s{W}")  # this instruction's word: its top byte names the variable in the current bank
        prefetch(nxt)
        if vtrust and form in ("CV", "VV"):
            # a variable whose column lies in [2^-46, 2^46] (flags bits 17-30, sr_tc_kernel) needs no range test: c / v and v / w
            # have bodies of their own that read the operands where they are; anything else comes back to the label below
            a(f"s_branch {lab(f'divv_{form}{fl}')}")
            a(f"{lab(f'divv_old_{form}{fl}')}:")
        if form in DIVIP:
            a(f"{lab(f'divold_{form}{fl}')}:")  # the in-place handler of this form arrives here when its block holds a zero divisor
        wait_cur()  # the current bank is overwritten or read below: its (possibly unused) prefetch must have landed
        if form == "SS":
            m0_stack(MODE["SRC0"] | MODE["SRC1"], -2 * K)   # (a packed move reads its two rows through source 0 and source 1)
            rows_mov(x, S0 + K)
            rows_mov(y, S0)
            a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - 2 * K)}")
            a(f"s_sub_u32 s{sH}, s{sH}, {K}")
        elif la == "S" or rb == "S":
            m0_stack(MODE["SRC0"] | MODE["SRC1"], -K)
            bank = x if la == "S" else y
            rows_mov(bank, S0)
            a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
        else:
            a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
        a("s_mov_b32 m0, 0")
        if la == "C":
            for k in range(K):
                a(f"v_mov_b32 v{x + k}, s{sA}")
        if rb == "C":
            for k in range(K):
                a(f"v_mov_b32 v{y + k}, s{sBop}")
        if vtrust and form in ("SV", "VS"):  # only the stack operand (now in T) needs the test when the variable is trusted
            trust_test(T1, 24, lab(f"divbody_{kind}{fl}"))
            a(f"s_branch {lab(f'divbody_{kind}{fl}_t')}")
        else:
            a(f"s_branch {lab(f'divbody_{kind}{fl}')}")

    def trust_test(reg, shift, untrusted):
        """s{reg} >> (shift + flags[7:5]) = index of a variable (flags[7:5] = log2 of the 1-KiB units between two variables in LDS,
        set with the trust bits by sr_tc_kernel's prologue, which trusts nobody when that distance is no power of two); falls
        through when the prologue found the variable's whole column in range (flags bit 17 + index; bit 31 is never set and
        stands for every variable from the 14th on)"""
        a(f"s_bfe_u32 s{T2}, s17, 0x30005")
        a(f"s_add_u32 s{T2}, s{T2}, {shift}")
        a(f"s_lshr_b32 s{reg}, s{reg}, s{T2}")
        a(f"s_add_u32 s{reg}, s{reg}, 17")
        a(f"s_min_u32 s{reg}, s{reg}, 31")
        a(f"s_bitcmp1_b32 s17, s{reg}")
        a(f"s_cbranch_scc0 {untrusted}")

    def opnd(o):
        return o if isinstance(o, str) else f"v{o}"

    def div_rows(xs, ys, qs, nanfix=True, temps=None):
        """IEEE division rows: q = (y == 0) ? NaN : x / y  up to (not including) v_div_fixup (forward.cu:183-187).
        Operands are VGPR numbers or operand strings (an SGPR holding a constant); `temps`: five VGPRs (default v18-v22)."""
        d3, d4, d6, d7, d8 = temps or DT
        xs, ys = [opnd(o) for o in xs], [opnd(o) for o in ys]
        assert not nanfix or all(o.startswith("v") for o in xs)
        if fast == 2:
            # "short" division: the IEEE sequence with its range scaling (v_div_scale / v_div_fmas / v_div_fixup) but ONE
            # residual correction instead of a refined reciprocal and two corrections: 9 instead of 13 operations.  Range-
            # safe like the full sequence; the quotient is the correctly rounded one except when the exact quotient lies
            # within ~2^-30 ulp of a rounding boundary (measured: 1 of 2^32 random mantissa pairs, ubench/div_faithful.hip).
            for x, y, q in zip(xs, ys, qs):
                if nanfix:
                    a(f"v_cmp_neq_f32 vcc, 0, {y}")
                    a(f"v_cndmask_b32 {x}, v8, {x}, vcc")
                a(f"v_div_scale_f32 v{d3}, vcc, {y}, {y}, {x}")
                a(f"v_rcp_f32 v{d4}, v{d3}")
                a(f"v_div_scale_f32 v{d6}, vcc, {x}, {y}, {x}")
                a(f"v_mul_f32 v{d7}, v{d6}, v{d4}")
                a(f"v_fma_f32 v{d8}, -v{d3}, v{d7}, v{d6}")
                a("s_nop 1")  # v_div_fmas reads the VCC of the second v_div_scale: 4 wait states
                a(f"v_div_fmas_f32 v{q}, v{d8}, v{d4}, v{d7}")
            return
        if fast:
            # "fast" division (opt-in): reciprocal, quotient, ONE residual correction -- faithfully rounded (the result is
            # the correctly rounded quotient unless the exact quotient lies within ~2^-23 ulp of a rounding boundary),
            # no range scaling: |y| > 2^126 gives 0, a quotient beyond 2^128 gives NaN.  Special operands are still put
            # right by v_div_fixup.  5 instead of 11 dependent operations per row.
            for x, y, q in zip(xs, ys, qs):
                a(f"v_rcp_f32 v{d4}, {y}")
                if nanfix:
                    a(f"v_cmp_neq_f32 vcc, 0, {y}")
                    a(f"v_cndmask_b32 {x}, v8, {x}, vcc")
                else:
                    a("s_nop 0")  # gfx940+: one wait state between a transcendental and the VALU that reads its result
                a(f"v_mul_f32 v{d7}, {x}, v{d4}")
                a(f"v_fma_f32 v{d8}, -{y}, v{d7}, {x}")
                a(f"v_fma_f32 v{q}, v{d8}, v{d4}, v{d7}")
            return
        for x, y, q in zip(xs, ys, qs):
            if nanfix:
                a(f"v_cmp_neq_f32 vcc, 0, {y}")
                a(f"v_cndmask_b32 {x}, v8, {x}, vcc")  # a NaN numerator makes the quotient NaN
            a(f"v_div_scale_f32 v{d3}, vcc, {y}, {y}, {x}")
            a(f"v_rcp_f32 v{d4}, v{d3}")
            a(f"v_div_scale_f32 v{d6}, vcc, {x}, {y}, {x}")
            a(f"v_fma_f32 v{d7}, -v{d3}, v{d4}, 1.0")
            a(f"v_fmac_f32 v{d4}, v{d7}, v{d4}")
            a(f"v_mul_f32 v{d7}, v{d6}, v{d4}")
            a(f"v_fma_f32 v{d8}, -v{d3}, v{d7}, v{d6}")
            a(f"v_fmac_f32 v{d7}, v{d8}, v{d4}")
            a(f"v_fma_f32 v{d3}, -v{d3}, v{d7}, v{d6}")
            a(f"v_div_fmas_f32 v{q}, v{d3}, v{d4}, v{d7}")

    # ---- the short division without its range scaling.  v_div_scale changes an operand only when an exponent is extreme
    # (numerator and denominator 2^96 apart, a denormal operand, reciprocal or quotient; MI300 ISA, V_DIV_SCALE_F32); where it
    # does not, the short rows are  r = rcp(y), q0 = x * r, e = fma(-y, q0, x), q = fma(e, r, q0)  -- v_div_fmas is a plain fma
    # when VCC is clear -- and v_div_fixup hands a finite quotient of finite operands through.  A block whose 2 K operands all
    # lie in [2^-46, 2^46] (NaN operands pass: they give NaN either way) runs exactly these operations, the two fmas as
    # v_pk_fma_f32 over row pairs (two fmas for the issue slot of one, scripts/ubench/valu_rates.hip), the second one straight
    # into the result's place: 14 instead of 31 VALU clocks per row, the same bits for every quotient that is a number (a NaN
    # may carry another payload: the tree's sum is canonicalised where it leaves the wave).  The test is one v_min3 and one
    # v_max3 per two values, kept apart for numerator and denominator, because the blocks that fail it are nearly always
    # (headline forest: always) a numerator or a denominator that is +-0 in every row and lane -- the constant 0 and what
    # multiplications make of it: such a block is K v_div_fixup (0 / y) or K moves of NaN (x / 0, forward.cu:183-187).
    # Any other block takes the rows with v_div_scale / v_div_fmas.
    use_range = DIVRANGE and fast in (1, 2) and K >= 2   # (the fast mode's own rows only run in blocks that fail the test)

    def minmax(vals, lo, hi):
        vals = [opnd(o) for o in vals]
        assert all(o.startswith("v") for o in vals) and len(vals) >= 3
        a(f"v_min3_f32 v{lo}, |{vals[0]}|, |{vals[1]}|, |{vals[2]}|")
        a(f"v_max3_f32 v{hi}, |{vals[0]}|, |{vals[1]}|, |{vals[2]}|")
        rest = vals[3:]
        while len(rest) >= 2:
            a(f"v_min3_f32 v{lo}, v{lo}, |{rest[0]}|, |{rest[1]}|")
            a(f"v_max3_f32 v{hi}, v{hi}, |{rest[0]}|, |{rest[1]}|")
            rest = rest[2:]
        if rest:
            a(f"v_min_f32 v{lo}, v{lo}, |{rest[0]}|")
            a(f"v_max_f32 v{hi}, v{hi}, |{rest[0]}|")

    def range_test(xvals, yvals, t, slow, between=None):
        """-> registers {lox, hix, loy, hiy} (those of an operand that is no constant); falls through when every value is in
        range or NaN, goes to `slow` otherwise.  between(): scalar tests placed behind the statistics"""
        st = {}
        if xvals:
            minmax(xvals, t[0], t[1])
            st["lox"], st["hix"] = t[0], t[1]
        if yvals:
            minmax(yvals, t[2], t[3])
            st["loy"], st["hiy"] = t[2], t[3]
        if xvals and yvals:
            a(f"v_min_f32 v{t[4]}, v{t[0]}, v{t[2]}")
            a(f"v_max_f32 v{t[5]}, v{t[1]}, v{t[3]}")
            lo, hi = t[4], t[5]
        else:
            lo, hi = (t[0], t[1]) if xvals else (t[2], t[3])
        if between:
            between()
        a(f"v_cmp_gt_f32 vcc, {hex(DIV_LO)}, v{lo}")
        a(f"s_cbranch_vccnz {slow}")
        a(f"v_cmp_lt_f32 vcc, {hex(DIV_HI)}, v{hi}")
        a(f"s_cbranch_vccnz {slow}")
        return st

    def pk(o, k):
        """operand of a packed instruction for rows k, k + 1: an aligned register pair -> (text, op_sel_hi bit); a constant's
        SGPR is read for both rows (op_sel_hi 0: the high lane takes the low dword too)"""
        if isinstance(o, str):
            n = int(o[1:])
            assert o[0] == "s" and n % 2 == 0, o
            return f"s[{n}:{n + 1}]", 0
        assert o[k] % 2 == 0 and o[k + 1] == o[k] + 1, o
        return f"v[{o[k]}:{o[k] + 1}]", 1

    def fast_pair_rows(xs, ys, temps, out, out_m0=None):
        """rows of a range-tested block.  xs / ys: K VGPR numbers, or ONE SGPR string for a constant operand; the quotient of
        row k goes to out[k].  temps: six registers (three aligned pairs) -- or twelve: two row pairs then run abreast, each
        instruction four or more away from the one it depends on (a wave issues in order; the reciprocal's latency otherwise
        waits for the other waves of the SIMD to fill).  out_m0: SGPR holding the M0 that indexes the destination (gather
        bodies: only the result is indexed, so M0 is switched around the instructions that write it); None: M0 stays as it is
        (in-place handlers: every operand is indexed)"""
        sets = [temps[i:i + 6] for i in range(0, len(temps), 6)]
        for r0, r1, q0, q1, e0, e1 in sets:
            assert r0 % 2 == 0 and q0 % 2 == 0 and e0 % 2 == 0 and (r1, q1, e1) == (r0 + 1, q0 + 1, e0 + 1)
        cx, cy = isinstance(xs, str), isinstance(ys, str)
        if cy:  # one reciprocal serves the block
            a(f"v_rcp_f32 v{sets[0][0]}, {ys}")
            a("s_nop 0")
            for st_ in sets:
                for r in st_[:2]:
                    if r != sets[0][0]:
                        a(f"v_mov_b32 v{r}, v{sets[0][0]}")
        pairs = list(range(0, K, 2))
        for g0 in range(0, len(pairs), len(sets)):
            grp = list(zip(pairs[g0:g0 + len(sets)], sets))
            X = {k: ([xs, xs] if cx else [f"v{xs[k]}", f"v{xs[k + 1]}"]) for k, _ in grp}
            Y = {k: ([ys, ys] if cy else [f"v{ys[k]}", f"v{ys[k + 1]}"]) for k, _ in grp}
            if not cy:
                for k, (r0, r1, *_) in grp:
                    a(f"v_rcp_f32 v{r0}, {Y[k][0]}")
                    a(f"v_rcp_f32 v{r1}, {Y[k][1]}")   # (also the wait state between a transcendental and the reader of its result)
            if PKARITH and not cy and len(grp) == 1:
                a("s_nop 0")  # one pair at a time: the packed product reads the reciprocal issued just before it (one wait state)
            for k, (r0, r1, q0, q1, e0, e1) in grp:
                if PKARITH:
                    xp, xh = pk(xs, k)
                    a(f"v_pk_mul_f32 v[{q0}:{q1}], {xp}, v[{r0}:{r1}] op_sel_hi:[{xh},1]")
                else:
                    a(f"v_mul_f32 v{q0}, {X[k][0]}, v{r0}")
                    a(f"v_mul_f32 v{q1}, {X[k][1]}, v{r1}")
            for k, (r0, r1, q0, q1, e0, e1) in grp:
                xp, xh = pk(xs, k)
                yp, yh = pk(ys, k)
                assert out[k] % 2 == 0 and out[k + 1] == out[k] + 1
                a(f"v_pk_fma_f32 v[{e0}:{e1}], {yp}, v[{q0}:{q1}], {xp} op_sel_hi:[{yh},1,{xh}] neg_lo:[1,0,0] neg_hi:[1,0,0]")
            if DIVFIX:
                for k, (r0, r1, q0, q1, e0, e1) in grp:
                    a(f"v_pk_fma_f32 v[{e0}:{e1}], v[{e0}:{e1}], v[{r0}:{r1}], v[{q0}:{q1}]")
            if out_m0 is not None:
                a(f"s_mov_b32 m0, s{out_m0}")
            for k, (r0, r1, q0, q1, e0, e1) in grp:
                if DIVFIX:
                    a(f"v_div_fixup_f32 v{out[k]}, v{e0}, {Y[k][0]}, {X[k][0]}")
                    a(f"v_div_fixup_f32 v{out[k + 1]}, v{e1}, {Y[k][1]}, {X[k][1]}")
                else:
                    a(f"v_pk_fma_f32 v[{out[k]}:{out[k] + 1}], v[{e0}:{e1}], v[{r0}:{r1}], v[{q0}:{q1}]")
            if out_m0 is not None and g0 + len(sets) < len(pairs):
                a("s_mov_b32 m0, 0")

    def zero_blocks(st, xzero, yzero):
        """behind a failed range test: a numerator that is +-0 (or NaN) in every row and lane -> label xzero, a denominator
        likewise -> yzero; falls through for any other block.  The two labels' rows are emitted by zero_rows."""
        if "hix" in st:
            a(f"v_cmp_neq_f32 vcc, 0, v{st['hix']}")
            a(f"s_cbranch_vccz {xzero}")
        if "hiy" in st:
            a(f"v_cmp_neq_f32 vcc, 0, v{st['hiy']}")
            a(f"s_cbranch_vccz {yzero}")

    def zero_rows(kind, xs, ys, out, relative):
        """kind 'x': 0 / y  = +-0, or NaN where y is 0 or NaN -- v_div_fixup's own rules, its quotient operand is not looked at;
        kind 'y': x / 0 = NaN whatever x is (forward.cu:183-187).  relative: the sources are indexed too (in-place handlers), so
        the NaN comes as a literal and not from v8"""
        for k in range(K):
            x = xs if isinstance(xs, str) else f"v{xs[k]}"
            y = ys if isinstance(ys, str) else f"v{ys[k]}"
            if kind == "x":
                a(f"v_div_fixup_f32 v{out[k]}, {y if not isinstance(ys, str) else x}, {y}, {x}")
            else:
                a(f"v_mov_b32 v{out[k]}, " + ("0x7fc00000" if relative else "v8"))

    for fl in (0, 1):
        cur, nxt = P[fl], P[1 - fl]
        for op in ("add", "sub", "mul"):
            for form in FORMS:
                arith(op, form, fl)
        for form in FORMS:
            div_stub(form, fl)
        # push constant (folded constant subtree, or a tree that is a single constant)
        def push_c(fl=fl, cur=cur, nxt=nxt):
            begin("push_c", fl)
            entry()
            a(f"s_movrels_b32 s{sA}, s{W + 1}")
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            if PKCONST and K >= 2:
                for k in range(0, K, 2):
                    a(f"v_pk_mov_b32 v[{S0 + k}:{S0 + k + 1}], s[{sA}:{sA + 1}], s[{sA}:{sA + 1}] op_sel:[0,0]")
            else:
                c = splat(sA)
                for k in range(K):
                    a(f"v_mov_b32 v{S0 + k}, {c}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
            epilogue()

        # push variable (a tree that is a single variable)
        def push_v(fl=fl, cur=cur, nxt=nxt):
            begin("push_v", fl)
            entry()
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            wait_cur()
            rows_mov(S0, cur)
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
            epilogue()
        push_c()
        push_v()
        begin("end", fl)
        a(f"s_branch {lab(f'endbody{fl}')}")
        # a tree the compiler could not take: leave its (marked) fitness word alone
        begin("skip", fl)
        a("s_set_gpr_idx_off")
        a(f"s_branch {lab('next_tree')}")
        # unary functions: the operand is the top of the stack (S: replaced in place) or a variable (V: pushed)
        for uop in UNARY:
            bit = {"neg": ("v_xor_b32", "0x80000000"), "abs": ("v_and_b32", "0x7fffffff")}.get(uop)
            if bit is None:
                # functions with a shared body: gather the operand (its magnitude for the loose variants) into the T bank,
                # remember where the result goes, run the body
                body = {"lsqrt": "sqrt"}.get(uop, uop)
                mag = uop in ("lsqrt", "llog")
                begin(f"{uop}_S", fl)
                entry()
                prefetch(nxt)
                m0_stack(MODE["SRC1"] if mag else MODE["SRC0"], -K)
                for k in range(K):
                    if mag:
                        a(f"v_and_b32 v{T + k}, 0x7fffffff, v{S0 + k}")
                    else:
                        a(f"v_mov_b32 v{T + k}, v{S0 + k}")
                a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
                a("s_mov_b32 m0, 0")
                a(f"s_branch {lab(f'trigbody_{body}')}")
                begin(f"{uop}_V", fl)
                entry()
                prefetch(nxt)
                wait_cur()
                for k in range(K):
                    if mag:
                        a(f"v_and_b32 v{T + k}, 0x7fffffff, v{cur + k}")
                    else:
                        a(f"v_mov_b32 v{T + k}, v{cur + k}")
                a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
                a(f"s_add_u32 s{sH}, s{sH}, {K}")
                a(f"s_branch {lab(f'trigbody_{body}')}")
                continue
            begin(f"{uop}_S", fl)
            entry()
            prefetch(nxt)
            m0_stack(MODE["SRC1"] | MODE["DST"], -K)
            for k in range(K):
                a(f"{bit[0]} v{S0 + k}, {bit[1]}, v{S0 + k}")
            epilogue()
            begin(f"{uop}_V", fl)
            entry()
            prefetch(nxt)
            m0_stack(MODE["DST"], 0)
            wait_cur()
            for k in range(K):
                a(f"{bit[0]} v{S0 + k}, {bit[1]}, v{cur + k}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")
            epilogue()
        # NEXT: the program continues in the tree's overflow block (word 0).  Refill the window, then do what every
        # handler does at its entry for the instruction that follows: handler address, operand prefetch into the other bank.
        begin("next", fl)
        a(f"s_movrels_b32 s{sA}, s{W + 1}")                  # byte distance to the overflow block
        a(f"s_add_u32 s{sREC}, s{sREC}, s{sA}")
        a(f"s_addc_u32 s{sREC + 1}, s{sREC + 1}, 0")
        load_window()
        # (sREC now points at the overflow block: the tile loop sees that and reloads the first block for the next pass)
        a("s_waitcnt lgkmcnt(0)")
        a(f"s_pack_lh_b32_b16 s{sPC}, s{W}, {BASE}")
        a(f"s_lshr_b32 {PF}, s{W}, 24")
        prefetch(nxt)
        a(f"s_mov_b32 s{sJ}, 0")
        a(f"s_mov_b32 m0, s{sJ}")
        a(f"s_setpc_b64 s[{sPC}:{sPC + 1}]")
        # ---- generic binary stubs: a -> T bank, b -> Q bank (whatever the form), result slot in sDST, then the body the
        # word's aux field names (max min < > <= >= loose-div ...: functions too rare or too long for one handler per form)
        for form in FORMS:
            begin(f"gbin_{form}", fl)
            entry()
            read_aux(T2)
            la, rb = form[0], form[1]
            if la == "C" or rb == "C" or form == "VV":
                a(f"s_movrels_b32 s{sA}, s{W + 1}")           # the word's one 32-bit operand: constant, or second variable
            if form == "VV":
                a(f"v_lshl_add_u32 v5, s{sA}, 4, v2")
                read_bank(Q, 5)
            prefetch(nxt)
            wait_cur()  # bodies may use the current operand bank as scratch: its prefetch must have landed
            if form == "SS":
                m0_stack(MODE["SRC0"], -2 * K)
                for k in range(K):
                    a(f"v_mov_b32 v{T + k}, v{S0 + K + k}")
                    a(f"v_mov_b32 v{Q + k}, v{S0 + k}")
                a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - 2 * K)}")
                a(f"s_sub_u32 s{sH}, s{sH}, {K}")
            elif la == "S" or rb == "S":
                m0_stack(MODE["SRC0"], -K)
                bank = T if la == "S" else Q
                for k in range(K):
                    a(f"v_mov_b32 v{bank + k}, v{S0 + k}")
                a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
            else:
                a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
                a(f"s_add_u32 s{sH}, s{sH}, {K}")
            a("s_mov_b32 m0, 0")
            for k in range(K):
                if la == "V":
                    a(f"v_mov_b32 v{T + k}, v{cur + k}")
                elif la == "C":
                    a(f"v_mov_b32 v{T + k}, s{sA}")
                if rb == "V" and form != "VV":
                    a(f"v_mov_b32 v{Q + k}, v{cur + k}")
                elif rb == "C":
                    a(f"v_mov_b32 v{Q + k}, s{sA}")
            if fl == 1:
                a(f"s_add_u32 s{T2}, s{T2}, {len(GBIN)}")     # (the second half of the body table: this flavour's copies)
            a(f"s_branch {lab('gbin_dispatch')}")
        # ---- generic unary stubs: operand -> T bank, result slot in sDST, body named by aux
        begin("gun_S", fl)
        entry()
        read_aux(T2)
        prefetch(nxt)
        m0_stack(MODE["SRC0"], -K)
        for k in range(K):
            a(f"v_mov_b32 v{T + k}, v{S0 + k}")
        a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
        a("s_mov_b32 m0, 0")
        a(f"s_branch {lab('gun_dispatch')}")
        begin("gun_V", fl)
        entry()
        read_aux(T2)
        prefetch(nxt)
        wait_cur()
        for k in range(K):
            a(f"v_mov_b32 v{T + k}, v{cur + k}")
        a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
        a(f"s_add_u32 s{sH}, s{sH}, {K}")
        a(f"s_branch {lab('gun_dispatch')}")
        # ---- IF (forward.cu:213-224): a > 0 ? b : c with all three operands on the stack (a on top); leaf operands are
        # pushed by the compiler, this function is too rare for 26 fused forms
        begin("if_sss", fl)
        entry()
        prefetch(nxt)
        m0_stack(MODE["SRC0"] | MODE["SRC1"] | MODE["DST"], -3 * K)
        for k in range(K):
            a(f"v_cmp_lt_f32 vcc, 0, v{S0 + 2 * K + k}")
            a(f"v_cndmask_b32 v{S0 + k}, v{S0 + k}, v{S0 + K + k}, vcc")
        a(f"s_sub_u32 s{sH}, s{sH}, {2 * K}")
        epilogue()
        # ---- multi-output programs (forward.cu:237-243).  The first out_len entries of the operand stack are the output
        # accumulators; mo_begin (aux = K * out_len) clears them and starts the stack above them, acc_s (aux = K * output
        # index) adds the top of the stack to one of them, end_mo folds this tile's errors of all outputs.
        begin("acc_s", fl)
        entry()
        read_aux(T2)
        prefetch(nxt)
        m0_stack(MODE["SRC0"] | MODE["SRC1"], -K)
        rows_mov(T, S0)
        a(f"s_add_u32 m0, s{T2}, {hex((MODE['SRC0'] | MODE['DST']) << 12)}")
        rows_op("add", S0, S0, T)
        a(f"s_sub_u32 s{sH}, s{sH}, {K}")
        epilogue()
        begin("mo_begin", fl)
        entry()
        read_aux(T2)
        prefetch(nxt)
        a(f"s_mov_b32 s{T1}, 0")
        a(f"{lab(f'mobegin_loop{fl}')}:")
        a(f"s_add_u32 m0, s{T1}, {hex(MODE['DST'] << 12)}")
        if PKARITH and K >= 2:
            for k in range(0, K, 2):
                a(f"v_pk_mov_b32 v[{S0 + k}:{S0 + k + 1}], 0, 0")
        else:
            for k in range(K):
                a(f"v_mov_b32 v{S0 + k}, 0")
        a(f"s_add_u32 s{T1}, s{T1}, {K}")
        a(f"s_cmp_lt_u32 s{T1}, s{T2}")
        a(f"s_cbranch_scc1 {lab(f'mobegin_loop{fl}')}")
        a(f"s_mov_b32 s{sH}, s{T2}")
        epilogue()
        begin("end_mo", fl)
        read_aux(T2)
        a(f"s_branch {lab('endmo_body')}")
        # ---- in-place divisions: the compiler picks these where one more stack entry is free above the operands
        def divip_stubs(fl=fl, nxt=nxt):
            for form in DIVIP:
                begin(f"divip_{form}", fl)
                entry()
                if form == "CS":
                    a(f"s_movrels_b32 s{sA}, s{W + 1}")
                if form == "SC":
                    a(f"s_movrels_b32 s{sBop}, s{W + 1}")
                prefetch(nxt)
                a(f"s_branch {lab(f'divip_body_{form}{fl}')}")
        divip_stubs()
        # ---- end of a CLASSIFIER program (no counterpart in forward.cu: the reference's Classification problem computes
        # batch_forward + soft-max + arg-max + compare in torch, classification.py:62-75): the class labels of the tile's rows
        # were prefetched into this flavour's bank like the regression labels of END; they move to the T bank for the shared body
        begin("end_cls", fl)
        read_aux(T2)
        a("s_waitcnt lgkmcnt(0)")
        for k in range(K):
            a(f"v_mov_b32 v{T + k}, v{P[fl] + k}")
        a(f"s_branch {lab('endcls_body')}")
        # SWAP: exchange the two top entries of the operand stack.  The general compiler puts it in front of a non-commutative function
        # whose operands it had evaluated in the other order (the larger subtree first: sr_tc.hip, compile_general's reordering pass)
        begin("swap", fl)
        entry()
        prefetch(P[1 - fl])
        m0_stack(MODE["SRC0"] | MODE["SRC1"], -2 * K)
        rows_mov(T, S0 + K)
        m0_stack(MODE["SRC0"] | MODE["SRC1"] | MODE["DST"], -2 * K)
        rows_mov(S0 + K, S0)
        m0_stack(MODE["DST"], -2 * K)
        rows_mov(S0, T)
        epilogue()
        # x / v with the reciprocal of v read instead of computed.  A launch in the SHORT / FAST division modes may stage a second copy of
        # the variables' columns behind the labels (sr_tc.hip, tc_stage_dataset; TcCompileParams::recip): when EVERY column of the
        # dataset lies in [2^-46, 2^46] (NaN allowed) it holds the correctly rounded reciprocals and flags bit 11 is set.  The compiler
        # then names a division by a variable divr_* and puts the KiB index of the variable's reciprocal column into the word's aux
        # field; everything else in the word is the division's.  The handler is the trusted-variable division (range test of a
        # stack numerator, quotient estimate, one correction) without its v_rcp_f32 -- the eight of them were half of such a handler's
        # vector clocks.  Without bit 11 (or in a build without the range-tested rows) the word runs through the division's own handler.
        def divr_stubs(fl=fl, nxt=nxt):
            for form in DIVR:
                begin(f"divr_{form}", fl)
                if not (use_range and TRUST):
                    a(f"s_branch {lab(f'h{fl}_div_{form}')}")
                    continue
                a("s_bitcmp1_b32 s17, 11")
                a(f"s_cbranch_scc0 {lab(f'h{fl}_div_{form}')}")
                read_aux(T2)
                entry()
                a(f"v_lshl_add_u32 v5, s{T2}, 10, v2")
                read_bank(Q, 5)                                   # the reciprocals of the divisor's rows
                if form == "VV":
                    a(f"s_movrels_b32 s{sBop}, s{W + 1}")
                    a(f"v_lshl_add_u32 v5, s{sBop}, 4, v2")
                    read_bank(T, 5)                               # the divisor's rows
                if form == "CV":   # a constant numerator outside [2^-46, 2^46] (0 / v among them): the division's own handler
                    a(f"s_movrels_b32 s{sA}, s{W + 1}")
                    a(f"s_and_b32 s{T2}, s{sA}, 0x7fffffff")
                    a(f"s_sub_u32 s{T2}, s{T2}, {hex(DIV_LO)}")
                    a(f"s_cmp_gt_u32 s{T2}, {hex(DIV_HI - DIV_LO)}")
                    a(f"s_cbranch_scc1 {lab(f'h{fl}_div_{form}')}")
                prefetch(nxt)
                wait_cur()
                if form == "SV":
                    m0_stack(MODE["SRC0"] | MODE["SRC1"], -K)
                    rows_mov(T, S0)
                    a(f"s_add_u32 s{sDST}, s{sH}, {hex((MODE['DST'] << 12) - K)}")
                    a("s_mov_b32 m0, 0")
                else:
                    a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
                    a(f"s_add_u32 s{sH}, s{sH}, {K}")
                a(f"s_branch {lab(f'divr_body_{form}{fl}')}")
        divr_stubs()
        # ---- the twins that do not prefetch (NOPF_TWINS)
        twin[0] = True
        for op in ("add", "sub", "mul"):
            for form in FORMS:
                arith(op, form, fl)
        push_c()
        push_v()
        divip_stubs()
        divr_stubs()
        twin[0] = False
    a(f".org {lab('hbase')}+{SLOT * 2 * NHF}")

    # In-place division bodies.  The gather forms above copy the operands into fixed banks because the division's temporaries
    # are fixed registers while stack operands are M0-relative -- 16 moves for S / S.  Here EVERYTHING is relative to one base
    # (M0 = all four REL bits | H - operands): the operands where they are, the five temporaries in the registers above
    # them (the compiler guarantees that entry exists), a constant operand straight from its SGPR, the quotient written over
    # the lower operand's slot row by row.  61 instead of 78 VALU instructions for S / S.  A block with a zero divisor (the
    # reference's NaN rule, forward.cu:183-187) goes to the gather form, which handles it.
    ALLREL = MODE["SRC0"] | MODE["SRC1"] | MODE["SRC2"] | MODE["DST"]
    for fl in (0, 1):
        for form in DIVIP:
            a(f"{lab(f'divip_body_{form}{fl}')}:")
            nops = 2 if form == "SS" else 1          # stack operands
            m0_stack(ALLREL, -nops * K)
            tb = S0 + nops * K                        # first register above the operands
            tmp = [tb + i for i in range(DIVIP_REGS)]
            q, acc = tmp[4], tmp[0]                   # the quotient replaces the residual; the zero test is over before the rows start
            if form == "SS":  # the first operand (the numerator) is on top of the stack, the divisor below it
                xs, ys = [S0 + K + k for k in range(K)], [S0 + k for k in range(K)]
            elif form == "CS":
                xs, ys = [f"s{sA}"] * K, [S0 + k for k in range(K)]
            else:
                xs, ys = [S0 + k for k in range(K)], [f"s{sBop}"] * K
            out = [S0 + k for k in range(K)]
            if use_range:
                slow, gen_, xz, yz = (lab(f"divip_{n}_{form}{fl}") for n in ("slow", "gen", "xz", "yz"))
                cx, cy = (f"s{sA}" if form == "CS" else None), (f"s{sBop}" if form == "SC" else None)
                if form != "SS":
                    a(f"s_and_b32 s{T1}, s{sA}, 0x7fffffff")
                if form == "CS":  # 0 / y needs no look at y
                    a(f"s_cmp_eq_u32 s{T1}, 0")
                    a(f"s_cbranch_scc1 {xz}")

                def const_range():  # the constant operand: one scalar test
                    a(f"s_sub_u32 s{T1}, s{T1}, {hex(DIV_LO)}")
                    a(f"s_cmp_gt_u32 s{T1}, {hex(DIV_HI - DIV_LO)}")
                    a(f"s_cbranch_scc1 {gen_}")

                st = range_test([] if cx else xs, [] if cy else ys, tmp, slow, between=const_range if form != "SS" else None)
                fast_pair_rows(cx or xs, cy or ys, tmp if DIVABREAST and K >= 8 else tmp[:6], out)
                if form == "SS":
                    a(f"s_sub_u32 s{sH}, s{sH}, {K}")
                epilogue()
                for kind, l in (("x", xz), ("y", yz)):
                    if kind == "y" and form == "SC":
                        continue  # (the compiler turns a division by the constant 0 into a multiplication by NaN)
                    a(f"{l}:")
                    zero_rows(kind, cx or xs, cy or ys, out, relative=True)
                    if form == "SS":
                        a(f"s_sub_u32 s{sH}, s{sH}, {K}")
                    epilogue()
                a(f"{slow}:")
                zero_blocks(st, xz, yz)
                a(f"{gen_}:")
                if form != "SC":
                    a(f"v_cmp_eq_f32 vcc, 0, v{st['loy']}")
                    a(f"s_cbranch_vccnz {lab(f'divold_{form}{fl}')}")
            elif form != "SC":  # (the compiler turns a division by the constant 0 into a multiplication by NaN)
                if K >= 3:
                    a(f"v_min3_f32 v{acc}, |v{ys[0]}|, |v{ys[1]}|, |v{ys[2]}|")
                    rest = ys[3:]
                else:
                    a(f"v_and_b32 v{acc}, 0x7fffffff, v{ys[0]}")
                    rest = ys[1:]
                while len(rest) >= 2:
                    a(f"v_min3_f32 v{acc}, v{acc}, |v{rest[0]}|, |v{rest[1]}|")
                    rest = rest[2:]
                if rest:
                    a(f"v_min_f32 v{acc}, v{acc}, |v{rest[0]}|")
                a(f"v_cmp_eq_f32 vcc, 0, v{acc}")
                a(f"s_cbranch_vccnz {lab(f'divold_{form}{fl}')}")
            for k in range(K):
                div_rows([xs[k]], [ys[k]], [q], nanfix=False, temps=tmp[:5])
                a(f"v_div_fixup_f32 v{S0 + k}, v{q}, {opnd(ys[k])}, {opnd(xs[k])}")
            if form == "SS":
                a(f"s_sub_u32 s{sH}, s{sH}, {K}")
            epilogue()

    # shared division bodies: K rows, then the scatter through v_div_fixup with an indexed destination
    for fl in (0, 1):
        for kind in ("Tc", "cT"):
            x, y = (T, P[fl]) if kind == "Tc" else (P[fl], T)
            # `b == 0 ? NaN : a / b` (forward.cu:183-187): ONE test for a zero divisor anywhere in the K x 64 block -- the
            # smallest |b| -- instead of a compare and a select per row; only a block with a zero takes the rows that
            # turn the numerator into NaN.  (A NaN divisor drops out of the minimum and makes its quotient NaN anyway.)
            ys_ = [y + k for k in range(K)]
            acc = DT[0]
            xs_ = [x + k for k in range(K)]
            out = [S0 + k for k in range(K)]
            TT = [18, 19, 20, 21, 22, 23]
            QT = [Q + i for i in range(6)]   # the quotient bank is free where the rows write their results themselves
            if use_range:
                slow, slow_t, xz, yz, rows, gen_, nz = (lab(f"div{n}_{kind}{fl}") for n in ("slow", "slowt", "xz", "yz", "rows", "gen", "nz"))
                if TRUST:
                    # entry for a TRUSTED variable in the current bank (its column is in range, div_stub): only the other operand,
                    # the copy of a stack entry in T, is tested
                    a(f"{lab(f'divbody_{kind}{fl}_t')}:")
                    st_t = range_test(xs_ if kind == "Tc" else [], ys_ if kind == "cT" else [], TT, slow_t)
                    a(f"s_branch {rows}")
                a(f"{lab(f'divbody_{kind}{fl}')}:")
                st = range_test(xs_, ys_, TT, slow)
                a(f"{rows}:")
                fast_pair_rows(xs_, ys_, TT + (QT if DIVABREAST and K >= 8 else []), out, out_m0=sDST)
                epilogue()
                for kd, l in (("x", xz), ("y", yz)):
                    a(f"{l}:")
                    a(f"s_mov_b32 m0, s{sDST}")
                    zero_rows(kd, xs_, ys_, out, relative=False)
                    epilogue()
                if TRUST:
                    a(f"{slow_t}:")
                    zero_blocks(st_t, xz, yz)
                    a(f"s_branch {nz if kind == 'Tc' else gen_}")  # a trusted divisor holds no zero
                a(f"{slow}:")
                zero_blocks(st, xz, yz)
                a(f"{gen_}:")
                a(f"v_cmp_eq_f32 vcc, 0, v{st['loy']}")
            else:
                a(f"{lab(f'divbody_{kind}{fl}')}:")
                if K >= 3:
                    a(f"v_min3_f32 v{acc}, |v{ys_[0]}|, |v{ys_[1]}|, |v{ys_[2]}|")
                    rest = ys_[3:]
                else:
                    a(f"v_and_b32 v{acc}, 0x7fffffff, v{ys_[0]}")
                    rest = ys_[1:]
                while len(rest) >= 2:
                    a(f"v_min3_f32 v{acc}, v{acc}, |v{rest[0]}|, |v{rest[1]}|")
                    rest = rest[2:]
                if rest:
                    a(f"v_min_f32 v{acc}, v{acc}, |v{rest[0]}|")
                a(f"v_cmp_eq_f32 vcc, 0, v{acc}")
            a(f"s_cbranch_vccnz {lab(f'divzero_{kind}{fl}')}")
            if use_range:
                a(f"{nz}:")
            for variant in (False, True):
                if variant:
                    a(f"{lab(f'divzero_{kind}{fl}')}:")
                div_rows([x + k for k in range(K)], ys_, [Q + k for k in range(K)], nanfix=variant)
                a(f"s_mov_b32 m0, s{sDST}")
                for k in range(K):
                    a(f"v_div_fixup_f32 v{S0 + k}, v{Q + k}, v{y + k}, v{x + k}")
                epilogue()

    # c / v and v / w with trusted variables: no gather, no test.  The operands are read where they are (the constant from its SGPR,
    # variables from their banks), the quotient goes straight to the new stack entry.
    if use_range and TRUST:
        TT = [18, 19, 20, 21, 22, 23]
        QT = [Q + i for i in range(6)]
        out = [S0 + k for k in range(K)]

        def push_dst():
            a(f"s_add_u32 s{sDST}, s{sH}, {hex(MODE['DST'] << 12)}")
            a(f"s_add_u32 s{sH}, s{sH}, {K}")

        for fl in (0, 1):
            cur = [P[fl] + k for k in range(K)]
            old_, rows, test, xz = (lab(f"divv_{n}_CV{fl}") for n in ("old", "rows", "test", "xz"))
            a(f"{lab(f'divv_CV{fl}')}:")
            a(f"s_and_b32 s{T2}, s{sA}, 0x7fffffff")
            a(f"s_cmp_eq_u32 s{T2}, 0")
            a(f"s_cbranch_scc1 {xz}")
            a(f"s_sub_u32 s{T2}, s{T2}, {hex(DIV_LO)}")
            a(f"s_cmp_gt_u32 s{T2}, {hex(DIV_HI - DIV_LO)}")
            a(f"s_cbranch_scc1 {old_}")
            trust_test(T1, 24, test)
            wait_cur()
            a(f"{rows}:")
            push_dst()
            fast_pair_rows(f"s{sA}", cur, TT + (QT if DIVABREAST and K >= 8 else []), out, out_m0=sDST)
            epilogue()
            a(f"{test}:")  # a variable with a value out of range somewhere: this block's values decide
            wait_cur()
            minmax(cur, TT[2], TT[3])
            a(f"v_cmp_gt_f32 vcc, {hex(DIV_LO)}, v{TT[2]}")
            a(f"s_cbranch_vccnz {old_}")
            a(f"v_cmp_lt_f32 vcc, {hex(DIV_HI)}, v{TT[3]}")
            a(f"s_cbranch_vccnz {old_}")
            a(f"s_branch {rows}")
            a(f"{xz}:")    # 0 / v
            wait_cur()
            push_dst()
            a(f"s_mov_b32 m0, s{sDST}")
            zero_rows("x", f"s{sA}", cur, out, relative=False)
            epilogue()
            # v / w
            old_ = lab(f"divv_old_VV{fl}")
            a(f"{lab(f'divv_VV{fl}')}:")
            trust_test(T1, 24, old_)
            a(f"s_mov_b32 s{T1}, s{sBop}")
            trust_test(T1, 6, old_)               # the second variable: its operand word is its LDS offset / 16
            wait_cur()
            push_dst()
            fast_pair_rows(cur, [T + k for k in range(K)], TT + (QT if DIVABREAST and K >= 8 else []), out, out_m0=sDST)
            epilogue()

    # divr_* (see the handlers): the rows of a division whose divisor's reciprocals are in the Q bank
    if use_range and TRUST:
        def recip_pair_rows(xs, ys, out):
            """fast_pair_rows with the reciprocals read from the Q bank; two row pairs abreast (quotient estimates in v18-v19 / v22-v23,
            residuals in v20-v21 / v4-v5)"""
            sets = [(18, 20), (22, 4)]
            pairs = list(range(0, K, 2))
            for g0 in range(0, len(pairs), len(sets)):
                grp = list(zip(pairs[g0:g0 + len(sets)], sets))
                for k, (q, e) in grp:
                    xp, xh = pk(xs, k)
                    a(f"v_pk_mul_f32 v[{q}:{q + 1}], {xp}, v[{Q + k}:{Q + k + 1}] op_sel_hi:[{xh},1]")
                for k, (q, e) in grp:
                    xp, xh = pk(xs, k)
                    yp, yh = pk(ys, k)
                    a(f"v_pk_fma_f32 v[{e}:{e + 1}], {yp}, v[{q}:{q + 1}], {xp} op_sel_hi:[{yh},1,{xh}] neg_lo:[1,0,0] neg_hi:[1,0,0]")
                a(f"s_mov_b32 m0, s{sDST}")
                for k, (q, e) in grp:
                    a(f"v_pk_fma_f32 v[{out[k]}:{out[k] + 1}], v[{e}:{e + 1}], v[{Q + k}:{Q + k + 1}], v[{q}:{q + 1}]")
                if g0 + len(sets) < len(pairs):
                    a("s_mov_b32 m0, 0")

        for fl in (0, 1):
            cur = [P[fl] + k for k in range(K)]
            tb = [T + k for k in range(K)]
            out = [S0 + k for k in range(K)]
            a(f"{lab(f'divr_body_SV{fl}')}:")                  # the numerator (a copy of the stack's top entry in T) is tested; a block
            range_test(tb, [], [18, 19, 20, 21, 22, 23], lab(f"divbody_Tc{fl}"))   # that fails goes to the division's own body
            recip_pair_rows(tb, cur, out)
            epilogue()
            a(f"{lab(f'divr_body_CV{fl}')}:")                  # (the compiler checked the constant's range)
            recip_pair_rows(f"s{sA}", cur, out)
            epilogue()
            a(f"{lab(f'divr_body_VV{fl}')}:")
            recip_pair_rows(cur, tb, out)
            epilogue()

    # ---- library sequences run row by row (pow, sinh, cosh: 120-190 instructions and 14-24 registers each; K unrolled
    # copies would not fit anywhere).  The row loop reads its operands and writes its result through M0-relative
    # addressing; the sequence itself is gen/ocml_bodies.py with its placeholders bound to registers that are free inside a
    # handler: v4 v5 v9 v18-v22 and the top HEAVY_REGS registers of the operand stack, which the program compiler leaves
    # unused wherever such a function occurs.  The sequences want more SGPRs than the interpreter has to spare: sixteen control
    # registers that no handler needs (the kernel's pointers and sizes, s8-s19; the batch bookkeeping, s26-s29) wait in the
    # lanes of one VGPR meanwhile.
    top = NV - HEAVY_REGS - (K if stats else 0)
    assert top % 2 == 0 and top >= S0
    VPAIRS = [4, 18, 20] + list(range(top, top + HEAVY_REGS, 2))
    VSINGLES = [9, 22]
    SPILL = list(range(8, 20)) + list(range(26, 30))
    SPAIRS = [T1, sA] + list(range(8, 20, 2)) + [26, 28]

    def bind(body, extra_v=0):
        """placeholder -> register; returns (format function, register map, extra VGPRs, spill VGPR, loop counter SGPR)"""
        vp, vs, sp = list(VPAIRS), list(VSINGLES), list(SPAIRS)
        vmap, smap = {}, {}
        for lo in body["vpairs"]:
            r = vp.pop(0); vmap[lo], vmap[lo + 1] = r, r + 1
        for lo in body["spairs"]:
            r = sp.pop(0); smap[lo], smap[lo + 1] = r, r + 1

        def single(pool_s, pool_p):
            if not pool_s:
                r = pool_p.pop(0); pool_s += [r, r + 1]
            return pool_s.pop(0)
        ss = []
        for r in body["vregs"]:
            if r not in vmap:
                vmap[r] = single(vs, vp)
        for r in body["sregs"]:
            if r not in smap:
                smap[r] = single(ss, sp)
        extra = [single(vs, vp) for _ in range(extra_v)]
        spill = single(vs, vp)
        counter = single(ss, sp)

        def fmt(line):
            import re

            def rep(m):
                kind, lo, pair = m.group(1), int(m.group(2)), m.group(3)
                mp = vmap if kind == "v" else smap
                return f"{kind}[{mp[lo]}:{mp[lo] + 1}]" if pair else f"{kind}{mp[lo]}"
            return re.sub(r"\{([vs])(\d+)(_\d+)?\}", rep, line)
        return fmt, vmap, extra, spill, counter

    def row_loop(label, body, nin, out_bank, pre=(), post=(), extra_v=0):
        """operands T[k] (and Q[k]) -> the sequence -> out_bank[k], for k = 0 .. K-1"""
        fmt, vmap, extra, spill, counter = bind(body, extra_v)
        for i, r in enumerate(SPILL):
            a(f"v_writelane_b32 v{spill}, s{r}, {i}")
        a(f"s_mov_b32 s{counter}, 0")
        a(f"{lab(label + '_row')}:")
        a(f"s_add_u32 m0, s{counter}, {hex(MODE['SRC0'] << 12)}")
        a(f"v_mov_b32 v{vmap[body['inputs'][0]]}, v{T}")
        if nin == 2:
            a(f"v_mov_b32 v{vmap[body['inputs'][1]]}, v{Q}")
        a("s_mov_b32 m0, 0")
        ctx = dict(a=vmap[body["inputs"][0]], b=vmap[body["inputs"][1]] if nin == 2 else None, o=vmap[body["output"]], x=extra)
        for ln in pre:
            a(ln(ctx))
        for ln in body["lines"]:
            a(fmt(ln))
        for ln in post:
            a(ln(ctx))
        a(f"s_add_u32 m0, s{counter}, {hex(MODE['DST'] << 12)}")
        a(f"v_mov_b32 v{out_bank}, v{vmap[body['output']]}")
        a(f"s_add_u32 s{counter}, s{counter}, 1")
        a(f"s_cmp_lt_u32 s{counter}, {K}")
        a(f"s_cbranch_scc1 {lab(label + '_row')}")
        a("s_mov_b32 m0, 0")
        for i, r in enumerate(SPILL):
            a(f"v_readlane_b32 s{r}, v{spill}, {i}")
    def row_pair_loop(label, body, nin, out_bank, vpool):
        """the sequence over row PAIRS (gen/pair_rows.py): operands T[k], T[k + 1] (and Q) -> two rows of the sequence with its
        multiplications, additions and fused multiply-adds packed -> out_bank[k], out_bank[k + 1].  `vpool`: the aligned VGPR pairs it
        may use.  Returns False (nothing emitted) when the registers do not suffice -- the caller then runs the rows one by one."""
        if not (LIBPK and K >= 2):
            return False
        # (ten more control registers wait in the spill register's lanes: jump target, J, H, the result slot, the tile, the mask of
        # evaluated trees, the record address)
        spill = SPILL + [20, 21, 22, 23, 24, 25, 30, 31, 32, 33]
        counter = 24
        try:
            res = Pairing(body, label).run().allocate(vpool, SPAIRS + [20, 22, 30, 32], vsingles=[22, 23])
        except AssertionError:
            return False
        check_pairing(body, res, label)                # (symbolic execution: both rows equal the library's sequence)
        spill_v = 9
        for i, r in enumerate(spill):
            a(f"v_writelane_b32 v{spill_v}, s{r}, {i}")
        a(f"s_mov_b32 s{counter}, 0")
        a(f"{lab(label + '_row')}:")
        a(f"s_add_u32 m0, s{counter}, {hex(MODE['SRC0'] << 12)}")
        for j in (0, 1):
            a(f"v_mov_b32 v{res['inputs'][0] + j}, v{T + j}")
            if nin == 2:
                a(f"v_mov_b32 v{res['inputs'][1] + j}, v{Q + j}")
        a("s_mov_b32 m0, 0")
        for ln in res["lines"]:
            a(ln)
        a(f"s_add_u32 m0, s{counter}, {hex(MODE['DST'] << 12)}")
        for j in (0, 1):
            a(f"v_mov_b32 v{out_bank + j}, v{res['output'] + j}")
        a(f"s_add_u32 s{counter}, s{counter}, 2")
        a(f"s_cmp_lt_u32 s{counter}, {K}")
        a(f"s_cbranch_scc1 {lab(label + '_row')}")
        a("s_mov_b32 m0, 0")
        for i, r in enumerate(spill):
            a(f"v_readlane_b32 s{r}, v{spill_v}, {i}")
        return True
    # ---- bodies behind the generic stubs.  Binary: a in T, b in Q, result left in T; unary: operand in T, result in Q.
    a(f"{lab('gbin_dispatch')}:")
    a(f"s_lshl_b32 s{T2}, s{T2}, 2")
    a(f"s_getpc_b64 s[{sA}:{sA + 1}]")
    a(f"{lab('gbin_pc')}:")
    a(f"s_add_u32 s{sA}, s{sA}, s{T2}")
    a(f"s_addc_u32 s{sA + 1}, s{sA + 1}, 0")
    a(f"s_add_u32 s{sA}, s{sA}, {lab('gbin_table')}-{lab('gbin_pc')}")
    a(f"s_addc_u32 s{sA + 1}, s{sA + 1}, 0")
    a(f"s_setpc_b64 s[{sA}:{sA + 1}]")
    a(f"{lab('gbin_table')}:")
    for fl_ in (0, 1):   # (the stubs of flavour 1 add len(GBIN) to the body number: pow over row pairs borrows the flavour's own operand bank)
        for body in GBIN:
            a(f"s_branch {lab('gbody_' + body + (f'_fl{fl_}' if body == 'pow' else ''))}")
    a(f"{lab('gun_dispatch')}:")
    a(f"s_lshl_b32 s{T2}, s{T2}, 2")
    a(f"s_getpc_b64 s[{sA}:{sA + 1}]")
    a(f"{lab('gun_pc')}:")
    a(f"s_add_u32 s{sA}, s{sA}, s{T2}")
    a(f"s_addc_u32 s{sA + 1}, s{sA + 1}, 0")
    a(f"s_add_u32 s{sA}, s{sA}, {lab('gun_table')}-{lab('gun_pc')}")
    a(f"s_addc_u32 s{sA + 1}, s{sA + 1}, 0")
    a(f"s_setpc_b64 s[{sA}:{sA + 1}]")
    a(f"{lab('gun_table')}:")
    for body in GUN:
        a(f"s_branch {lab('ubody_' + body)}")
    for body in GBIN:
        a(f"{lab('gbody_' + body)}:")
        if body in ("max", "min"):      # forward.cu:201-204: a >= b ? a : b  /  a <= b ? a : b  (a NaN operand selects b)
            cmp = {"max": "ge", "min": "le"}[body]
            for k in range(K):
                a(f"v_cmp_{cmp}_f32 vcc, v{T + k}, v{Q + k}")
                a(f"v_cndmask_b32 v{T + k}, v{Q + k}, v{T + k}, vcc")
        elif body in ("lt", "gt", "le", "ge"):   # forward.cu:205-212: 1 or -1
            for k in range(K):
                a(f"v_cmp_{body}_f32 vcc, v{T + k}, v{Q + k}")
                a(f"v_cndmask_b32_e64 v{T + k}, -1.0, 1.0, vcc")
        elif body == "ldiv":            # forward.cu:188-192: |b| <= DELTA -> b = copysign(DELTA, b); a / b (never a zero divisor)
            a(f"s_mov_b32 s{T1}, 0x3089705f")        # DELTA = 1e-9f
            a(f"s_brev_b32 s{T2}, -2")               # 0x7fffffff
            a(f"v_mov_b32 v9, s{T1}")
            for k in range(K):
                a(f"v_cmp_le_f32_e64 vcc, |v{Q + k}|, s{T1}")
                a(f"v_bfi_b32 v5, s{T2}, v9, v{Q + k}")
                a(f"v_cndmask_b32 v{Q + k}, v{Q + k}, v5, vcc")
            for k in range(K):
                div_rows([T + k], [Q + k], [4], nanfix=False)
                a(f"v_div_fixup_f32 v{T + k}, v4, v{Q + k}, v{T + k}")
        elif body == "pow":             # forward.cu:193-194
            # (over row pairs the sequence needs 19 register pairs: the registers above the stack, the temporaries and the CURRENT
            # operand bank -- the stubs have copied its values to T / Q --, hence one copy of the body per flavour)
            base_pool = [4, 18, 20] + list(range(top, top + HEAVY_REGS, 2))
            a(f"{lab('gbody_pow_fl0')}:")
            if not row_pair_loop("pow_fl0", BODIES["pow"], 2, T, base_pool + list(range(P[0], P[0] + K - 1, 2))):
                row_loop("pow_fl0", BODIES["pow"], 2, T)
            a(f"s_branch {lab('gbin_tail')}")
            a(f"{lab('gbody_pow_fl1')}:")
            if not row_pair_loop("pow_fl1", BODIES["pow"], 2, T, base_pool + list(range(P[1], P[1] + K - 1, 2))):
                row_loop("pow_fl1", BODIES["pow"], 2, T)
        elif body == "lpow":            # forward.cu:195-200: (a == 0 && b == 0) ? 0 : pow(|a|, b)
            row_loop("lpow", BODIES["pow"], 2, T, extra_v=1,
                     pre=(lambda c: f"v_and_b32 v{c['a']}, 0x7fffffff, v{c['a']}",
                          lambda c: f"v_and_b32 v{c['x'][0]}, 0x7fffffff, v{c['b']}",
                          lambda c: f"v_or_b32 v{c['x'][0]}, v{c['x'][0]}, v{c['a']}"),   # zero exactly when both operands are +-0
                     post=(lambda c: f"v_cmp_eq_u32 vcc, 0, v{c['x'][0]}",
                           lambda c: "s_nop 1",
                           lambda c: f"v_cndmask_b32_e64 v{c['o']}, v{c['o']}, 0, vcc"))
        a(f"s_branch {lab('gbin_tail')}")
    a(f"{lab('gbin_tail')}:")
    a(f"s_mov_b32 m0, s{sDST}")
    for k in range(K):
        a(f"v_mov_b32 v{S0 + k}, v{T + k}")
    epilogue()
    for body in GUN:
        a(f"{lab('ubody_' + body)}:")
        if body == "zero":              # a unary node whose function id is unknown yields 0 (forward.cu:117)
            for k in range(K):
                a(f"v_mov_b32 v{Q + k}, 0")
        elif body == "one":             # pow(x, 0), pow(1, y): 1 whatever the other operand is (sr_tc.hip pow_fold_kind)
            for k in range(K):
                a(f"v_mov_b32 v{Q + k}, 1.0")
        elif body == "rcp":             # pow(x, -1): the plain IEEE 1 / x (+-0 gives +-inf: not the reference's DIV rule)
            a("v_mov_b32 v9, 1.0")
            for k in range(K):
                div_rows([9], [T + k], [4], nanfix=False)
                a(f"v_div_fixup_f32 v{Q + k}, v4, v{T + k}, v9")
        elif body in ("sinh", "cosh"):  # forward.cu:129-134
            if not row_pair_loop(body, BODIES[body], 1, Q, [4, 18, 20] + list(range(top, top + HEAVY_REGS, 2))):
                row_loop(body, BODIES[body], 1, Q)
        elif body == "tanh":            # the library's tanhf: both of its branches, then the select it makes with the exec mask
            t2, t3, t4, t5 = 18, 19, 20, 21
            for k in range(K):
                x = T + k
                a(f"v_add_f32_e64 v{t2}, |v{x}|, |v{x}|")
                a(f"v_mul_f32 v{t3}, 0x3fb8aa3b, v{t2}")
                a(f"s_mov_b32 s{T1}, 0x3fb8aa3b")
                a(f"v_rndne_f32 v{t4}, v{t3}")
                a(f"v_sub_f32 v{t5}, v{t3}, v{t4}")
                a(f"v_fma_f32 v{t3}, v{t2}, s{T1}, -v{t3}")
                a(f"v_fmamk_f32 v{t3}, v{t2}, 0x32a5705f, v{t3}")
                a(f"v_add_f32 v{t3}, v{t5}, v{t3}")
                a(f"v_exp_f32 v{t3}, v{t3}")
                a(f"v_cvt_i32_f32 v{t4}, v{t4}")
                a(f"s_mov_b32 s{T1}, 0xc2ce8ed0")
                a(f"v_cmp_ngt_f32 vcc, s{T1}, v{t2}")
                a(f"s_mov_b32 s{T1}, 0x42b17218")
                a(f"v_ldexp_f32 v{t3}, v{t3}, v{t4}")
                a(f"v_cndmask_b32 v{t3}, 0, v{t3}, vcc")
                a(f"v_mov_b32 v{t4}, 0x7f800000")
                a(f"v_cmp_nlt_f32 vcc, s{T1}, v{t2}")
                a("s_nop 1")
                a(f"v_cndmask_b32 v{t2}, v{t4}, v{t3}, vcc")
                a(f"v_add_f32 v{t2}, 1.0, v{t2}")
                a(f"v_rcp_f32 v{t2}, v{t2}")
                a("s_nop 0")
                a(f"v_fma_f32 v{t2}, v{t2}, -2.0, 1.0")          # |x| >= 0.625: 1 - 2 / (exp(2|x|) + 1)
                a(f"v_mul_f32 v{t3}, v{x}, v{x}")
                a(f"v_mov_b32 v{t4}, 0x3ca908c9")
                a(f"v_fmac_f32 v{t4}, 0xbbbac73d, v{t3}")
                a(f"v_fmaak_f32 v{t4}, v{t3}, v{t4}, 0xbd5c1c4e")
                a(f"v_fmaak_f32 v{t4}, v{t3}, v{t4}, 0x3e088382")
                a(f"v_fmaak_f32 v{t4}, v{t3}, v{t4}, 0xbeaaaa99")
                a(f"v_mul_f32_e64 v{t4}, |v{x}|, v{t4}")
                a(f"v_fma_f32 v{t3}, v{t3}, v{t4}, |v{x}|")     # |x| < 0.625: polynomial
                a(f"s_mov_b32 s{T1}, 0x3f200000")
                a(f"v_cmp_nlt_f32_e64 vcc, |v{x}|, s{T1}")
                a("s_nop 1")
                a(f"v_cndmask_b32 v{t2}, v{t3}, v{t2}, vcc")
                a(f"s_brev_b32 s{T1}, -2")
                a(f"v_bfi_b32 v{Q + k}, s{T1}, v{t2}, v{x}")     # sign of x
        a(f"s_branch {lab('gun_tail')}")
    a(f"{lab('gun_tail')}:")
    a(f"s_mov_b32 m0, s{sDST}")
    for k in range(K):
        a(f"v_mov_b32 v{S0 + k}, v{Q + k}")
    epilogue()

    # ---- sin / cos / tan: the small-argument path of the device math library (|x| < 2^17), transcribed from the ISA that
    # hipcc emits for sinf / cosf / tanf (three-term Cody-Waite reduction by pi/2, the library's polynomials, quadrant
    # selection, NaN for non-finite operands).  A block with a finite operand of 2^17 or more would need the library's
    # Payne-Hanek reduction: the tree is handed to the register kernels instead (runtime bail-out below).
    tx, tn, tr, ts2, tp, tq, tu = 18, 19, 20, 21, 22, 9, 4   # per-row temporaries (v4 / v9 are free inside a body)
    for uop in ("sin", "cos", "tan"):
        if uop not in UNARY:
            continue
        a(f"{lab(f'trigbody_{uop}')}:")
        xs = [T + k for k in range(K)]
        if K >= 3:
            a(f"v_max3_f32 v{tx}, |v{xs[0]}|, |v{xs[1]}|, |v{xs[2]}|")
            rest = xs[3:]
        else:
            a(f"v_and_b32 v{tx}, 0x7fffffff, v{xs[0]}")
            rest = xs[1:]
        while len(rest) >= 2:
            a(f"v_max3_f32 v{tx}, v{tx}, |v{rest[0]}|, |v{rest[1]}|")
            rest = rest[2:]
        if rest:
            a(f"v_max_f32 v{tx}, v{tx}, |v{rest[0]}|")
        a(f"s_mov_b32 s{T1}, 0x48000000")             # 2^17
        a(f"v_cmp_le_f32 vcc, s{T1}, v{tx}")
        a(f"s_cbranch_vccnz {lab(f'triglib_{uop}')}")   # some row needs the library's Payne-Hanek reduction: the whole function, row by row (below)
        if TRIGPK and K >= 2:
            # Row PAIRS through packed instructions: the reduction, the polynomials and tan's correction steps are multiplications and
            # fused multiply-adds -- v_pk_mul_f32 / v_pk_fma_f32 do two rows for the issue slot of one (the same IEEE operations as the
            # library's v_fma / v_fmac / v_fmaak / v_fmamk, so the same bits); a constant comes from one SGPR for both halves
            # (op_sel_hi 0; packed instructions take no literal), an addend constant that meets an SGPR factor is two moves.  Rounding,
            # conversion, reciprocals and the quadrant logic stay per row.  39 / 35 / 42 instead of 50 / 46 / 54 vector instructions
            # per row pair of sin / cos / tan.
            A, B, C, D = 4, 18, 20, 22

            def pk_c(ins, dst, s0, s1, s2, cpos, const, neg=""):
                """packed instruction with the constant `const` at source position cpos (1 or 2)"""
                a(f"s_mov_b32 s{T1}, {const}")
                ops = [f"v[{s0}:{s0 + 1}]" if s0 is not None else None, f"v[{s1}:{s1 + 1}]" if s1 is not None else None,
                       f"v[{s2}:{s2 + 1}]" if s2 is not None else None]
                ops[cpos] = f"s[{T1}:{T2}]"
                ops = [o for o in ops if o is not None]
                sel = ["1"] * len(ops)
                sel[cpos] = "0"
                a(f"{ins} v[{dst}:{dst + 1}], {', '.join(ops)} op_sel_hi:[{','.join(sel)}]{neg}")

            for k in range(0, K, 2):
                X, E = T + k, Q + k
                for j in (0, 1):
                    a(f"v_and_b32 v{A + j}, 0x7fffffff, v{X + j}")
                pk_c("v_pk_mul_f32", B, A, None, None, 1, "0x3f22f983")           # |x| * 2 / pi
                for j in (0, 1):
                    a(f"v_rndne_f32 v{B + j}, v{B + j}")
                for j in (0, 1):
                    a(f"v_cvt_i32_f32 v{C + j}, v{B + j}")                          # quadrant
                pk_c("v_pk_fma_f32", A, B, None, A, 1, "0xbfc90fda")              # r = n * (-pi/2, three parts) + |x|
                pk_c("v_pk_fma_f32", A, B, None, A, 1, "0xb3a22168")
                pk_c("v_pk_fma_f32", A, B, None, A, 1, "0xa7c234c4")
                a(f"v_pk_mul_f32 v[{B}:{B + 1}], v[{A}:{A + 1}], v[{A}:{A + 1}]")    # s2 = r * r
                if uop in ("sin", "cos"):
                    for j in (0, 1):
                        a(f"v_mov_b32 v{D + j}, 0x3c0881c4")
                    pk_c("v_pk_fma_f32", D, B, None, D, 1, "0xb94c1982")
                    pk_c("v_pk_fma_f32", D, B, D, None, 2, "0xbe2aaa9d")
                    a(f"v_pk_mul_f32 v[{D}:{D + 1}], v[{B}:{B + 1}], v[{D}:{D + 1}]")
                    a(f"v_pk_fma_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{D}:{D + 1}], v[{A}:{A + 1}]")   # sin(r)
                    for j in (0, 1):
                        a(f"v_mov_b32 v{E + j}, 0xbab64f3b")
                    pk_c("v_pk_fma_f32", E, B, None, E, 1, "0x37d75334")
                    pk_c("v_pk_fma_f32", E, B, E, None, 2, "0x3d2aabf7")
                    pk_c("v_pk_fma_f32", E, B, E, None, 2, "0xbf000004")
                    pk_c("v_pk_fma_f32", E, B, E, None, 2, "0x3f800000")          # cos(r)
                    for j in (0, 1):
                        tr_, tq_, tu_, x_ = A + j, E + j, C + j, X + j
                        a(f"v_and_b32 v9, 1, v{tu_}")
                        a(f"v_lshlrev_b32 v{tu_}, 30, v{tu_}")
                        a("v_cmp_eq_u32 vcc, 0, v9")
                        a(f"v_and_b32 v{tu_}, 0x80000000, v{tu_}")               # sign from bit 1 of the quadrant
                        if uop == "sin":
                            a(f"v_and_b32 v9, 0x80000000, v{x_}")                # sin is odd: the sign of x
                            a(f"v_cndmask_b32 v{tq_}, v{tq_}, v{tr_}, vcc")      # even quadrant: sin(r), odd: cos(r)
                            a(f"v_xor_b32 v{tu_}, v9, v{tu_}")
                            a(f"v_xor_b32 v{tq_}, v{tu_}, v{tq_}")
                        else:
                            a(f"v_cndmask_b32_e64 v{tq_}, -v{tr_}, v{tq_}, vcc")  # even quadrant: cos(r), odd: -sin(r)
                            a(f"v_xor_b32 v{tq_}, v{tu_}, v{tq_}")
                else:
                    for j in (0, 1):
                        a(f"v_mov_b32 v{D + j}, 0xbf039337")
                    pk_c("v_pk_fma_f32", D, B, None, D, 1, "0x3c971480")
                    pk_c("v_pk_fma_f32", D, B, D, None, 2, "0x3f93f425")
                    for j in (0, 1):
                        a(f"v_rcp_f32 v{D + j}, v{D + j}")
                    for j in (0, 1):
                        a(f"v_mov_b32 v{E + j}, 0x3ec54587")
                    pk_c("v_pk_fma_f32", E, B, None, E, 1, "0xbc8cedd3")
                    for j in (0, 1):
                        a(f"v_and_b32 v{C + j}, 1, v{C + j}")
                    a(f"v_pk_mul_f32 v[{D}:{D + 1}], v[{E}:{E + 1}], v[{D}:{D + 1}]")
                    a(f"v_pk_mul_f32 v[{B}:{B + 1}], v[{B}:{B + 1}], v[{D}:{D + 1}]")                    # z
                    a(f"v_pk_fma_f32 v[{D}:{D + 1}], v[{B}:{B + 1}], v[{A}:{A + 1}], v[{A}:{A + 1}]")   # t = tan(r)
                    for j in (0, 1):
                        a(f"v_rcp_f32 v{E + j}, v{D + j}")
                    for j in (0, 1):
                        a(f"v_sub_f32 v9, v{D + j}, v{A + j}")
                        a(f"v_fma_f32 v{A + j}, v{B + j}, v{A + j}, -v9")
                    pk_c("v_pk_fma_f32", B, D, E, None, 2, "0x3f800000", " neg_lo:[0,1,0] neg_hi:[0,1,0]")
                    a(f"v_pk_fma_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{E}:{E + 1}], v[{B}:{B + 1}] neg_lo:[0,1,0] neg_hi:[0,1,0]")
                    a(f"v_pk_fma_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{E}:{E + 1}], v[{E}:{E + 1}] neg_lo:[0,1,1] neg_hi:[0,1,1]")   # -cot(r)
                    for j in (0, 1):
                        a(f"v_cmp_eq_u32 vcc, 0, v{C + j}")
                        a(f"v_and_b32 v9, 0x80000000, v{X + j}")                 # tan is odd
                        a(f"v_cndmask_b32 v{A + j}, v{A + j}, v{D + j}, vcc")
                        a(f"v_xor_b32 v{E + j}, v9, v{A + j}")
        else:
            for k in range(K):
                x, res = T + k, Q + k
                a(f"s_mov_b32 s{T1}, 0x3f22f983")         # 2 / pi
                a(f"v_mul_f32_e64 v{tn}, |v{x}|, s{T1}")
                a(f"v_rndne_f32 v{tn}, v{tn}")
                a(f"s_mov_b32 s{T1}, 0xbfc90fda")         # -pi/2, high part
                a(f"v_cvt_i32_f32 v{tu}, v{tn}")          # quadrant
                a(f"v_fma_f32 v{tr}, v{tn}, s{T1}, |v{x}|")
                a(f"v_fmamk_f32 v{tr}, v{tn}, 0xb3a22168, v{tr}")
                a(f"v_fmamk_f32 v{tr}, v{tn}, 0xa7c234c4, v{tr}")
                a(f"v_mul_f32 v{ts2}, v{tr}, v{tr}")
                if uop in ("sin", "cos"):
                    a(f"v_mov_b32 v{tp}, 0x3c0881c4")
                    a(f"v_fmac_f32 v{tp}, 0xb94c1982, v{ts2}")
                    a(f"v_fmaak_f32 v{tp}, v{ts2}, v{tp}, 0xbe2aaa9d")
                    a(f"v_mul_f32 v{tp}, v{ts2}, v{tp}")
                    a(f"v_fmac_f32 v{tr}, v{tr}, v{tp}")                 # sin(r)
                    a(f"v_mov_b32 v{tq}, 0xbab64f3b")
                    a(f"v_fmac_f32 v{tq}, 0x37d75334, v{ts2}")
                    a(f"v_fmaak_f32 v{tq}, v{ts2}, v{tq}, 0x3d2aabf7")
                    a(f"v_fmaak_f32 v{tq}, v{ts2}, v{tq}, 0xbf000004")
                    a(f"v_fma_f32 v{tq}, v{ts2}, v{tq}, 1.0")            # cos(r)
                    a(f"v_and_b32 v{tp}, 1, v{tu}")
                    a(f"v_lshlrev_b32 v{tu}, 30, v{tu}")
                    a(f"v_cmp_eq_u32 vcc, 0, v{tp}")
                    a(f"v_and_b32 v{tu}, 0x80000000, v{tu}")             # sign from bit 1 of the quadrant
                    if uop == "sin":
                        a(f"v_and_b32 v{tp}, 0x80000000, v{x}")          # sin is odd: the sign of x
                        a(f"v_cndmask_b32 v{tq}, v{tq}, v{tr}, vcc")     # even quadrant: sin(r), odd: cos(r)
                        a(f"v_xor_b32 v{tu}, v{tp}, v{tu}")
                        a(f"v_xor_b32 v{res}, v{tu}, v{tq}")
                    else:
                        a(f"v_cndmask_b32_e64 v{tq}, -v{tr}, v{tq}, vcc")  # even quadrant: cos(r), odd: -sin(r)
                        a(f"v_xor_b32 v{res}, v{tu}, v{tq}")
                else:
                    a(f"v_mov_b32 v{tp}, 0xbf039337")
                    a(f"v_fmac_f32 v{tp}, 0x3c971480, v{ts2}")
                    a(f"v_fmaak_f32 v{tp}, v{ts2}, v{tp}, 0x3f93f425")
                    a(f"v_rcp_f32 v{tp}, v{tp}")
                    a(f"v_mov_b32 v{tq}, 0x3ec54587")
                    a(f"v_fmac_f32 v{tq}, 0xbc8cedd3, v{ts2}")
                    a(f"v_and_b32 v{tu}, 1, v{tu}")
                    a(f"v_mul_f32 v{tp}, v{tq}, v{tp}")
                    a(f"v_mul_f32 v{ts2}, v{ts2}, v{tp}")                # z
                    a(f"v_fma_f32 v{tp}, v{ts2}, v{tr}, v{tr}")          # t = tan(r)
                    a(f"v_rcp_f32 v{tq}, v{tp}")
                    a(f"v_sub_f32 v{tn}, v{tp}, v{tr}")
                    a(f"v_fma_f32 v{tr}, v{ts2}, v{tr}, -v{tn}")
                    a(f"v_cmp_eq_u32 vcc, 0, v{tu}")
                    a(f"v_fma_f32 v{ts2}, v{tp}, -v{tq}, 1.0")
                    a(f"v_fma_f32 v{tr}, v{tr}, -v{tq}, v{ts2}")
                    a(f"v_fma_f32 v{tr}, v{tr}, -v{tq}, -v{tq}")         # -cot(r)
                    a(f"v_cndmask_b32 v{tr}, v{tr}, v{tp}, vcc")
                    a(f"v_and_b32 v{tu}, 0x80000000, v{x}")              # tan is odd
                    a(f"v_xor_b32 v{res}, v{tu}, v{tr}")
                # (the library's final "NaN for a non-finite operand" select is not needed: infinities never get past the
                # range test above, and a NaN operand makes every step of the sequence NaN)
        a(f"{lab(f'trigdone_{uop}')}:")
        a(f"s_mov_b32 m0, s{sDST}")
        for k in range(K):
            a(f"v_mov_b32 v{S0 + k}, v{Q + k}")
        epilogue()
    # ---- sqrt / exp / log: the device math library's sequences for sqrtf / expf / logf, transcribed like the trigonometric
    # ones (full range: no bail-out).  llog = the reference's LOOSE_LOG: log|x|, and -1e9 for x = 0 (forward.cu:139-144).
    t1, t2, t3, t4, t5 = 18, 19, 20, 21, 22
    for uop in ("sqrt", "exp", "log", "llog"):
        if uop not in UNARY:
            continue
        a(f"{lab(f'trigbody_{uop}')}:")
        if uop == "llog":
            a("v_mov_b32 v9, 0xce6e6b28")                           # -1e9
        if uop == "exp":
            a("v_mov_b32 v9, 0x7f800000")
        if LIBPK and K >= 2 and uop in ("exp", "log", "llog"):
            # Row PAIRS (round 5, as sin / cos / tan in round 4): the multiplications, fused multiply-adds and additions of the library's
            # sequence as v_pk_* over two rows -- the same IEEE operations, the same bits --, a constant from one SGPR for both halves;
            # rounding, conversion, v_exp / v_log, ldexp and the range selects stay per row.  21 instead of 30 vector instructions per pair.
            A, B, C, D = 4, 18, 20, 22

            def pkc(ins, dst, srcs, const, cpos, neg=None):
                """packed instruction, the constant `const` at source position cpos"""
                a(f"s_mov_b32 s{T1}, {const}")
                ops = [f"v[{r}:{r + 1}]" for r in srcs]
                ops.insert(cpos, f"s[{T1}:{T2}]")
                sel = ["1"] * len(ops)
                sel[cpos] = "0"
                tail = f" op_sel_hi:[{','.join(sel)}]"
                if neg:
                    tail += f" neg_lo:[{','.join(neg)}] neg_hi:[{','.join(neg)}]"
                a(f"{ins} v[{dst}:{dst + 1}], {', '.join(ops)}{tail}")

            for k in range(0, K, 2):
                X, R = T + k, Q + k
                if uop == "exp":
                    pkc("v_pk_mul_f32", A, [X], "0x3fb8aa3b", 1)                                  # t2 = x * log2(e)
                    pkc("v_pk_fma_f32", B, [X, A], "0x3fb8aa3b", 1, ["0", "0", "1"])              # t3 = x * log2(e) - t2
                    for j in (0, 1):
                        a(f"v_rndne_f32 v{C + j}, v{A + j}")
                    pkc("v_pk_fma_f32", B, [X, B], "0x32a5705f", 1)                               # t3 = x * (low part) + t3
                    a(f"v_pk_add_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{C}:{C + 1}] neg_lo:[0,1] neg_hi:[0,1]")
                    a(f"v_pk_add_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{B}:{B + 1}]")
                    for j in (0, 1):
                        a(f"v_cvt_i32_f32 v{C + j}, v{C + j}")
                    for j in (0, 1):
                        a(f"v_exp_f32 v{A + j}, v{A + j}")
                    a(f"s_mov_b32 s{T1}, 0xc2ce8ed0")
                    for j in (0, 1):
                        a(f"v_cmp_ngt_f32 vcc, s{T1}, v{X + j}")
                        a(f"v_ldexp_f32 v{A + j}, v{A + j}, v{C + j}")
                        a(f"v_cndmask_b32 v{A + j}, 0, v{A + j}, vcc")
                    a(f"s_mov_b32 s{T1}, 0x42b17218")
                    for j in (0, 1):
                        a(f"v_cmp_nlt_f32 vcc, s{T1}, v{X + j}")
                        a("s_nop 1")
                        a(f"v_cndmask_b32 v{R + j}, v9, v{A + j}, vcc")
                else:
                    a(f"s_mov_b32 s{T1}, 0x800000")
                    for j in (0, 1):
                        a(f"v_cmp_gt_f32 vcc, s{T1}, v{X + j}")                                    # denormal operand: scale by 2^32
                        a(f"v_cndmask_b32_e64 v{B + j}, 0, 32, vcc")
                        a(f"v_ldexp_f32 v{A + j}, v{X + j}, v{B + j}")
                        a(f"v_mov_b32 v{B + j}, 0x41b17218")
                        a(f"v_cndmask_b32 v{B + j}, 0, v{B + j}, vcc")
                    for j in (0, 1):
                        a(f"v_log_f32 v{A + j}, v{A + j}")
                    a("s_nop 1")
                    pkc("v_pk_mul_f32", C, [A], "0x3f317217", 1)
                    pkc("v_pk_fma_f32", D, [A, C], "0x3f317217", 1, ["0", "0", "1"])
                    pkc("v_pk_fma_f32", D, [A, D], "0x3377d1cf", 1)
                    a(f"v_pk_add_f32 v[{C}:{C + 1}], v[{C}:{C + 1}], v[{D}:{D + 1}]")
                    a(f"s_mov_b32 s{T1}, 0x7f800000")
                    for j in (0, 1):
                        a(f"v_cmp_lt_f32_e64 vcc, |v{A + j}|, s{T1}")
                        a("s_nop 1")
                        a(f"v_cndmask_b32 v{A + j}, v{A + j}, v{C + j}, vcc")
                    if uop == "log":
                        a(f"v_pk_add_f32 v[{R}:{R + 1}], v[{A}:{A + 1}], v[{B}:{B + 1}] neg_lo:[0,1] neg_hi:[0,1]")
                    else:
                        a(f"v_pk_add_f32 v[{A}:{A + 1}], v[{A}:{A + 1}], v[{B}:{B + 1}] neg_lo:[0,1] neg_hi:[0,1]")
                        for j in (0, 1):
                            a(f"v_cmp_eq_f32 vcc, 0, v{X + j}")
                            a("s_nop 1")
                            a(f"v_cndmask_b32 v{R + j}, v{A + j}, v9, vcc")                        # LOOSE_LOG(0) = -MAX_VAL
        for k in range(K if not (LIBPK and K >= 2 and uop in ("exp", "log", "llog")) else 0):
            x, res = T + k, Q + k
            if uop == "sqrt":
                a(f"v_mul_f32 v{t2}, 0x4f800000, v{x}")
                a(f"s_mov_b32 s{T1}, 0xf800000")
                a(f"v_cmp_gt_f32 vcc, s{T1}, v{x}")                 # tiny operand: scale by 2^32
                a(f"v_cndmask_b32 v{t1}, v{x}, v{t2}, vcc")
                a(f"v_sqrt_f32 v{t2}, v{t1}")
                a("s_nop 0")
                a(f"v_add_u32 v{t3}, -1, v{t2}")
                a(f"v_add_u32 v{t4}, 1, v{t2}")
                a(f"v_fma_f32 v{t5}, -v{t3}, v{t2}, v{t1}")
                a(f"v_fma_f32 v4, -v{t4}, v{t2}, v{t1}")
                a(f"v_cmp_ge_f32_e64 s[{T1}:{T2}], 0, v{t5}")
                a("s_nop 1")
                a(f"v_cndmask_b32_e64 v{t2}, v{t2}, v{t3}, s[{T1}:{T2}]")
                a(f"v_cmp_lt_f32_e64 s[{T1}:{T2}], 0, v4")
                a("s_nop 1")
                a(f"v_cndmask_b32_e64 v{t2}, v{t2}, v{t4}, s[{T1}:{T2}]")
                a(f"v_mul_f32 v{t3}, 0x37800000, v{t2}")
                a(f"v_cndmask_b32 v{t2}, v{t2}, v{t3}, vcc")
                a(f"s_movk_i32 s{T1}, 0x260")                       # +inf, +0, -0: the operand itself
                a(f"v_cmp_class_f32_e64 vcc, v{t1}, s{T1}")
                a("s_nop 1")
                a(f"v_cndmask_b32 v{res}, v{t2}, v{t1}, vcc")
            elif uop == "exp":
                a(f"v_mul_f32 v{t2}, 0x3fb8aa3b, v{x}")
                a(f"s_mov_b32 s{T1}, 0x3fb8aa3b")
                a(f"v_fma_f32 v{t3}, v{x}, s{T1}, -v{t2}")
                a(f"v_rndne_f32 v{t4}, v{t2}")
                a(f"v_fmamk_f32 v{t3}, v{x}, 0x32a5705f, v{t3}")
                a(f"v_sub_f32 v{t2}, v{t2}, v{t4}")
                a(f"v_add_f32 v{t2}, v{t2}, v{t3}")
                a(f"v_cvt_i32_f32 v{t4}, v{t4}")
                a(f"v_exp_f32 v{t2}, v{t2}")
                a(f"s_mov_b32 s{T1}, 0xc2ce8ed0")
                a(f"v_cmp_ngt_f32 vcc, s{T1}, v{x}")
                a(f"v_ldexp_f32 v{t2}, v{t2}, v{t4}")
                a(f"v_cndmask_b32 v{t2}, 0, v{t2}, vcc")
                a(f"s_mov_b32 s{T1}, 0x42b17218")
                a(f"v_cmp_nlt_f32 vcc, s{T1}, v{x}")
                a("s_nop 1")
                a(f"v_cndmask_b32 v{res}, v9, v{t2}, vcc")
            else:
                a(f"s_mov_b32 s{T1}, 0x800000")
                a(f"v_cmp_gt_f32 vcc, s{T1}, v{x}")                 # denormal operand: scale by 2^32
                a(f"v_cndmask_b32_e64 v{t2}, 0, 32, vcc")
                a(f"v_ldexp_f32 v{t1}, v{x}, v{t2}")
                a(f"v_log_f32 v{t1}, v{t1}")
                a(f"v_mov_b32 v{t2}, 0x41b17218")
                a(f"v_cndmask_b32 v{t2}, 0, v{t2}, vcc")
                a(f"v_mul_f32 v{t3}, 0x3f317217, v{t1}")
                a(f"s_mov_b32 s{T1}, 0x3f317217")
                a(f"v_fma_f32 v{t4}, v{t1}, s{T1}, -v{t3}")
                a(f"v_fmamk_f32 v{t4}, v{t1}, 0x3377d1cf, v{t4}")
                a(f"v_add_f32 v{t3}, v{t3}, v{t4}")
                a(f"s_mov_b32 s{T1}, 0x7f800000")
                a(f"v_cmp_lt_f32_e64 vcc, |v{t1}|, s{T1}")
                a("s_nop 1")
                a(f"v_cndmask_b32 v{t1}, v{t1}, v{t3}, vcc")
                if uop == "log":
                    a(f"v_sub_f32 v{res}, v{t1}, v{t2}")
                else:
                    a(f"v_sub_f32 v{t1}, v{t1}, v{t2}")
                    a(f"v_cmp_eq_f32 vcc, 0, v{x}")
                    a("s_nop 1")
                    a(f"v_cndmask_b32 v{res}, v{t1}, v9, vcc")      # LOOSE_LOG(0) = -MAX_VAL
        a(f"s_mov_b32 m0, s{sDST}")
        for k in range(K):
            a(f"v_mov_b32 v{S0 + k}, v{Q + k}")
        epilogue()
    # ---- sin / cos / tan of a block that holds an operand of 2^17 or more (or an infinity): the device library's WHOLE function --
    # both reductions under the exec masks the compiler gave them, gen/ocml_transcribe.py -- row by row like pow, in the registers above
    # the live operand stack.  (Round 3 handed such a tree to the register kernels at run time: 3 % of an evolved example/uci_sr.py
    # population after 30 generations, each costing as much as a thousand trees that stay.)  Only a stack that reaches into those
    # registers still bails out.
    for uop in ("sin", "cos", "tan"):
        if uop not in UNARY:
            continue
        a(f"{lab(f'triglib_{uop}')}:")
        a(f"s_cmp_gt_u32 s{sH}, {top - S0}")
        a(f"s_cbranch_scc1 {lab('bail_far')}")
        row_loop(f"triglib_{uop}", BODIES[uop], 1, Q)
        a(f"s_branch {lab(f'trigdone_{uop}')}")
    # The bodies above sit behind the 64-KiB-aligned handler table; the wave's outer loops sit in front of it, up to 64 KiB of
    # padding away.  A branch reaches 128 KiB: the pieces that jump BACK into the outer loops -- the run-time bail-out and
    # everything from the END handlers to the end of the kernel -- are therefore placed in front of the padding (reached from the
    # table's slots), and the one jump to them from a body behind the table goes through a computed address.
    a(f"{lab('bail_far')}:")
    a(f"s_getpc_b64 s[{T1}:{T2}]")
    a(f"{lab('bail_pc')}:")
    a(f"s_add_u32 s{T1}, s{T1}, {lab('bail')}-{lab('bail_pc')}")
    a(f"s_addc_u32 s{T2}, s{T2}, -1")             # (the distance is negative: sign extension of its low word)
    a(f"s_setpc_b64 s[{T1}:{T2}]")
    tail = []
    sect[0] = tail
    # runtime bail-out: this tree needs something the handlers do not carry.  Its fitness word gets the sentinel of the
    # register kernels, their pending flag is raised (flags bit 16: the call has a marks block, 128 bytes in front of the
    # first counter line), and the wave goes on with its next tree.
    a(f"{lab('bail')}:")
    a("s_set_gpr_idx_off")
    a(f"s_add_u32 s{T1}, s{sT0}, s{sB}")
    a(f"s_lshl_b32 s{T1}, s{T1}, 2")
    a(f"v_mov_b32 v4, s{T1}")
    a(f"v_mov_b32 v5, {hex(SENTINEL_HEAVY)}")
    a("s_mov_b64 exec, 1")
    a("global_store_dword v4, v5, s[10:11]")
    a("s_bitcmp1_b32 s17, 16")
    a(f"s_cbranch_scc0 {lab('bail_done')}")
    if fused:
        a("v_mov_b32 v9, 1")
        a("global_store_dword v[10:11], v9, off")         # (v[10:11]: word 0 of the scratch block itself)
    else:
        a(f"s_getreg_b32 s{T1}, hwreg(HW_REG_XCC_ID, 0, 4)")
        a(f"s_bfe_u32 s{T2}, s17, 0x4000c")
        a(f"s_and_b32 s{T1}, s{T1}, s{T2}")
        a(f"s_lshl_b32 s{T1}, s{T1}, 7")
        a(f"s_add_u32 s{T1}, s{T1}, 128")                 # this XCD's counter line -> word 0 of the scratch block
        a(f"v_sub_co_u32 v4, vcc, v10, s{T1}")
        a("v_subbrev_co_u32 v5, vcc, 0, v11, vcc")
        a("v_mov_b32 v9, 1")
        a("global_store_dword v[4:5], v9, off")
    a(f"{lab('bail_done')}:")
    a("s_mov_b64 exec, -1")
    a(f"s_branch {lab('next_tree')}")

    # end of a multi-output program: sum over the outputs of this tile's errors (forward.cu:383-390).  aux = K * out_len is
    # in T2; the labels of output o sit out_len "variables" behind X in LDS (o * variable stride behind those of output 0).
    a(f"{lab('endmo_body')}:")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_mul_i32 s{sA}, s15, {G * 1024}")           # LDS bytes between two variables / outputs
    a(f"s_mov_b32 s{sX}, 0")                          # K * output index
    a("v_mov_b32 v4, v3")                             # labels of output 0, this tile
    a(f"s_add_u32 s{T1}, s{sTILE}, 1")
    a(f"s_cmp_lt_u32 s{T1}, s15")
    a(f"s_cselect_b32 s{T1}, 0, s17")                 # flag bit 1 (ragged) counts only on the last tile ...
    a("s_bitcmp1_b32 s17, 4")
    a(f"s_cselect_b32 s{T1}, s17, s{T1}")             # ... or on every tile (bit 4: a piece of a dataset that is run in pieces)
    a(f"s_and_b32 s{T1}, s{T1}, 2")
    a(f"{lab('endmo_loop')}:")
    read_bank(T, 4)
    a(f"s_add_u32 m0, s{sX}, {hex(MODE['SRC0'] << 12)}")
    for k in range(K):
        a(f"v_mov_b32 v{Q + k}, v{S0 + k}")           # accumulator of this output
    a("s_mov_b32 m0, 0")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_cmp_eq_u32 s{T1}, 0")
    a(f"s_cbranch_scc1 {lab('endmo_full')}")
    a(f"s_mul_i32 s{T4}, s{sTILE}, {64 * K}")         # ragged tile: rows >= D contribute nothing
    a(f"v_mul_u32_u24 v5, {RPL}, v0")
    a(f"v_add_u32 v5, s{T4}, v5")
    for k in range(K):
        g, q = divmod(k, 4)
        a(f"v_add_u32 v9, {g * 256 + q}, v5")
        a(f"v_sub_f32 v{T + k}, v{T + k}, v{Q + k}")
        a("v_cmp_gt_u32 vcc, s13, v9")
        a(f"v_cndmask_b32 v{T + k}, 0, v{T + k}, vcc")  # a zero difference adds nothing, squared or not
    a(f"s_branch {lab('endmo_sum')}")
    a(f"{lab('endmo_full')}:")
    rows_op("sub", T, T, Q)       # (M0 is 0 here, but VGPR indexing is on: packed instructions, see rows_op)
    a(f"{lab('endmo_sum')}:")
    a("s_bitcmp0_b32 s17, 0")
    a(f"s_cbranch_scc1 {lab('endmo_abs')}")
    rows_op("mul", T, T, T)
    for k in range(K):
        a(f"v_add_f32 v6, v6, v{T + k}")
    a(f"s_branch {lab('endmo_next')}")
    a(f"{lab('endmo_abs')}:")
    for k in range(K):
        a(f"v_add_f32_e64 v6, v6, |v{T + k}|")
    a(f"{lab('endmo_next')}:")
    a(f"s_add_u32 s{sX}, s{sX}, {K}")
    a(f"v_add_u32 v4, s{sA}, v4")
    a(f"s_cmp_lt_u32 s{sX}, s{T2}")
    a(f"s_cbranch_scc1 {lab('endmo_loop')}")
    a("s_set_gpr_idx_off")
    a(f"s_add_u32 s{T1}, s{sTILE}, 1")
    a(f"s_branch {lab('end_acc')}")

    # end of a classifier program: per row, is the arg-max over the out_len accumulators -- as torch.argmax(clip(softmax(x))) sees it:
    # the FIRST maximum; index 0 when any output is NaN or the maximum is infinite (the soft-max row is then NaN) -- the row's label?
    # v6 counts the hits (as a float: exact below 2^24 rows).  aux = K * out_len is in T2, the labels (int32 bits) in the T bank.
    # The launch staged the rows IN THE ORDER OF THEIR LABELS (sr_tc.hip, tc_label_groups_kernel): the 64 lanes of a row register hold
    # one label, or a few in consecutive lanes (-1: no row), so the label L of a SEGMENT of lanes is a scalar and the question needs no
    # index search.  With P_j = max(out_0 .. out_j), the running maxima written over the accumulators, and M = P_(n-1):
    #     hit        <=>  P_L == M  and  P_(L-1) < M - d       (output L is the maximum and nothing in front of it comes close)
    #     ambiguous  <=>  P_L >= M - d  and  P_(L-1) < M  and not hit
    # d = 1.25 * 2^-23 (interp.hpp kSoftmaxTieMargin): an output within d of the maximum may round to the same soft-max probability, and
    # torch then returns whichever comes first.  A tree with an ambiguous row leaves through the run-time bail-out: sr_wide.hip's recount
    # kernel evaluates it with torch's own arithmetic.  (Ambiguity only matters where the label is one of the close outputs; a near-tie
    # between two other outputs in front of the label also trips the test: rare, and the recount is exact.)  Rows whose maximum is
    # infinite or that hold a NaN -- a NaN sum <=> a NaN output, or both infinities, which the infinite maximum covers -- get M = NaN:
    # every compare fails, and they are hits where the label is 0.
    # Per tile and 10 outputs: ~270 vector instructions instead of the ~620 of the index search over all outputs (round 5).
    Mx, Sm = P[0], P[1]
    LOGK = K.bit_length() - 1
    a(f"{lab('endcls_body')}:")
    a("s_mov_b32 m0, 0")
    rows_mov(Sm, S0)
    a(f"s_mov_b32 s{sX}, {K}")                         # K * output index
    a(f"s_cmp_lt_u32 s{sX}, s{T2}")
    a(f"s_cbranch_scc0 {lab('endcls_first')}")
    a(f"{lab('endcls_max')}:")
    a(f"s_add_u32 m0, s{sX}, {hex(MODE['SRC0'] << 12)}")
    if K >= 2:
        for k in range(0, K, 2):
            a(f"v_pk_add_f32 v[{Sm + k}:{Sm + k + 1}], v[{S0 + k}:{S0 + k + 1}], v[{Sm + k}:{Sm + k + 1}]")
    else:
        a(f"v_add_f32 v{Sm}, v{S0}, v{Sm}")
    a(f"s_add_u32 m0, s{sX}, {hex((MODE['SRC0'] | MODE['SRC1'] | MODE['DST']) << 12)}")
    for k in range(K):
        a(f"v_max_f32 v{S0 + k}, v{S0 + k}, v{S0 - K + k}")      # P_j = max(out_j, P_(j-1)), in place
    a(f"s_add_u32 s{sX}, s{sX}, {K}")
    a(f"s_cmp_lt_u32 s{sX}, s{T2}")
    a(f"s_cbranch_scc1 {lab('endcls_max')}")
    a(f"{lab('endcls_first')}:")
    a(f"s_sub_u32 s{sX}, s{T2}, {K}")
    a(f"s_add_u32 m0, s{sX}, {hex(MODE['SRC0'] << 12)}")
    for k in range(K):
        a(f"v_mov_b32 v{Mx + k}, v{S0 + k}")
    a("s_mov_b32 m0, 0")
    # (control registers that are dead until the next pass of the program re-initialises them -- the handler address's low word, J, H --
    # serve as scratch: three mask registers besides VCC)
    sL, sM2 = sPC, sJ
    assert sJ % 2 == 0 and sH == sJ + 1
    masks = ["vcc", f"s[{sM2}:{sM2 + 1}]", f"s[{T1}:{T2}]"]
    for k in range(K):
        a(f"v_fma_f32 v{Q + k}, v{Mx + k}, 0, v{Sm + k}")          # NaN <=> a NaN sum or an infinite (NaN) maximum
    for k in range(K + 2):                                        # (the mask of row k is read two instructions after its compare)
        if k < K:
            a(f"v_cmp_u_f32 {masks[k % 3]}, v{Q + k}, v{Q + k}")
        if k >= 2:
            j = k - 2
            if K < 3:
                a("s_nop 1")
            a(f"v_cndmask_b32_e64 v{Mx + j}, v{Mx + j}, v8, {masks[j % 3]}")   # v8 = NaN
    for k in range(K):
        a(f"v_add_f32 v{Sm + k}, 0xb4200000, v{Mx + k}")           # M - d
        a(f"v_mov_b32 v{Q + k}, 0xff800000")                       # "the maximum in front of output 0" (the registers below the accumulators)
    for k in range(K):
        seg, row, nsp = lab(f"endcls_seg{k}"), lab(f"endcls_row{k}"), lab(f"endcls_nsp{k}")
        a(f"v_cmp_le_i32 vcc, 0, v{T + k}")
        a(f"s_mov_b64 s[{sA}:{sA + 1}], vcc")                      # lanes of this register still to be judged
        a(f"{seg}:")
        a(f"s_cmp_eq_u64 s[{sA}:{sA + 1}], 0")
        a(f"s_cbranch_scc1 {row}")
        a(f"s_ff1_i32_b64 s{T4}, s[{sA}:{sA + 1}]")
        a(f"v_readlane_b32 s{T4}, v{T + k}, s{T4}")                # the segment's label
        a("s_nop 1")
        a(f"v_cmp_eq_u32 vcc, s{T4}, v{T + k}")
        a(f"s_lshl_b32 s{sL}, s{T4}, {LOGK}")
        a(f"s_andn2_b64 s[{sA}:{sA + 1}], s[{sA}:{sA + 1}], vcc")
        a("s_mov_b64 exec, vcc")
        a(f"s_add_u32 m0, s{sL}, {hex(MODE['SRC0'] << 12)}")       # source 0: P_L / P_(L-1)
        a(f"v_cmp_eq_f32 vcc, v{S0 + k}, v{Mx + k}")
        a(f"v_cmp_lt_f32 s[{T1}:{T2}], v{S0 - K + k}, v{Sm + k}")
        a(f"v_cmp_ge_f32 s[{sM2}:{sM2 + 1}], v{S0 + k}, v{Sm + k}")
        a(f"s_and_b64 s[{T1}:{T2}], vcc, s[{T1}:{T2}]")            # hit
        a(f"v_cmp_lt_f32 vcc, v{S0 - K + k}, v{Mx + k}")
        a("s_mov_b32 m0, 0")
        a(f"s_and_b64 s[{sM2}:{sM2 + 1}], s[{sM2}:{sM2 + 1}], vcc")
        a(f"s_andn2_b64 s[{sM2}:{sM2 + 1}], s[{sM2}:{sM2 + 1}], s[{T1}:{T2}]")   # ambiguous
        a(f"s_cmp_lg_u32 s{T4}, 0")
        a(f"s_cbranch_scc1 {nsp}")
        a(f"v_cmp_u_f32 vcc, v{Mx + k}, v{Mx + k}")                # label 0: the rows without a soft-max
        a(f"s_or_b64 s[{T1}:{T2}], s[{T1}:{T2}], vcc")
        a(f"{nsp}:")
        a(f"v_cndmask_b32_e64 v9, 0, 1.0, s[{T1}:{T2}]")
        a("v_add_f32 v6, v6, v9")
        a("s_mov_b64 exec, -1")
        a(f"s_cmp_lg_u64 s[{sM2}:{sM2 + 1}], 0")
        a(f"s_cbranch_scc1 {lab('bail')}")
        a(f"s_branch {seg}")
        a(f"{row}:")
    a("s_set_gpr_idx_off")
    a(f"s_add_u32 s{T1}, s{sTILE}, 1")
    a(f"s_branch {lab('end_acc')}")

    # end of the program: fold this tile's errors into the accumulator.  The labels of the tile were prefetched by the
    # last instruction of the program (the compiler gives END the LDS offset of y as its "variable"), so they sit in
    # the current operand bank of END's flavour.
    for fl in (0, 1):
        Y = P[fl]
        a(f"{lab(f'endbody{fl}')}:")
        a("s_set_gpr_idx_off")
        if stats:
            pass
        a(f"s_add_u32 s{T1}, s{sTILE}, 1")
        a("s_waitcnt lgkmcnt(0)")
        if EARLYREC and not stats:
            # the last tile of a tree that has a successor in its batch: the program window is dead from here on, so the next
            # tree's record starts its way now -- under the rows of END and the reduction -- and not after them (flags bit 10,
            # set by the host for single-output launches, tells tree_done that it did)
            a(f"s_cmp_lt_u32 s{T1}, s15")
            a(f"s_cbranch_scc1 {lab(f'no_early{fl}')}")
            a(f"s_add_u32 s{T2}, s{sB}, 1")
            a(f"s_cmp_lt_u32 s{T2}, s{sNB}")
            a(f"s_cbranch_scc0 {lab(f'no_early{fl}')}")
            if not fused:
                a(f"s_add_u32 s{T2}, s{T2}, s{sT0}")
            rec_addr(sREC, sREC + 1, f"s{T2}", T2)
            load_window()
            a(f"{lab(f'no_early{fl}')}:")
        a(f"s_and_b32 s{T2}, s17, 0x12")   # a launch without a ragged tile (neither bit 1 nor bit 4) needs none of the following
        a(f"s_cbranch_scc0 {lab(f'end_full{fl}')}")
        a(f"s_cmp_lt_u32 s{T1}, s15")
        a(f"s_cselect_b32 s{T2}, 0, s17")  # flag bit 1 (ragged) survives only on the last tile ...
        a("s_bitcmp1_b32 s17, 4")
        a(f"s_cselect_b32 s{T2}, s17, s{T2}")  # ... unless bit 4 says that any tile of this launch can reach past its rows
        a(f"s_and_b32 s{T2}, s{T2}, 2")
        a("s_waitcnt lgkmcnt(0)")
        a(f"s_cmp_eq_u32 s{T2}, 0")
        a(f"s_cbranch_scc1 {lab(f'end_full{fl}')}")
        # ragged tile: rows >= D contribute nothing
        a(f"s_mul_i32 s{T2}, s{sTILE}, {64 * K}")
        a(f"v_mul_u32_u24 v5, {RPL}, v0")
        for k in range(K):
            g, q = divmod(k, 4)
            a(f"s_add_u32 s{T4}, s{T2}, {g * 256 + q}")
            a(f"v_add_u32 v4, s{T4}, v5")
            a(f"v_sub_f32 v9, v{Y + k}, v{S0 + k}")
            a("v_cmp_gt_u32 vcc, s13, v4")
            a("s_bitcmp0_b32 s17, 0")
            a(f"s_cbranch_scc1 {lab(f'rag_abs{fl}_{k}')}")
            a("v_mul_f32 v9, v9, v9")
            a(f"{lab(f'rag_abs{fl}_{k}')}:")
            a("v_and_b32 v9, 0x7fffffff, v9")
            a("v_cndmask_b32 v9, 0, v9, vcc")
            a("v_add_f32 v6, v6, v9")
        a(f"s_branch {lab('end_acc')}")
        a(f"{lab(f'end_full{fl}')}:")
        a("s_bitcmp0_b32 s17, 0")
        a(f"s_cbranch_scc1 {lab(f'end_abs{fl}')}")
        # (differences and squares in the result's own registers, row after row independent: only the K additions into the
        # accumulator form a chain -- with one temporary the 3 K instructions were one)
        if FMA_LOSS:
            for k in range(K):
                a(f"v_sub_f32 v9, v{Y + k}, v{S0 + k}")
                a("v_fmac_f32 v6, v9, v9")
        else:
            for k in range(K):
                a(f"v_sub_f32 v{S0 + k}, v{Y + k}, v{S0 + k}")
            for k in range(K):
                a(f"v_mul_f32 v{S0 + k}, v{S0 + k}, v{S0 + k}")
            for k in range(K):
                a(f"v_add_f32 v6, v6, v{S0 + k}")
        a(f"s_branch {lab('end_acc')}")
        a(f"{lab(f'end_abs{fl}')}:")
        for k in range(K):
            a(f"v_sub_f32 v{S0 + k}, v{Y + k}, v{S0 + k}")
        for k in range(K):
            a(f"v_add_f32_e64 v6, v6, |v{S0 + k}|")
        if fl == 0:
            a(f"s_branch {lab('end_acc')}")
    a(f"{lab('end_acc')}:")
    a(f"s_mov_b32 s{sTILE}, s{T1}")
    a(f"s_cmp_lt_u32 s{sTILE}, s15")
    a(f"s_cbranch_scc0 {lab('tree_done')}")
    # another pass over the next tile: a two-block program left its overflow block in the window (NEXT moved sREC)
    if fused:
        rec_addr(T1, T2, f"s{sB}", T1)
    else:
        a(f"s_add_u32 s{T1}, s{sT0}, s{sB}")
        rec_addr(T1, T2, f"s{T1}", T1)
    a(f"s_cmp_eq_u32 s{T1}, s{sREC}")
    a(f"s_cbranch_scc1 {lab('tile')}")
    a(f"s_mov_b32 s{sREC}, s{T1}")
    a(f"s_mov_b32 s{sREC + 1}, s{T2}")
    load_window()
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_branch {lab('tile')}")
    a(f"{lab('tree_done')}:")
    # the tree is finished: fixed-order sum of the 64 lanes, lane b of v7 receives it
    for ctl in ("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0",
                "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0",
                "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"):
        a("s_nop 1")
        a(f"v_add_f32_dpp v6, v6, v6 {ctl}")
    a("s_nop 1")
    a(f"v_readlane_b32 s{T1}, v6, 63")
    # a NaN sum leaves as THE quiet NaN: which operand's payload an instruction hands on is its own business (the division rows
    # above differ in it), and a payload must never look like one of the register kernels' sentinels
    a(f"s_and_b32 s{T2}, s{T1}, 0x7fffffff")
    a(f"s_cmp_gt_u32 s{T2}, 0x7f800000")
    a(f"s_cselect_b32 s{T1}, 0x7fc00000, s{T1}")
    a(f"s_mov_b32 m0, s{sB}")
    a(f"s_bitset1_b64 s[{sOK}:{sOK + 1}], s{sB}")
    a(f"v_writelane_b32 v7, s{T1}, m0")
    if EARLYREC and not stats:
        a(f"s_add_u32 s{sB}, s{sB}, 1")
        a(f"s_cmp_lt_u32 s{sB}, s{sNB}")
        a(f"s_cbranch_scc0 {lab('batch_end')}")
        a("s_bitcmp1_b32 s17, 10")                 # single-output launch: END has sent for this tree's record
        a(f"s_cbranch_scc1 {lab('tree_loaded')}")
        a(f"s_branch {lab('tree')}")
    a(f"{lab('next_tree')}:")
    a(f"s_add_u32 s{sB}, s{sB}, 1")
    a(f"s_cmp_lt_u32 s{sB}, s{sNB}")
    a(f"s_cbranch_scc1 {lab('tree')}")
    a(f"{lab('batch_end')}:")
    # batch finished: everything outstanding has long landed (warm-up load, the next dynamic grab); take the grab
    # first, then mean = sum / D and one coalesced store for the evaluated trees
    tick_begin()
    if fused:
        a("s_waitcnt vmcnt(0)")
    else:
        # the grab of the NEXT batch.  From an XCD's region it is a global atomic, the oldest of the vector-memory operations in flight (the
        # warm-up loads behind it may stay out: nobody reads their registers); from the workgroup's pool it is an LDS operation, and the
        # warm-up loads -- issued one tree ago -- are not waited for at all (round 6: they were, with the pool's first build: 20-30 %
        # of a wave's clocks at 100 k trees)
        a("s_cmp_eq_u32 s18, 1")
        a(f"s_cbranch_scc1 {lab('pool_batch_end')}")
        a("s_cmp_eq_u32 s18, 0")
        a(f"s_cbranch_scc1 {lab('dyn_batch_end')}")
        a("s_waitcnt vmcnt(0)")                       # (s18 == 2: the pool ran dry during this batch; the region's grab is the youngest operation in flight)
        a("s_mov_b32 s18, 0")
        a(f"s_branch {lab('pool_batch_end')}")
        a(f"{lab('dyn_batch_end')}:")
        a(f"s_waitcnt vmcnt({2 if L2WARM else 0})")
        a(f"{lab('pool_batch_end')}:")
        a("s_waitcnt lgkmcnt(0)")
    tick_end(A_WORK)
    NEXT_BATCH = lab('exit') if fused else lab('batch')
    if not fused:
        a(f"v_readfirstlane_b32 s{sT0N}, v12")
        a("s_cmp_eq_u32 s18, 0")
        a(f"s_cselect_b32 s{T1}, {DYN}, {CUR}")
        a(f"s_add_u32 s{sT0N}, s{sT0N}, s{T1}")
    a(f"s_cmp_eq_u64 s[{sOK}:{sOK + 1}], 0")
    a(f"s_cbranch_scc1 {NEXT_BATCH}")
    a(f"s_mov_b64 exec, s[{sOK}:{sOK + 1}]")
    # A dataset larger than LDS is run in pieces, one launch per piece: flags bit 2 = add this piece's sums to what the fitness
    # words hold (a word that holds the register kernels' sentinel -- a run-time bail-out in an earlier piece -- keeps it),
    # bit 3 = store the sum itself (the mean is taken behind the last piece, by tc_scale_kernel).
    a(f"v_add_u32 v4, s{sT0}, v0")
    a("v_lshlrev_b32 v4, 2, v4")
    a("s_bitcmp0_b32 s17, 2")
    a(f"s_cbranch_scc1 {lab('fin_first')}")
    a("global_load_dword v5, v4, s[10:11]")
    a(f"s_mov_b32 s{T1}, {hex(SENTINEL_HEAVY)}")
    a("s_waitcnt vmcnt(0)")
    a("v_add_f32 v7, v5, v7")
    a(f"v_cmp_eq_u32 vcc, s{T1}, v5")
    a("s_nop 1")
    a("v_cndmask_b32 v7, v7, v5, vcc")
    a(f"{lab('fin_first')}:")
    a("s_bitcmp0_b32 s17, 3")
    a(f"s_cbranch_scc1 {lab('fin_mean')}")
    a("global_store_dword v4, v7, s[10:11]")
    a("s_mov_b64 exec, -1")
    a(f"s_branch {NEXT_BATCH}")
    a(f"{lab('fin_mean')}:")
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
    a(f"global_store_dword v4, v{d3}, s[10:11]")
    a("s_mov_b64 exec, -1")
    a(f"s_branch {NEXT_BATCH}")

    if fused:
        a(f"{lab('exit')}:")
        a(f"s_getpc_b64 s[{T1}:{T2}]")
        a(f"{lab('exit_pc')}:")
        a(f"s_add_u32 s{T1}, s{T1}, {lab('exit_here')}-{lab('exit_pc')}")
        a(f"s_addc_u32 s{T2}, s{T2}, 0")
        a(f"s_setpc_b64 s[{T1}:{T2}]")
    else:
        a(f"{lab('exit')}:")
    if stats:  # {record wait, work wait, trees, 4 * dispatches, wave ticks, waves} += this wave's counters
        a(f"s_memtime s[{T1}:{T2}]")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_sub_u32 v{A_START}, s{T1}, v{A_START}")
        a(f"s_load_dwordx2 s[{T1}:{T2}], %[karg], 0x58")
        a("s_waitcnt lgkmcnt(0)")
        a(f"v_mov_b32 v10, s{T1}")
        a(f"v_mov_b32 v11, s{T2}")
        a("v_mov_b32 v13, 0")
        a("s_mov_b64 exec, 1")
        for i, src in enumerate((A_REC, A_WORK, A_TREES, A_DISP, A_START, None)):
            if src is None:
                a("v_mov_b32 v12, 1")
            else:
                a(f"v_mov_b32 v12, v{src}")
            a(f"global_atomic_add_x2 v[10:11], v[12:13], off offset:{8 * i}")
        a("s_waitcnt vmcnt(0)")
    if not fused:
        a("s_endpgm")
    sect[0] = L
    a(f"{lab('code_end')}:")
    if CNDE64:
        import re
        L[:] = [re.sub(r"^v_cndmask_b32 (.*), vcc$", r"v_cndmask_b32_e64 \1, vcc", x) for x in L]
        tail[:] = [re.sub(r"^v_cndmask_b32 (.*), vcc$", r"v_cndmask_b32_e64 \1, vcc", x) for x in tail]
    at = L.index(".p2align 16")
    L[at:at] = tail
    if fused:
        # the statement's end -- behind the handler table and its bodies, up to 64 KiB + the bodies away from the batch epilogue,
        # which therefore comes here through a computed address (a branch reaches 128 KiB, but only just)
        a(f"{lab('exit_here')}:")

    if info is not None:
        info["K"], info["depth"], info["nhandlers"], info["slot"] = K, DEPTH, NHF, SLOT
        info["handlers"] = {n: dict(id=i, **count_path(L, lab("h0_" + n), {lab("end_full0"), lab("tile")}, idx_on=not n.startswith("end")))
                            for n, i in hid.items()}
    body = "\n".join(f'    "{line}\\n\\t"' for line in L)
    clob = ['"memory"', '"vcc"', '"scc"'] + [f'"s{i}"' for i in range(8, 102)] + [f'"v{i}"' for i in range(0, NV)]
    clob_txt = ", ".join(clob)
    name = f"K{K}" + ("W" if wide else "") + ("S" if stats else "") + ("", "F", "H")[fast] + ("U" if fused else "")
    out = f"// GENERATED by gen/gen_tc_asm.py (K = {K} rows per lane, {DEPTH}-entry operand stack, VGPRs v0..v{NV - 1}) — do not edit.\n"
    out += f"#define EVOGP_TC_{name}_DEPTH {DEPTH}\n#define EVOGP_TC_{name}_VGPRS {NV}\n"
    if K == 8 and not stats and not fast and not fused and not wide:
        out += f"#define EVOGP_TC_SLOT {SLOT}\n#define EVOGP_TC_NHANDLERS {NHF}\n#define EVOGP_TC_UNARY_MASK {(1 << len(UNARY)) - 1}\n#define EVOGP_TC_HEAVY_REGS {HEAVY_REGS}\n#define EVOGP_TC_DIVIP_REGS {DIVIP_REGS}\n"
        for n, i in sorted(hid.items(), key=lambda kv: kv[1]):
            out += f"#define EVOGP_TC_H_{n.upper()} {i}\n"
    if fused:
        # v12-v17 are not touched (the compiler keeps its own state across the statement there)
        clob = ['"memory"', '"vcc"', '"scc"', '"m0"'] + [f'"s{i}"' for i in range(8, 102)] + [f'"v{i}"' for i in range(0, NV) if not 13 <= i <= 17]
        clob_txt = ", ".join(clob)
        out += f"#define EVOGP_TC_FUSED_TOUCH {1 if TOUCH else 0}\n" if name == "K8U" else ""
        out += f"#define EVOGP_TC_FUSED_SIZE_OFF {FUSED_SIZE_OFF}\n#define EVOGP_TC_FUSED_LEN_OFF {FUSED_LEN_OFF}\n" if name == "K8U" else ""
        out += f"#define EVOGP_TC_ASM_{name}(karg_, t0_, nb_, roff_, ldsx_, trust_, taddr_, t0n_, lensn_) \\\n  asm volatile( \\\n"
        out += "\n".join(line + " \\" for line in body.split("\n"))
        out += f'''
    : [lensn] "=&v"(lensn_) \\
    : [karg] "s"(karg_), [t0] "s"(t0_), [nb] "s"(nb_), [roff] "s"(roff_), [ldsx] "s"(ldsx_), [trust] "s"(trust_), [taddr] "v"(taddr_), [t0n] "s"(t0n_) \\
    : {clob_txt})
'''
        return out
    out += f"#define EVOGP_TC_ASM_{name}(karg_, wgid_, ldsx_, wave_, dyn_, pf_, base_) \\\n  asm volatile( \\\n"
    out += "\n".join(line + " \\" for line in body.split("\n"))
    out += f'''
    : [cur] "+s"(wgid_), [lim] "+s"(ldsx_), [stride] "+s"(wave_), [dyn] "+s"(dyn_), [pf] "+s"(pf_), [base] "+s"(base_) \\
    : [karg] "s"(karg_) \\
    : {clob_txt})
'''
    return out


if __name__ == "__main__":
    import os
    KWARM = os.environ.get("EVOGP_TC_GEN_KWARM", "0") == "1"
    CNDE64 = os.environ.get("EVOGP_TC_GEN_CNDE64", "0") == "1"
    CODEWARM = os.environ.get("EVOGP_TC_GEN_CODEWARM", "0") == "1"
    L2WARM = os.environ.get("EVOGP_TC_GEN_L2WARM", "1") != "0"
    EARLYREC = os.environ.get("EVOGP_TC_GEN_EARLYREC", "1") != "0"
    KWARM_LINES = int(os.environ.get("EVOGP_TC_GEN_KWARM_LINES", "4"))
    NOPF = os.environ.get("EVOGP_TC_GEN_NOPF", "0") == "1"
    FMA_LOSS = os.environ.get("EVOGP_TC_GEN_FMA_LOSS", "0") == "1"
    SPLAT = os.environ.get("EVOGP_TC_GEN_SPLAT", "0") == "1"
    DIVRANGE = os.environ.get("EVOGP_TC_GEN_DIVRANGE", "1") != "0"
    DIVFIX = os.environ.get("EVOGP_TC_GEN_DIVFIX", "0") == "1"
    TRUST = os.environ.get("EVOGP_TC_GEN_TRUST", "1") != "0"
    PKCONST = os.environ.get("EVOGP_TC_GEN_PKCONST", "1") != "0"
    DIVABREAST = os.environ.get("EVOGP_TC_GEN_DIVABREAST", "1") != "0"
    PKARITH = os.environ.get("EVOGP_TC_GEN_PKARITH", "1") != "0"
    RECGLC = os.environ.get("EVOGP_TC_GEN_RECGLC", "0") == "1"
    TOUCH = os.environ.get("EVOGP_TC_GEN_TOUCH", "1") != "0"
    TRIGPK = os.environ.get("EVOGP_TC_GEN_TRIGPK", "1") != "0"
    LIBPK = os.environ.get("EVOGP_TC_GEN_LIBPK", "1") != "0"
    outdir = sys.argv[1] if len(sys.argv) > 1 else "."
    import json
    table = {}
    # (K = 16 -- tiles of 1024 rows, 248 VGPRs, two waves per SIMD, half the dispatches and scalar instructions per row --
    # generates and runs, fitness words identical, but is 7 % SLOWER at 1 M trees, with the division's row pairs abreast or not:
    # profiles/r03M_div_range_ab.log; docs/DESIGN_history_r01_r03.md section 3.1d.  Not built.)
    for K, depth in ((8, 9), (4, 15), (1, 44)):
        with open(f"{outdir}/tc_interp_k{K}.inc", "w") as f:
            for fast, tag in ((0, "ieee"), (1, "fast"), (2, "short")):  # division: IEEE / no range scaling / one correction
                info = {}
                f.write(gen(K, depth, fast=fast, info=info))
                table[f"K{K}_{tag}"] = info
            if K == 8:
                for fast in (0, 1, 2):   # the fused build (one batch per entry, sr_fused_kernel)
                    f.write(gen(K, depth, fast=fast, fused=True))
                f.write(gen(K, depth, stats=True))  # cycle-accounting build (its top stack slot holds the counters)
                # the WIDE-stack build: 13 entries (160 VGPRs, three waves per SIMD, workgroups of 12 waves) for forests of 7-10 outputs,
                # whose accumulators do not fit the nine entries above and which otherwise run 4 rows per lane (BASELINE configs[3]:
                # 10 outputs, 1797 rows -- twice the dispatches for the same rows)
                for fast in (0, 1, 2):
                    f.write(gen(K, 13, fast=fast, wide=True))
        print("wrote", f"{outdir}/tc_interp_k{K}.inc")
    # per-handler instruction counts of the generated interpreters, read by bench.py (a build artefact like the .inc files)
    os.makedirs(f"{outdir}/../lib", exist_ok=True)
    with open(f"{outdir}/../lib/tc_handlers.json", "w") as f:
        json.dump(table, f, indent=1)
