"""CPU: host-side logic of the reference API surface (descriptor tensors, selection / crossover /
mutation index arithmetic, Forest container protocol, pipeline loop) on CPU tensors with the
test-only oracle-backed ops."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cpu_ops  # noqa: E402

cpu_ops.register()
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming  # noqa: E402
from evogp_amd.pipeline import StandardPipeline  # noqa: E402
from evogp_amd.problem import SymbolicRegression  # noqa: E402
from evogp_amd.tree import MAX_STACK, Forest, GenerateDescriptor, NType, Tree, randint, set_default_device  # noqa: E402

from evogp_amd.tree import utils as _tree_utils  # noqa: E402


@pytest.fixture(autouse=True)
def _cpu_default_device():
    """These tests run the host logic on CPU tensors; restore the device afterwards so that GPU tests
    collected in the same session are unaffected."""
    saved = _tree_utils._DEVICE
    set_default_device("cpu")
    yield
    _tree_utils._DEVICE = saved


def desc(**kw):
    base = dict(max_tree_len=64, input_len=3, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    base.update(kw)
    return GenerateDescriptor(**base)


def test_descriptor_tensors():
    d = desc()
    assert d.depth2leaf_probs.tolist() == pytest.approx([0.2] * 4 + [1.0] * 6)
    assert d.roulette_funcs.tolist() == [0.0, 0.25, 0.5, 0.75] + [1.0] * 25
    assert d.roulette_bfuncs.tolist() == d.roulette_funcs.tolist() and d.roulette_ufuncs.sum() == 0
    assert d.const_samples.tolist() == [-1.0, 0.0, 1.0]
    d2 = d.update(max_layer_cnt=3, layer_leaf_prob=0.5)
    assert d2.depth2leaf_probs.tolist() == pytest.approx([0.5] * 2 + [1.0] * 8) and d2.max_tree_len == 64
    w = desc(using_funcs={"+": 3.0, "sin": 1.0})
    assert w.roulette_funcs[1].item() == pytest.approx(0.75) and w.roulette_funcs[28].item() == pytest.approx(1.0)
    r = desc(const_samples=None, const_range=(2.0, 3.0), sample_cnt=10)
    assert r.const_samples.shape == (10,) and (r.const_samples >= 2).all() and (r.const_samples <= 3).all()


def test_descriptor_validation():
    # (AssertionError is the reference's error type for all of these: descriptor.py:8-32, 64-75)
    with pytest.raises(AssertionError, match="31 nodes"):
        desc(max_tree_len=16, max_layer_cnt=5)            # full binary tree of 5 layers
    with pytest.raises(AssertionError, match="121 nodes"):
        desc(using_funcs=["if", "+"], max_tree_len=64, max_layer_cnt=5)  # ternary
    with pytest.raises(AssertionError, match="operand-stack bound"):
        desc(max_tree_len=MAX_STACK + 1)
    with pytest.raises(AssertionError, match="frobnicate"):
        desc(using_funcs=["+", "frobnicate"])
    with pytest.raises(AssertionError):
        desc(const_prob=1.5)


def test_forest_container_protocol():
    f = Forest.random_generate(40, desc(), keys=torch.tensor([3, 4]))
    assert len(f) == 40 and f.max_tree_len == 64 and isinstance(f[0], Tree)
    assert len(f[5:15]) == 10 and len(f[torch.tensor([1, 3, 5])]) == 3 and len(f[np.array([0, 1])]) == 2
    g = f + f[:3] + f[0]
    assert len(g) == 44
    f2 = Forest.zero_generate(40, 64, 3, 1)
    f2[10:20] = f[0:10]
    assert torch.equal(f2.batch_node_type[10:20], f.batch_node_type[:10])
    f2[0] = f[39]
    assert torch.equal(f2.batch_subtree_size[0], f.batch_subtree_size[39])
    assert sum(1 for _ in f) == 40
    h = pickle.loads(pickle.dumps(f))
    assert torch.equal(h.batch_node_value.view(torch.int32), f.batch_node_value.view(torch.int32)) and h.input_len == 3
    with pytest.raises(Exception):
        f["x"]
    assert Forest.zero_generate(2, 8, 1, 1).batch_node_type[:, 0].tolist() == [NType.CONST] * 2
    assert (randint((1000,), 3, 9) >= 3).all() and (randint((1000,), 3, 9) < 9).all()


def test_default_selection_counts_and_order():
    f = Forest.zero_generate(10, 8, 1, 1)
    fit = torch.tensor([0.1, 0.9, 0.5, 0.9, -1.0, 0.3, 0.2, 0.8, 0.0, 0.4])
    elite, surv = DefaultSelection(survival_rate=0.3, elite_cnt=2)(f, fit)
    assert elite.dtype == torch.int32 and elite.tolist() == [1, 3] and surv.tolist() == [1, 3, 7]  # stable on the tie
    elite, surv = DefaultSelection(survival_rate=0.5, elite_rate=0.1)(f, fit)
    assert elite.tolist() == [1] and len(surv) == 5
    assert DefaultSelection(0.3)(f, fit)[0].numel() == 0
    with pytest.raises(AssertionError):
        DefaultSelection(0.3, elite_cnt=1, elite_rate=0.1)


def test_crossover_mutation_keep_population_valid_and_sized():
    from oracle.pyoracle import Oracle

    o = Oracle("port")
    torch.manual_seed(1)
    d = desc()
    f = Forest.random_generate(300, d, keys=torch.tensor([8, 8]))
    fit = torch.randn(300)
    algo = GeneticProgramming(f, DefaultCrossover(), DefaultMutation(0.3, d.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.02),
                              enable_pareto_front=True)
    elites = f[torch.sort(fit, descending=True, stable=True).indices[:6]]
    nxt = algo.step(fit)
    assert nxt.pop_size == 300
    assert torch.equal(nxt.batch_node_type[:6], elites.batch_node_type)  # elites first, unchanged
    t, s = nxt.batch_node_type.numpy(), nxt.batch_subtree_size.numpy()
    assert all(o.validate_tree(t[i], s[i]) == 0 for i in range(300))
    assert (s[:, 0] <= 64).all() and np.isfinite(algo.pareto_front.fitness.numpy()).any()
    assert DefaultMutation(0.0, d)(nxt) is nxt  # nothing selected: the forest comes back untouched


def test_pipeline_xor_improves_on_cpu_ops(capsys):
    torch.manual_seed(0)
    X = torch.tensor([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=torch.float32)
    y = (X.sum(1) % 2)[:, None]
    prob = SymbolicRegression(datapoints=X, labels=y)
    assert prob.problem_dim == 3 and prob.solution_dim == 1
    d = desc()
    algo = GeneticProgramming(Forest.random_generate(400, d), DefaultCrossover(), DefaultMutation(0.2, d.update(max_layer_cnt=3)),
                              DefaultSelection(0.3, elite_rate=0.01))
    pipe = StandardPipeline(algo, prob, generation_limit=8, is_show_details=True)
    best = pipe.run()
    assert "gen    7" in capsys.readouterr().out
    assert isinstance(best, Tree) and float(pipe.best_fitness) > -0.5
    assert best.forward(X).shape == (8, 1) and pipe.fitness.shape == (400,)
    early = StandardPipeline(algo, prob, fitness_target=-10.0, is_show_details=False)
    early.run()
    assert float(early.best_fitness) >= -10.0


def test_sr_problem_modes_agree_and_generate_data():
    torch.manual_seed(0)
    prob = SymbolicRegression(func=lambda x: x[0] * x[1] - x[2], num_inputs=3, num_data=50, lower_bounds=-2, upper_bounds=2)
    assert prob.datapoints.shape == (50, 3) and prob.labels.shape == (50, 1)
    assert torch.allclose(prob.labels[:, 0], prob.datapoints[:, 0] * prob.datapoints[:, 1] - prob.datapoints[:, 2])
    f = Forest.random_generate(100, desc(), keys=torch.tensor([5, 6]))
    a = prob.evaluate(f)
    prob.execute_mode = "torch"
    b = prob.evaluate(f)
    ok = torch.isfinite(a) & torch.isfinite(b)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.allclose(a[ok], b[ok], rtol=1e-4)
    with pytest.raises(AssertionError):
        SymbolicRegression(datapoints=prob.datapoints, labels=prob.labels, execute_mode="fast please")


def test_tree_views_and_printing():
    t = Tree(3, 1, node_value=torch.tensor([3., 2., 0., 2., 2., 0., 2., 0.]), node_type=torch.tensor([3, 3, 0, 0, 3, 0, 0, 0], dtype=torch.int16),
             subtree_size=torch.tensor([7, 3, 1, 1, 3, 1, 1, 0], dtype=torch.int16))
    assert t.to_infix() == "((x0 - x2) * (x0 - x2))"
    assert str(t.to_sympy_expr()) in ("(x0 - x2)**2",)
    assert t.SR_fitness(torch.tensor([[0., 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1]]), torch.tensor([[0.], [1], [1], [0]])).item() == 0.5
    assert t.forward(torch.tensor([1.0, 0.0, 3.0])).tolist() == [4.0]
    assert t.to_forest().pop_size == 1


# ---- TournamentSelection against the reference's own operator (tests/golden/make_tournament_golden.py) ---------------------
_TGOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", ["t3_p1", "t7_p08_replace", "t4_p09_noreplace", "t5_p05_many_passes"])
def test_tournament_selection_replays_the_reference(case):
    """The reference's TournamentSelection (selection/tournament.py:59-133) was run with its draws logged; fed the same
    contenders and the same uniform numbers, ours must name the same survivors (repeats and order included) and elites."""
    import json

    from evogp_amd.algorithm.selection import TournamentSelection

    g = np.load(os.path.join(_TGOLD, f"tournament_{case}.npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    n, kw = meta["pop"], meta["kwargs"]
    sel = TournamentSelection(**kw)
    n_elite, n_surv = sel.counts(n)
    assert n_elite == len(g["elites"]) and n_surv == len(g["survivors"])
    per_pass, passes = sel.selector.passes(n, n_surv)
    assert g["contenders"].shape == (passes, per_pass * kw["tournament_size"])       # the reference's pass structure, :117-121
    fitness = torch.from_numpy(g["fitness"])
    contenders = torch.from_numpy(g["contenders"]).reshape(-1, kw["tournament_size"])[:n_surv]
    survivors = sel.selector.apply(fitness, contenders, torch.from_numpy(g["u"]))
    assert survivors.dtype == torch.int32 and survivors.tolist() == g["survivors"].tolist()
    elites, _ = sel(type("P", (), {"pop_size": n})(), fitness)
    assert elites.tolist() == g["elites"].tolist()
    # the operator's own draws have the reference's shape: every pass is a set of disjoint tournaments when replace=False
    c, u = sel.selector.draw(n, n_surv, "cpu")
    assert c.shape == (n_surv, kw["tournament_size"]) and int(c.min()) >= 0 and int(c.max()) < n
    if not kw["replace"]:
        for k in range(passes):
            block = c[k * per_pass:(k + 1) * per_pass].reshape(-1)
            assert block.unique().numel() == block.numel()


def test_forest_function_mask_follows_the_trees():
    """bit f of Forest.func_mask = function id f may occur; set from the descriptor, joined by the operators, unknown (0) for raw
    tensors, after an in-place edit, or as soon as one source is unknown"""
    d = desc()
    assert d.func_mask == 0b11110
    assert desc(using_funcs={"+": 1.0, "sin": 0.0, "if": 2.0}, max_layer_cnt=3).func_mask == 0b11          # zero weight: cannot be generated
    assert GenerateDescriptor(64, 3, 1, roulette_funcs=d.roulette_funcs, depth2leaf_probs=d.depth2leaf_probs, const_samples=[0.0]).func_mask == 0
    f = Forest.random_generate(30, d, keys=torch.tensor([1, 2]))
    g = Forest.random_generate(30, desc(using_funcs=["sin", "+"]), keys=torch.tensor([3, 4]))
    assert f.func_mask == 0b11110 and g.func_mask == (1 << 14) | 0b10
    assert (f + g).func_mask == f.func_mask | g.func_mask and f[3:9].func_mask == f.func_mask
    assert f.mutate(torch.zeros(30, dtype=torch.int32), g).func_mask == f.func_mask | g.func_mask
    raw = Forest(3, 1, f.batch_node_value.clone(), f.batch_node_type.clone(), f.batch_subtree_size.clone())
    assert raw.func_mask == 0 and (f + raw).func_mask == 0
    f[0:5] = g[0:5]
    assert f.func_mask == 0b11110 | (1 << 14)
    f.batch_node_value[0, 0] = 3.0
    assert f.func_mask == 0


def test_interpreter_generator_switches_still_generate():
    """gen/gen_tc_asm.py is the source of the interpreter; its A/B switches (docs/DESIGN_history_r01_r03.md section 3.1d, scripts/build_variant.sh) must keep
    producing a program for every build (K = 8, 4, 1; the three division modes), with the per-handler instruction counts bench.py
    reads.  (Assembling is the build's job: a handler that outgrows its 256-byte slot fails there.)"""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "evogp_amd", "csrc", "gen", "gen_tc_asm.py")
    spec = importlib.util.spec_from_file_location("gen_tc_asm_under_test", path)
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    defaults = {k: getattr(g, k) for k in ("DIVRANGE", "TRUST", "PKCONST", "PKARITH", "DIVABREAST", "DIVFIX", "EARLYREC", "L2WARM", "KWARM", "TRIGPK",
                                           "TOUCH", "RECGLC", "LIBPK")}
    try:
        for flip in [None] + list(defaults):
            for k, v in defaults.items():
                setattr(g, k, v)
            if flip:
                setattr(g, flip, not defaults[flip])
            for K, depth in ((8, 9), (4, 15), (1, 44)):
                for fast in (0, 1, 2):
                    info = {}
                    text = g.gen(K, depth, fast=fast, info=info)
                    assert "s_setpc_b64" in text and info["nhandlers"] == g.NHF
                    h = info["handlers"]
                    assert set(("end", "mul_SS", "divip_SS", "div_VV", "push_c")) <= set(h)
                    assert all(v["valu_clk"] >= v["valu"] for v in h.values())
                    packed = "v_pk_fma_f32" in text   # the range-tested division rows, (round 4) sin / cos / tan and (round 5) pow / sinh / cosh over row pairs
                    assert packed == ((g.DIVRANGE and fast in (1, 2) and K >= 2) or (g.TRIGPK and K >= 2) or (g.LIBPK and K >= 2)), (flip, K, fast)
                    assert ("Ltc_pow_fl1_row" in text) and (("v_pk_mov_b32" in text.split("Ltc_sinh_row")[1].split("s_cbranch_scc1")[0]) == (g.LIBPK and K >= 2))
                    assert "swap" in h and "Ltc_triglib_sin" in text
                    if K == 8:   # the build that runs ONE batch per entry and returns (sr_fused_kernel)
                        fused = g.gen(K, depth, fast=fast, fused=True)
                        assert "s_endpgm" not in fused and "Ltc_exit_here" in fused and "%[t0n]" in fused
                        assert ("glc" in fused) == g.RECGLC and ("%[taddr]" in fused.split("asm volatile")[1].split(": [lensn]")[0]) == g.TOUCH
    finally:
        for k, v in defaults.items():
            setattr(g, k, v)


def test_packed_compiler_plan_shares_out_more_trees_per_pass_than_consecutive_packing():
    """DESIGN.md section 3.1's figures for tc_compile_packed_kernel's plan, restated on the host: a wave takes 32 consecutive trees,
    orders them by length / 8 (stable) and fills a pass with the largest tree that is left, then the largest that still fits the 64
    lanes -- 2.15 trees per pass on the headline forest's lengths, against 1.76 for consecutive trees and a bound of 64 / mean."""
    from oracle.pyoracle import Oracle, depth2leaf, roulette_uniform

    lens = Oracle("port").generate(32000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])[2][:, 0].astype(int)

    def consecutive(batch):
        passes, used = 1, 0
        for n in batch:
            if used + n > 64:
                passes, used = passes + 1, 0
            used += n
        return passes

    def largest_first(batch):
        left = sorted(batch, key=lambda n: -(n >> 3))   # (python's sort is stable, like the kernel's radix rounds)
        passes = 0
        while left:
            passes, used, i = passes + 1, 0, 0
            while i < len(left):
                if used + left[i] <= 64:
                    used += left.pop(i)
                else:
                    i += 1
        return passes

    batches = [list(lens[i:i + 32]) for i in range(0, len(lens), 32)]
    per_pass_consecutive = len(lens) / sum(consecutive(b) for b in batches)
    per_pass_planned = len(lens) / sum(largest_first(b) for b in batches)
    assert 1.70 < per_pass_consecutive < 1.82, per_pass_consecutive
    assert 2.10 < per_pass_planned < 64 / lens.mean(), (per_pass_planned, 64 / lens.mean())


def test_pmc_json_takes_the_headline_compilers_counters_not_the_general_compilers(tmp_path):
    """scripts/pmc_json.py on the round's own PMC tables: the program compiler's traffic must come from the packed kernel's rows (a
    substring match once picked tc_compile_general_kernel's four empty launches: 11 KB instead of 640 MB in `call_hbm_bytes`)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    tables = [os.path.join(prof, f"r04Z_05_pmc{i}.md") for i in (1, 2, 3)]
    if not all(os.path.exists(t) for t in tables):
        pytest.skip("no PMC tables of this round in profiles/")
    line = tmp_path / "bench.log"
    line.write_text(json.dumps({"metric": "tree_evals_per_s", "config": {"pop_per_gpu": 1000000}}) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "pmc_json.py"), *tables, str(line)], capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert "tc_compile_packed_kernel" in d["kernels"]["tc_compile_kernel"]["rocprof_name"]
    assert d["tc_compile_kernel_hbm_bytes_per_launch"] > 3e8 and d["sr_tc_kernel_hbm_bytes_per_launch"] > 2e8
    assert abs(d["call_hbm_bytes"] - d["tc_compile_kernel_hbm_bytes_per_launch"] - d["sr_tc_kernel_hbm_bytes_per_launch"]) < 1.0


def test_next_kept_lane_by_a_carry():
    """csrc/sr_tc.hip next_word_has_variable: the word that runs after a lane's is that of the nearest kept lane BELOW it; with K the kept
    lanes and U those of them whose word has a variable operand, R = (~K + (U << 1)) & K marks the kept lanes whose predecessor (the next
    kept lane below) is in U -- a carry that starts one above a lane of U runs through the lanes that are not kept and stops at the next
    kept one.  Restated here with Python integers against the definition, on random masks."""
    rng = np.random.default_rng(5)
    M = (1 << 64) - 1
    for _ in range(2000):
        K = int(rng.integers(0, 2**63)) | (int(rng.integers(0, 2)) << 63)
        K &= int(rng.integers(0, 2**63)) | (int(rng.integers(0, 2)) << 63) if rng.random() < 0.5 else M   # sparse and dense masks
        U = K & (int(rng.integers(0, 2**63)) | (int(rng.integers(0, 2)) << 63))
        R = (((~K & M) + ((U << 1) & M)) & M) & K
        want, below = 0, None   # below: the nearest kept lane under the current one
        for lane in range(64):
            if (K >> lane) & 1:
                if below is not None and (U >> below) & 1:
                    want |= 1 << lane
                below = lane
        assert R == want, (hex(K), hex(U), hex(R), hex(want))


def test_classifier_hit_rule_on_running_maxima():
    """gen_tc_asm.py endcls_body: with P_j = max(out_0 .. out_j) and M = P_(n-1), a row whose label is L is a hit
    iff P_L == M and P_(L-1) < M - d, and ambiguous iff P_L >= M - d and P_(L-1) < M and it is no hit; rows with a NaN or an infinite maximum
    are hits iff L == 0.  Against the definition it replaces -- arg = first index of the maximum (0 for such rows); ambiguous when an output
    IN FRONT of arg lies within d of the maximum -- the rule must agree wherever it says "hit" or "miss", and may only say "ambiguous"
    more often (those trees are recounted exactly)."""
    rng = np.random.default_rng(11)
    d = np.float32(1.25 * 2.0 ** -23)
    for _ in range(4000):
        n = int(rng.integers(2, 11))
        x = rng.choice(np.array([-2.0, -1.0, 0.0, 0.5, 1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0, np.inf, -np.inf, np.nan], np.float32), n,
                       p=[0.12, 0.12, 0.12, 0.12, 0.2, 0.1, 0.1, 0.08, 0.015, 0.015, 0.01]).astype(np.float32)
        L = int(rng.integers(0, n))
        special = bool(np.isnan(x).any() or np.isinf(np.nanmax(x) if not np.isnan(x).all() else np.nan))
        # the definition
        if special:
            arg, amb_def = 0, False
        else:
            M = x.max()
            arg = int(np.argmax(x))                      # first index of the maximum
            amb_def = bool((x[:arg] >= M - d).any())     # an output in front of it within d (it is below M there by the choice of arg)
        # the rule on running maxima
        with np.errstate(invalid="ignore"):
            P = np.fmax.accumulate(x)                    # v_max_f32 drops NaN operands
        if special:
            hit, amb = L == 0, False
        else:
            M = P[-1]
            thr = np.float32(M - d)
            before = P[L - 1] if L > 0 else np.float32(-np.inf)
            hit = bool(P[L] == M and before < thr)
            amb = bool(P[L] >= thr and before < M and not hit)
        if amb:
            continue                                     # (recounted with torch's arithmetic)
        assert not (amb_def and hit), (x, L)             # a hit is never claimed where the definition wavers about the label's class
        if not amb_def:
            assert hit == (arg == L), (x, L, arg, hit)
        else:   # the definition wavers between arg and the earlier outputs within d of it: the label is none of them (else the rule said so)
            assert not hit and L not in set(np.flatnonzero(x[:arg + 1] >= M - d).tolist()), (x, L, arg)


def test_division_through_a_rounded_reciprocal_is_faithful():
    """gen_tc_asm.py divr_* (DESIGN.md section 3.1): q = a * r, e = fma(-q, x, a), q' = fma(e, r, q) with r the correctly rounded 1 / x read
    from LDS.  In float64 arithmetic rounded to float32 where the device rounds (a product of two floats is exact in float64): the result
    is the correctly rounded quotient almost always and never more than one ulp away, for operands in the range the handler's rows assume;
    without the correction (the first build of this round) one pair in three is off and errors reach 1.5 ulp."""
    rng = np.random.default_rng(3)
    n = 1_000_000
    def operands():
        m = rng.uniform(1.0, 2.0, n).astype(np.float32)
        e = rng.integers(-46, 46, n)
        return (np.ldexp(m, e) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    a, x = operands(), operands()
    f32 = lambda v: v.astype(np.float32)
    r = f32(1.0 / x.astype(np.float64))
    q = f32(a.astype(np.float64) * r.astype(np.float64))
    e = f32(-q.astype(np.float64) * x.astype(np.float64) + a.astype(np.float64))
    qc = f32(e.astype(np.float64) * r.astype(np.float64) + q.astype(np.float64))
    want = f32(a.astype(np.float64) / x.astype(np.float64))
    ulps = lambda got: np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulps(qc).max() <= 1 and (ulps(qc) == 0).mean() > 0.9999, (ulps(qc).max(), (ulps(qc) == 0).mean())
    assert (ulps(q) != 0).mean() > 0.2 and ulps(q).max() >= 1      # (what the correction is for)
