"""GPU: the reference's Python surface on top of the HIP engine — torch.ops.evogp_cuda.*, Forest,
GenerateDescriptor, Default{Selection,Crossover,Mutation}, GeneticProgramming, SymbolicRegression,
StandardPipeline — exercised the way the reference's own scripts do (test/test_bind_success.py,
test/fix_bug.py, example/basic.py, src/evogp/sr_test.py)."""
import pickle

import numpy as np
import pytest

from helpers import assert_close_classes, bits, c2_dataset, depth2leaf, roulette_uniform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available()
    import evogp_amd  # noqa: F401

    return torch


def _desc_tensors(torch, mlc=3):
    keys = torch.tensor([42, 0], dtype=torch.uint32, device="cuda")
    d2l = torch.tensor([0.1] * (mlc - 1) + [1.0] * (10 - (mlc - 1)), dtype=torch.float32, device="cuda")
    rou = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0] + [1.0] * 24, dtype=torch.float32, device="cuda")
    cs = torch.tensor([-1.0, 0.0, 1.0], dtype=torch.float32, device="cuda")
    return keys, d2l, rou, cs


def test_bind_success_script_flow(torch_mod, oracle):
    """test/test_bind_success.py: generate -> crossover -> evaluate -> SR fitness through torch.ops."""
    torch = torch_mod
    keys, d2l, rou, cs = _desc_tensors(torch)
    v, t, s = torch.ops.evogp_cuda.tree_generate(2, 64, 2, 1, 3, 0.3, 0.5, keys, d2l, rou, cs)
    assert v.dtype == torch.float32 and t.dtype == torch.int16 and s.dtype == torch.int16 and v.shape == (2, 64)
    assert t[0, :7].tolist() == [3, 3, 0, 1, 3, 0, 1] and s[1, :7].tolist() == [7, 3, 1, 1, 3, 1, 1]  # Appendix B1
    res = torch.ops.evogp_cuda.tree_evaluate(2, 64, 2, 1, v, t, s, torch.tensor([[1.0, 2.0], [3.0, 4.0]], device="cuda"))
    assert res.ravel().tolist() == [3.0, 1.0]
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device="cuda")
    cv, ct, cs_ = torch.ops.evogp_cuda.tree_crossover(2, 1, 64, v, t, s, i32([0]), i32([1]), i32([2]), i32([4]))
    assert cs_[0, :9].tolist() == [9, 5, 3, 1, 1, 1, 3, 1, 1]
    v1, t1, s1 = torch.ops.evogp_cuda.tree_generate(2, 64, 1, 1, 3, 0.3, 0.5, keys, d2l, rou, cs)
    fit = torch.ops.evogp_cuda.tree_SR_fitness(2, 2, 64, 1, 1, True, v1, t1, s1, torch.tensor([[1.0], [2.0]], device="cuda"),
                                               torch.tensor([[1.0], [3.0]], device="cuda"), 0)
    assert fit.tolist() == [0.5, 2.0]  # Appendix B2
    mv, mt, ms = torch.ops.evogp_cuda.tree_mutate(2, 64, v, t, s, i32([1, 0]), v.flip(0).contiguous(), t.flip(0).contiguous(), s.flip(0).contiguous())
    assert ms[1, 0].item() == 7 and ms[0, 0].item() == 11


def test_argument_errors_are_runtime_errors(torch_mod):
    torch = torch_mod
    keys, d2l, rou, cs = _desc_tensors(torch)
    with pytest.raises(RuntimeError, match="gp_len must be in range"):
        torch.ops.evogp_cuda.tree_generate(2, 2000, 2, 1, 3, 0.3, 0.5, keys, d2l, rou, cs)
    with pytest.raises(RuntimeError, match="pop_size must be larger than 0"):
        torch.ops.evogp_cuda.tree_generate(0, 64, 2, 1, 3, 0.3, 0.5, keys, d2l, rou, cs)
    with pytest.raises(RuntimeError, match="roulette_funcs must have shape"):
        torch.ops.evogp_cuda.tree_generate(2, 64, 2, 1, 3, 0.3, 0.5, keys, d2l, rou[:24].contiguous(), cs)
    with pytest.raises(RuntimeError, match="out_prob must be in range"):
        torch.ops.evogp_cuda.tree_generate(2, 64, 2, 1, 3, 1.3, 0.5, keys, d2l, rou, cs)
    v, t, s = torch.ops.evogp_cuda.tree_generate(4, 64, 2, 1, 3, 0.3, 0.5, keys, d2l, rou, cs)
    with pytest.raises(RuntimeError, match="contiguous CUDA tensor"):
        torch.ops.evogp_cuda.tree_evaluate(4, 64, 2, 1, v, t, s, torch.zeros(2, 4, device="cuda").t())
    with pytest.raises(RuntimeError, match="variables must have shape"):
        torch.ops.evogp_cuda.tree_evaluate(4, 64, 2, 1, v, t, s, torch.zeros(3, 2, device="cuda"))


def test_fix_bug_tree_known_answer(torch_mod):
    """test/fix_bug.py: (x0-x2)*(x0-x2) on four XOR rows -> MSE 0.5 in both execution modes."""
    torch = torch_mod
    from evogp_amd.tree import Tree

    tree = Tree(3, 1,
                node_type=torch.tensor([3, 3, 0, 0, 3, 0, 0, 0], dtype=torch.int16, device="cuda"),
                node_value=torch.tensor([3., 2., 0., 2., 2., 0., 2., 0.], dtype=torch.float32, device="cuda"),
                subtree_size=torch.tensor([7, 3, 1, 1, 3, 1, 1, 0], dtype=torch.int16, device="cuda"))
    X = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1]], dtype=torch.float, device="cuda")
    y = torch.tensor([[0], [1], [1], [0]], dtype=torch.float, device="cuda")
    assert tree.SR_fitness(X, y, execute_mode="hybrid parallel").item() == 0.5
    assert tree.SR_fitness(X, y, execute_mode="data parallel").item() == 0.5
    assert tree.forward(X).ravel().tolist() == [0.0, 1.0, 0.0, 1.0]
    assert tree.forward(X[1]).tolist() == [1.0]
    assert "x0" in str(tree) and "x2" in str(tree)


def test_forest_api_matches_oracle(torch_mod, oracle):
    torch = torch_mod
    from evogp_amd.tree import Forest, GenerateDescriptor, Tree

    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"],
                              max_layer_cnt=6, const_samples=[-1, 0, 1])
    keys = torch.tensor([42, 0], dtype=torch.uint32, device="cuda")
    forest = Forest.random_generate(2000, desc, keys=keys)
    want = oracle.generate(2000, 64, 10, 1, 0.5, 0.5, [42, 0], depth2leaf(6), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    assert np.array_equal(forest.batch_subtree_size.cpu().numpy(), want[2])
    assert np.array_equal(bits(forest.batch_node_value.cpu().numpy()), bits(want[0]))
    X, y = c2_dataset()
    fit = forest.SR_fitness(torch.from_numpy(X), torch.from_numpy(y))
    assert_close_classes(fit.cpu().numpy(), oracle.sr_fitness(*want, X, y), 1e-5, what="Forest.SR_fitness")
    # forward (one row per tree) and batch_forward (shared rows) agree with each other and the oracle
    xb = torch.from_numpy(X[:5]).cuda()
    bf = forest.batch_forward(xb)
    assert bf.shape == (2000, 5, 1)
    fw = forest.forward(xb[2:3].repeat(2000, 1))
    assert torch.equal(torch.nan_to_num(bf[:, 2, :], nan=7.0), torch.nan_to_num(fw, nan=7.0))
    # container protocol
    assert isinstance(forest[3], Tree) and len(forest[10:20]) == 10 and len(forest + forest[0]) == 2001
    mask = torch.zeros(2000, dtype=torch.bool); mask[::7] = True
    assert len(forest[mask]) == int(mask.sum())
    again = pickle.loads(pickle.dumps(forest[:50]))
    assert torch.equal(again.batch_node_type, forest.batch_node_type[:50])
    z = Forest.zero_generate(4, 16, 3, 1)
    assert z.forward(torch.ones(4, 3)).ravel().tolist() == [0.0] * 4


def test_basic_example_pipeline_runs_and_improves(torch_mod):
    """example/basic.py scaled to a few generations: XOR-3d SR with the default operator set."""
    torch = torch_mod
    from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
    from evogp_amd.pipeline import StandardPipeline
    from evogp_amd.problem import SymbolicRegression
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(0)
    X = torch.tensor([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=torch.float, device="cuda")
    y = (X.sum(1) % 2)[:, None]
    problem = SymbolicRegression(datapoints=X, labels=y)
    desc = GenerateDescriptor(max_tree_len=64, input_len=problem.problem_dim, output_len=problem.solution_dim,
                              using_funcs=["+", "-", "*", "/"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    algo = GeneticProgramming(initial_forest=Forest.random_generate(pop_size=5000, descriptor=desc),
                              crossover=DefaultCrossover(), mutation=DefaultMutation(0.2, desc.update(max_layer_cnt=3)),
                              selection=DefaultSelection(survival_rate=0.3, elite_rate=0.01), enable_pareto_front=True)
    pipe = StandardPipeline(algo, problem, generation_limit=15, is_show_details=False)
    first = float(torch.nan_to_num(problem.evaluate(algo.forest), nan=-1e9).max())
    best = pipe.run()
    assert algo.forest.pop_size == 5000
    assert float(pipe.best_fitness) >= first  # elitism: never worse
    assert float(pipe.best_fitness) > -0.5    # better than any constant the leaf set offers (0, 1, -1)
    pred = best.forward(X)
    assert pred.shape == (8, 1)
    # every tree of the final population is structurally valid
    from oracle.pyoracle import Oracle
    o = Oracle("port")
    tt, ss = algo.forest.batch_node_type.cpu().numpy(), algo.forest.batch_subtree_size.cpu().numpy()
    assert all(o.validate_tree(tt[i], ss[i]) == 0 for i in range(0, 5000, 13))


def test_sr_installation_scenario(torch_mod):
    """src/evogp/sr_test.py shape: pop 1000, 1000 datapoints, func-generated data, torch and kernel modes agree."""
    torch = torch_mod
    from evogp_amd.problem import SymbolicRegression
    from evogp_amd.tree import Forest, GenerateDescriptor

    torch.manual_seed(0)
    prob = SymbolicRegression(func=lambda x: (x[0] + x[1]) ** 2, num_inputs=2, num_data=1000, lower_bounds=-5, upper_bounds=5)
    assert prob.datapoints.shape == (1000, 2) and prob.labels.shape == (1000, 1)
    desc = GenerateDescriptor(max_tree_len=64, input_len=2, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5,
                              const_range=(-1, 1), sample_cnt=8)
    forest = Forest.random_generate(1000, desc)
    a = prob.evaluate(forest)
    prob.execute_mode = "torch"
    b = prob.evaluate(forest)
    assert_close_classes(a.cpu().numpy(), b.cpu().numpy(), 1e-4, what="kernel vs torch mode")


def test_program_record_buffer_is_capped_and_released():
    import torch

    """the engine-owned program-record buffer of tree_SR_fitness (include/evogp_hip.h): its size follows the documented law, a cap
    sends the call to the register interpreters with the same results, release frees it and the next call allocates again"""
    import evogp_amd
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    pop = 30_000
    f = Forest.random_generate(pop, desc, keys=torch.tensor([3, 4], dtype=torch.uint32, device=dev))
    X = torch.rand(512, 4, device=dev) * 4 - 2
    y = (X[:, 0] - X[:, 1] * X[:, 2]).unsqueeze(1)
    evogp_amd.release_workspaces()
    assert evogp_amd.program_buffer_bytes() == 0 and evogp_amd.record_ring_bytes() == 0
    a = f.SR_fitness(X, y)
    # (an eager call also leaves ONE record ring behind for calls that arrive inside a graph capture: include/evogp_hip.h)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert evogp_amd.record_ring_bytes() in (0, cus * 16 * 2 * 16 * 256), evogp_amd.record_ring_bytes()
    array = (pop * 256 + 4095) // 4096 * 4096
    law = array                       # include/evogp_hip.h: ONE array of records for a forest whose function mask holds no unary function
    held = evogp_amd.program_buffer_bytes()
    assert law <= held <= law + law // 8 + 8 * 256, (held, law)
    # the same trees without a mask (a forest built from raw tensors, the reference's own operator): the engine looks at the forest
    # (round 6: include/evogp_hip.h, tests/test_gpu_learned.py) and still holds ONE array
    plain = Forest(f.input_len, f.output_len, f.batch_node_value, f.batch_node_type, f.batch_subtree_size)
    assert torch.equal(plain.SR_fitness(X, y).view(torch.int32), a.view(torch.int32))
    assert evogp_amd.program_buffer_bytes() == held
    # ... and a forest with unary functions: programs may have up to 64 words, three arrays at gp_len 64
    du = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/", "sin"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    fu = Forest.random_generate(pop, du, keys=torch.tensor([3, 4], dtype=torch.uint32, device=dev))
    fu.SR_fitness(X, y)
    law3 = array * max(2, (64 + 2 + 30) // 31)
    assert law3 <= evogp_amd.program_buffer_bytes() <= law3 + law3 // 8 + 8 * 256, (evogp_amd.program_buffer_bytes(), law3)
    evogp_amd.release_workspaces()
    assert evogp_amd.program_buffer_bytes() == 0 and evogp_amd.record_ring_bytes() == 0
    try:
        evogp_amd.set_program_buffer_limit(1 << 20)        # far below the law: the compiled path is not eligible
        b = f.SR_fitness(X, y)
        assert evogp_amd.program_buffer_bytes() == 0
    finally:
        evogp_amd.set_program_buffer_limit(16 << 30)
    c = f.SR_fitness(X, y)
    assert evogp_amd.program_buffer_bytes() == held
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    ok = torch.isfinite(a)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.allclose(a[ok], b[ok], rtol=1e-5, atol=0)


def test_function_mask_skips_launches_but_never_changes_a_result():
    """A Forest remembers the function set of the descriptors its trees came from and tree_SR_fitness then leaves out the launches
    such a forest cannot need (include/evogp_hip.h evogp_hip_sr_fitness_hinted: no general compiler on + - * /).  The mask decides
    which kernels run, never what they compute: bit-identical fitness where no tree needs a follow-up kernel, within the contract
    where one does (a 40-entry operand stack), and a mask that promises too much (max / if / sin in a forest declared + - * /) is
    caught by the last follow-up kernel."""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.tree import Forest, GenerateDescriptor
    from oracle.pyoracle import Oracle

    dev = torch.device("cuda", 0)
    oracle = Oracle("port")
    X, y = c2_dataset()
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    # (the engine honours the mask only under EVOGP_TC_FUNC_MASK=1, read once per process: whichever way this process reads it,
    # the results below must hold)

    def raw(f):   # the same rows without any knowledge attached
        return Forest(f.input_len, f.output_len, f.batch_node_value.clone(), f.batch_node_type.clone(), f.batch_subtree_size.clone())

    desc = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
    f = Forest.random_generate(30_000, desc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev))
    assert f.func_mask == 0b11110 and raw(f).func_mask == 0
    a, b = f.SR_fitness(Xd, yd), raw(f).SR_fitness(Xd, yd)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # handed on by the operators, dropped by an in-place edit
    child = f.crossover(*(torch.zeros(100, dtype=torch.int32, device=dev) for _ in range(4)))
    assert child.func_mask == f.func_mask and (f[:10] + f[10:20]).func_mask == f.func_mask
    # a tree that needs a follow-up kernel: a left-deep chain of 31 subtractions (operand stack of 32 entries)
    g = raw(f); g._func_mask = (0b11110, g._forest_key())
    n = 63; k = (n - 1) // 2
    v, t, s = g.batch_node_value, g.batch_node_type, g.batch_subtree_size
    v[7] = 0; t[7] = 0; s[7] = 0
    t[7, :k] = 3; v[7, :k] = 2.0; s[7, :k] = (n - 2 * torch.arange(k, device=dev)).to(torch.int16)
    t[7, k:n] = 0; v[7, k:n] = (torch.arange(n - k, device=dev) % 10).float(); s[7, k:n] = 1
    g._func_mask = (0b11110, g._forest_key())                     # (the edits above dropped it)
    want = oracle.sr_fitness(v.cpu().numpy(), t.cpu().numpy(), s.cpu().numpy(), X, y)
    for forest in (g, raw(g)):
        assert_close_classes(forest.SR_fitness(Xd, yd).cpu().numpy(), want, 1e-5, what="deep tree")
    # a mask that promises too much
    wild = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "*", "max", "if", "sin"], max_layer_cnt=4, const_samples=[-1, 0, 1])
    h = Forest.random_generate(5000, wild, keys=torch.tensor([5, 0], dtype=torch.uint32, device=dev))
    honest = h.SR_fitness(Xd, yd)
    lying = raw(h); lying._func_mask = (0b11110, lying._forest_key())
    got = lying.SR_fitness(Xd, yd).cpu().numpy()
    ok = np.isfinite(honest.cpu().numpy())
    assert np.array_equal(np.isnan(got), np.isnan(honest.cpu().numpy())) and np.allclose(got[ok], honest.cpu().numpy()[ok], rtol=1e-4)


def test_record_memory_is_visible_to_torchs_allocator():
    """The engine's program-record buffer comes out of torch's caching allocator when the libtorch binding is loaded (include/evogp_hip.h
    evogp_hip_set_allocator): torch.cuda.memory_allocated() rises by what evogp_amd.program_buffer_bytes() reports and falls again after
    release_workspaces (256-768 MB that torch's statistics did not show)."""
    import torch

    import evogp_amd
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    desc = GenerateDescriptor(max_tree_len=64, input_len=4, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=5, const_samples=[-1, 0, 1])
    X = torch.rand(300, 4, device=dev); y = torch.rand(300, 1, device=dev)
    evogp_amd.release_workspaces()
    forest = Forest.random_generate(200_000, desc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    fit = forest.SR_fitness(X, y)
    torch.cuda.synchronize()
    held = evogp_amd.program_buffer_bytes()
    assert held >= 200_000 * 256
    grown = torch.cuda.memory_allocated() - before - fit.numel() * 4
    assert grown >= held, (grown, held)
    del fit
    evogp_amd.release_workspaces()
    assert evogp_amd.program_buffer_bytes() == 0 and torch.cuda.memory_allocated() <= before + 4096


def test_fitness_scores_is_the_sign_and_the_nan_scrub_of_a_generation():
    """evogp_hip::fitness_scores = SymbolicRegression.evaluate's sign + StandardPipeline.step's NaN -> -inf (pipeline/standard.py:41-43)
    in one launch; SymbolicRegression.scores and the pipeline's step use it"""
    import torch

    import evogp_amd  # noqa: F401
    from evogp_amd.problem import SymbolicRegression
    from evogp_amd.tree import Forest, GenerateDescriptor

    dev = torch.device("cuda", 0)
    e = torch.tensor([1.5, float("nan"), 0.0, float("inf"), -2.0, float("-inf")], device=dev)
    for negate in (True, False):
        got = torch.ops.evogp_hip.fitness_scores(e, negate)
        f = -e if negate else e
        want = torch.where(torch.isnan(f), torch.full_like(f, float("-inf")), f)
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    desc = GenerateDescriptor(max_tree_len=32, input_len=3, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=4, const_samples=[-1, 0, 1])
    f = Forest.random_generate(3000, desc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
    X = torch.rand(64, 3, device=dev) * 2 - 1
    prob = SymbolicRegression(datapoints=X, labels=(X[:, :1] * X[:, 1:2]).contiguous())
    ev = prob.evaluate(f)
    want = torch.where(torch.isnan(ev), torch.full_like(ev, float("-inf")), ev)
    assert torch.isnan(ev).any() and torch.equal(prob.scores(f).view(torch.int32), want.view(torch.int32))
