"""GPU: the reference's import root and operator namespace, used the way its scripts use them.

A script written against EMI-Group/evogp imports ``evogp.tree / evogp.algorithm / evogp.problem / evogp.pipeline`` and
reaches the kernels through ``torch.ops.evogp_cuda.*`` (registered from C++ by libevogp_torch.so).  These tests run such a
script in a FRESH interpreter — nothing of ``evogp_amd`` is named — and check the binding layer itself."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

PARITY_SCRIPT = textwrap.dedent("""
    import torch
    import evogp.evogp_cuda                                   # what the reference's tree/__init__.py does
    from evogp.tree import Forest, GenerateDescriptor
    from evogp.algorithm import GeneticProgramming, DefaultSelection, DefaultMutation, DefaultCrossover
    from evogp.problem import SymbolicRegression
    from evogp.pipeline import StandardPipeline

    torch.manual_seed(3)
    X = torch.tensor([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], device="cuda")
    y = (X.sum(1) % 2)[:, None].contiguous()                  # 3-input parity
    problem = SymbolicRegression(datapoints=X, labels=y)
    desc = GenerateDescriptor(max_tree_len=32, input_len=problem.problem_dim, output_len=problem.solution_dim,
                              using_funcs=["+", "-", "*", "/"], max_layer_cnt=4, const_samples=[-1, 0, 1])
    algo = GeneticProgramming(initial_forest=Forest.random_generate(pop_size=2000, descriptor=desc),
                              crossover=DefaultCrossover(),
                              mutation=DefaultMutation(mutation_rate=0.2, descriptor=desc.update(max_layer_cnt=3)),
                              selection=DefaultSelection(survival_rate=0.3, elite_rate=0.01))
    pipe = StandardPipeline(algo, problem, generation_limit=12, is_show_details=False)
    first = float(pipe.step().max())
    best = pipe.run()
    import sys
    assert "evogp.tree" in sys.modules and type(best).__module__.startswith("evogp")
    assert float(pipe.best_fitness) >= first, (float(pipe.best_fitness), first)
    out = best.forward(X[:1])
    assert out.shape == (1, 1)
    print("DROPIN_OK", first, float(pipe.best_fitness))
""")


def test_reference_style_script_runs_unmodified():
    r = subprocess.run([sys.executable, "-c", PARITY_SCRIPT], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert "DROPIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_ops_are_registered_from_cpp_not_python():
    import evogp_amd  # noqa: F401

    loaded = [os.path.basename(p) for p in torch.ops.loaded_libraries]
    assert "libevogp_torch.so" in loaded
    # a C++ kernel registration has no Python function behind the dispatch key
    assert torch._C._dispatch_has_kernel_for_dispatch_key("evogp_cuda::tree_SR_fitness", "CUDA")


def test_binding_rejects_bad_arguments_like_the_reference_wrapper():
    import evogp_amd  # noqa: F401

    dev = "cuda"
    v = torch.zeros(4, 8, device=dev); t = torch.zeros(4, 8, dtype=torch.int16, device=dev); s = torch.ones(4, 8, dtype=torch.int16, device=dev)
    x = torch.zeros(4, 2, device=dev)
    with pytest.raises(RuntimeError, match="gp_len"):
        torch.ops.evogp_cuda.tree_evaluate(4, 2000, 2, 1, v, t, s, x)
    with pytest.raises(RuntimeError, match="shape"):
        torch.ops.evogp_cuda.tree_evaluate(4, 8, 3, 1, v, t, s, x)
    with pytest.raises(RuntimeError, match="scalar type"):
        torch.ops.evogp_cuda.tree_evaluate(4, 8, 2, 1, v, t.to(torch.int32), s, x)
    with pytest.raises(RuntimeError, match="contiguous"):
        torch.ops.evogp_cuda.tree_evaluate(4, 8, 2, 1, v, t, s, torch.zeros(2, 4, device=dev).t())
    with pytest.raises(RuntimeError, match="kernel_type"):
        torch.ops.evogp_cuda.tree_SR_fitness(4, 4, 8, 2, 1, True, v, t, s, torch.zeros(4, 2, device=dev), torch.zeros(4, 1, device=dev), 7)
    # (that CPU tensors are rejected -- no CPU implementation, no fallback -- is checked in a fresh interpreter by
    # tests/test_capi.py: other test modules register the TEST-ONLY oracle-backed CPU ops in this process)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_operands_on_different_devices_are_rejected():
    import evogp_amd  # noqa: F401

    v = torch.zeros(4, 8, device="cuda:1"); t = torch.zeros(4, 8, dtype=torch.int16, device="cuda:1"); s = torch.ones(4, 8, dtype=torch.int16, device="cuda:1")
    with pytest.raises(RuntimeError, match="share a device"):
        torch.ops.evogp_cuda.tree_evaluate(4, 8, 2, 1, v, t, s, torch.zeros(4, 2, device="cuda:0"))
