"""CPU: the C-ABI shared library loads and exports every symbol include/evogp_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="evogp_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evogp_hip_\w+)\s*\(", text)))


def test_header_declares_the_five_reference_entry_points():
    syms = declared_symbols()
    for name in ("generate", "mutate", "crossover", "evaluate", "sr_fitness"):
        assert f"evogp_hip_{name}" in syms


def test_library_exports_every_declared_symbol():
    from evogp_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"libevogp_hip.so does not export {sym}"
    assert set(_lib.PROTOTYPES) | {"evogp_hip_error_string", "evogp_hip_evaluate_workspace_bytes", "evogp_hip_select_workspace_bytes",
                                  "evogp_hip_program_buffer_bytes", "evogp_hip_record_ring_bytes"} == set(declared_symbols())
    assert lib.evogp_hip_abi_version() == _lib.ABI_VERSION
    # the measurement and test hooks live in a header of their own: the boundary a maintainer of the reference binds holds none of them
    assert not [s for s in declared_symbols() if "debug" in s or "timer" in s]
    for sym in declared_symbols("evogp_hip_debug.h"):
        assert hasattr(lib, sym), f"libevogp_hip.so does not export {sym}"
    assert set(_lib.DEBUG_PROTOTYPES) == set(declared_symbols("evogp_hip_debug.h"))


def test_error_strings_and_argument_errors_without_gpu():
    from evogp_amd import _lib

    assert b"success" in _lib.lib.evogp_hip_error_string(0)
    # argument validation happens on the host before any launch: callable without a GPU
    rc = _lib.lib.evogp_hip_sr_fitness(0, 8, 32, 3, 1, 1, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"out of range" in _lib.lib.evogp_hip_error_string(rc)
    rc = _lib.lib.evogp_hip_generate(4, 2000, 3, 1, 3, 0.5, 0.5, None, None, None, None, None, None, None, 0, None)
    assert rc == -1
    rc = _lib.lib.evogp_hip_generate(4, 32, 3, 1, 3, 0.5, 0.5, None, None, None, None, None, None, None, 0, None)
    assert rc == -2
    rc = _lib.lib.evogp_hip_crossover(4, 4, 0, None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1
    # the entry points without a counterpart in the reference validate the same way
    assert _lib.lib.evogp_hip_select(0, 0, 1, None, None, None, None) == -1          # empty vector
    assert _lib.lib.evogp_hip_select(10, 5, 4, None, None, None, None) == -1         # more elites than kept trees
    assert _lib.lib.evogp_hip_select(10, 1, 11, None, None, None, None) == -1        # more kept trees than trees
    assert _lib.lib.evogp_hip_select(10, 1, 3, None, None, None, None) == -2         # null pointers
    assert _lib.lib.evogp_hip_select_workspace_bytes() > 0
    assert _lib.lib.evogp_hip_batch_argmax_count(4, 8, 32, 3, 1, None, None, None, None, None, None, None) == -1   # one output: no arg-max
    assert _lib.lib.evogp_hip_batch_argmax_count(4, 8, 32, 3, 2, None, None, None, None, None, None, None) == -2
    assert _lib.lib.evogp_hip_evaluate_prepare(4, 32, 3, 1, None, None, None, None, 0, None) == -1               # prepared lists are for multi-output forests
    # the breeding pass with separate elite / parent lists
    assert _lib.lib.evogp_hip_breed_lists(10, 10, 0, 1, 3, *([None] * 6), 0, *([None] * 7), 0, 10, None) == -1       # gp_len 0
    assert _lib.lib.evogp_hip_breed_lists(10, 10, 8, 1, 0, *([None] * 6), 0, *([None] * 7), 0, 10, None) == -1       # no parent
    assert _lib.lib.evogp_hip_breed_lists(10, 10, 8, 1, 30, *([None] * 6), 0, *([None] * 7), 0, 10, None) == -2      # null pointers (more parents than trees is legal)
    # the engine-owned program-record buffer: nothing held before the first fitness call; the cap is a plain setter
    assert _lib.lib.evogp_hip_set_program_buffer_limit(1 << 34) == 0
    # which program compiler a fitness call uses (tests, A/B): -1 packed with the batch by population, 0 one tree per pass, up to 64
    assert _lib.lib.evogp_hip_debug_compile_batch(65) == -1 and _lib.lib.evogp_hip_debug_compile_batch(-2) == -1
    assert _lib.lib.evogp_hip_debug_compile_batch(0) == 0 and _lib.lib.evogp_hip_debug_compile_batch(32) == 0
    assert _lib.lib.evogp_hip_debug_compile_batch(-1) == 0


def test_ops_are_registered_with_reference_schemas():
    import torch

    import evogp_amd  # noqa: F401

    s = str(torch.ops.evogp_cuda.tree_SR_fitness.default._schema)
    assert "int i1, int i2, int i3, int i4, int i5, bool b1" in s and "int i6" in s
    for op in ("tree_generate", "tree_mutate", "tree_crossover", "tree_evaluate", "tree_SR_fitness"):
        assert hasattr(torch.ops.evogp_cuda, op)


def test_product_registers_no_cpu_implementation():
    """No CPU implementation and no fallback: in a fresh interpreter (other tests may have registered
    the TEST-ONLY oracle-backed CPU ops) CPU tensors are rejected, not silently computed."""
    import subprocess
    import sys

    code = (
        "import torch, evogp_amd\n"
        "try:\n"
        "    torch.ops.evogp_cuda.tree_evaluate(1, 8, 1, 1, torch.zeros(1, 8), torch.zeros(1, 8, dtype=torch.int16),"
        " torch.zeros(1, 8, dtype=torch.int16), torch.zeros(1, 1))\n"
        "except (RuntimeError, NotImplementedError) as e:\n"
        "    print('REJECTED')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "REJECTED" in r.stdout, r.stdout + r.stderr
