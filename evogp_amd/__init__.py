"""evogp_amd — MI355X (gfx950) native tree-evaluation and genetic-operation engine behind the
EvoGP operator boundary (``torch.ops.evogp_cuda.*``) and the ``evogp.tree / algorithm / problem /
pipeline`` API surface.  See DESIGN.md and INTEGRATION.md."""
from . import ops  # noqa: F401  (loads evogp_amd/lib/libevogp_hip.so and registers the ops)
from . import tree, algorithm, problem, pipeline  # noqa: F401

__version__ = "0.1.0"


def set_sr_division(mode: str) -> None:
    """Division in the threaded-code ``tree_SR_fitness`` path (include/evogp_hip.h ``evogp_hip_set_sr_division``):
    ``"short"`` (default) — the IEEE sequence with its range scaling and special-case fix-up but one residual correction:
    faithfully rounded, the correctly rounded quotient for all but ~1 operand pair in 4e9; ``"ieee"`` — always the
    correctly rounded quotient (+20 % time); ``"fast"`` — no range scaling.  The reference fixes this at build time
    (``-use_fast_math``, setup.py:55: CUDA's approximate division)."""
    from . import _lib

    codes = {"ieee": 0, "fast": 1, "short": 2}
    assert mode in codes, f"mode should be one of {list(codes)}, but got {mode}"
    _lib.check(_lib.lib.evogp_hip_set_sr_division(codes[mode]), "set_sr_division")


def get_sr_division() -> str:
    from . import _lib

    return ("ieee", "fast", "short")[_lib.lib.evogp_hip_get_sr_division()]


def set_program_buffer_limit(nbytes: int) -> None:
    """Cap of the engine-owned program-record buffer of ``tree_SR_fitness`` (include/evogp_hip.h: size law, default 16 GiB).
    A call that would need more runs on the register interpreters (same results, slower); 0 disables the compiled path."""
    from . import _lib

    _lib.check(_lib.lib.evogp_hip_set_program_buffer_limit(int(nbytes)), "set_program_buffer_limit")


def program_buffer_bytes() -> int:
    """bytes of program records the engine holds on the current device (not visible to torch's allocator)"""
    from . import _lib

    return int(_lib.lib.evogp_hip_program_buffer_bytes())


def record_ring_bytes() -> int:
    """bytes of record rings the one-kernel fitness call holds on the current device (include/evogp_hip.h: a fixed size per stream)"""
    from . import _lib

    return int(_lib.lib.evogp_hip_record_ring_bytes())


def release_workspaces() -> None:
    """Wait for the current device and free the engine-owned program-record buffer; the next fitness call allocates again."""
    from . import _lib

    _lib.check(_lib.lib.evogp_hip_release_workspaces(), "release_workspaces")
