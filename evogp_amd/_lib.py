"""Loader for the C-ABI engine ``evogp_amd/lib/libevogp_hip.so`` (include/evogp_hip.h).

There is NO fallback: if the shared object is missing or fails to load, importing the package
raises.  The library is built in-tree by ``__graft_entry__.build()`` / ``make -C evogp_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVOGP_HIP_LIB: alternative build of the same engine (A/B benchmarking of compiler flags only)
LIB_PATH = os.environ.get("EVOGP_HIP_LIB") or os.path.join(_HERE, "lib", "libevogp_hip.so")

ABI_VERSION = 5

_vp = C.c_void_p
_u = C.c_uint
_i = C.c_int
_f = C.c_float

# name -> argtypes, exactly the prototypes of include/evogp_hip.h
PROTOTYPES = {
    "evogp_hip_generate": [_u, _u, _u, _u, _u, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp],
    "evogp_hip_mutate": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_crossover": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_evaluate": [_u, _u, _u, _u, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_sr_fitness": [_u, _u, _u, _u, _u, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp],
    "evogp_hip_generate_masked": [_u, _u, _u, _u, _u, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp, _u, _vp],
    "evogp_hip_breed_default": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_breed_default_rows": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evogp_hip_breed_default_table": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evogp_hip_breed_lists": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evogp_hip_generate_masked_hashed": [_u, _u, _u, _u, _u, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _u, C.c_longlong, C.c_longlong, _u, _vp],
    "evogp_hip_breed_lists_hashed": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_longlong, C.c_longlong, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evogp_hip_sr_fitness_hinted": [_u, _u, _u, _u, _u, _i, _vp, _vp, _vp, _vp, _vp, _vp, _u, _u, _vp],
    "evogp_hip_batch_evaluate": [_u, _u, _u, _u, _u, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_batch_argmax_count": [_u, _u, _u, _u, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_evaluate_prepare": [_u, _u, _u, _u, _vp, _vp, _vp, _vp, C.c_size_t, _vp],
    "evogp_hip_evaluate_prepared": [_u, _u, _u, _u, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "evogp_hip_structural_mutate": [_i, _i, _i, _f, _i, _i, _i, C.c_longlong, C.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_insert_mutate": [_i, _i, C.c_uint, _i, C.c_longlong, C.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_point_mutate": [_i, _i, _i, _f, _f, _i, _i, _i, _i, _i, _i, _i, C.c_longlong, C.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_random_words": [C.c_longlong, C.c_longlong, _i, C.c_longlong, C.c_longlong, C.c_longlong, _vp, _vp],
    "evogp_hip_fitness_scores": [_u, _i, _vp, _vp, _vp],
    "evogp_hip_select": [_u, _u, _u, _vp, _vp, _vp, _vp],
    "evogp_hip_select_alternating": [_u, _u, _u, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_tournament_select": [_u, _u, _u, C.c_longlong, C.c_longlong, _vp, _vp, _vp],
    "evogp_hip_set_program_buffer_limit": [C.c_ulonglong],
    "evogp_hip_release_workspaces": [],
    "evogp_hip_set_allocator": [_vp, _vp],
    "evogp_hip_set_sr_division": [C.c_int],
    "evogp_hip_get_sr_division": [],
    "evogp_hip_abi_version": [],
}

# include/evogp_hip_debug.h: measurement and test hooks (bench.py, scripts/, tests/); nothing in this package calls them
DEBUG_PROTOTYPES = {
    "evogp_hip_timer_begin": [_vp],
    "evogp_hip_timer_end": [_vp, C.POINTER(C.c_float)],
    "evogp_hip_debug_set_stats": [_vp],
    "evogp_hip_debug_compile_batch": [_i],
    "evogp_hip_debug_long_compiler": [_i],
    "evogp_hip_debug_twins": [_i],
    "evogp_hip_debug_profile": [_i],
    "evogp_hip_debug_profile_read": [C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "evogp_hip_debug_tc_histogram": [_u, _vp, _i, _vp],
    "evogp_hip_debug_tc_nhandlers": [],
    "evogp_hip_debug_tc_program": [_u, _vp, _i],
    "evogp_hip_debug_forget_function_classes": [],
    "evogp_hip_debug_structural_mutate_given": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_debug_insert_mutate_given": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evogp_hip_debug_point_mutate_given": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()'  or  make -C evogp_amd/csrc). "
            "evogp_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.argtypes = argtypes
        fn.restype = _i
    lib.evogp_hip_evaluate_workspace_bytes.argtypes = [_u, _u]
    lib.evogp_hip_evaluate_workspace_bytes.restype = C.c_size_t
    lib.evogp_hip_select_workspace_bytes.argtypes = []
    lib.evogp_hip_select_workspace_bytes.restype = C.c_size_t
    lib.evogp_hip_program_buffer_bytes.argtypes = []
    lib.evogp_hip_program_buffer_bytes.restype = C.c_ulonglong
    lib.evogp_hip_record_ring_bytes.argtypes = []
    lib.evogp_hip_record_ring_bytes.restype = C.c_ulonglong
    lib.evogp_hip_error_string.argtypes = [_i]
    lib.evogp_hip_error_string.restype = C.c_char_p
    got = lib.evogp_hip_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libevogp_hip.so ABI version {got}, expected {ABI_VERSION}: rebuild the engine")
    return lib


lib = _load()


def check(code: int, what: str) -> None:
    """Map a non-zero return code to RuntimeError — the error convention of the reference's
    TORCH_CHECKs (src/evogp/cuda/torch_wrapper.cu:7-17,48-54)."""
    if code != 0:
        raise RuntimeError(f"{what} failed: {lib.evogp_hip_error_string(code).decode()} (code {code})")
