"""``torch.ops.evogp_cuda.*`` — the reference's operator boundary, backed by the HIP engine.

The five schemas are the reference's, verbatim (src/evogp/cuda/torch_wrapper.cu:294-298), so code
written against ``torch.ops.evogp_cuda.tree_generate / tree_mutate / tree_crossover /
tree_evaluate / tree_SR_fitness`` runs unchanged.  Each implementation does what the reference's
C++ wrapper does — validate sizes, require contiguous device tensors of the exact shape
(``check_tensor``, torch_wrapper.cu:7-17), allocate the outputs on the device of the designated
input (:63,116,168,217,264) — and then calls the C ABI of include/evogp_hip.h with raw device
pointers on torch's CURRENT stream (the reference uses the legacy default stream).  Errors surface
as ``RuntimeError`` exactly like ``TORCH_CHECK``.

ROCm builds of PyTorch dispatch HIP tensors on the ``CUDA`` key, so the implementations are
registered for ``"CUDA"``; there is deliberately no CPU implementation (the reference has none,
torch_wrapper.cu:301) and no fallback.

Extra ops live in the ``evogp_hip`` namespace (they have no counterpart in the reference):
``evogp_hip::tree_generate_offset`` (tree-index offset for sharded populations),
``evogp_hip::tree_batch_evaluate`` (non-replicating ``Forest.batch_forward``),
``evogp_hip::tree_generate_masked`` and ``evogp_hip::breed_default`` (the default generation step in two launches,
SURVEY.md §8f N2).
"""
from __future__ import annotations

import torch

from . import _lib

MAX_STACK = 1024
MAX_FULL_DEPTH = 10
FUNC_END = 29

_lib_h = _lib.lib

# ---- schemas (verbatim) -----------------------------------------------------------------------
torch.library.define(
    "evogp_cuda::tree_generate",
    "(int i1, int i2, int i3, int i4, int i5, float f1, float f2, Tensor t1, Tensor t2, Tensor t3, Tensor t4)"
    " -> (Tensor t5, Tensor t6, Tensor t7)",
)
torch.library.define(
    "evogp_cuda::tree_mutate",
    "(int i1, int i2, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, Tensor t6, Tensor t7)"
    " -> (Tensor t8, Tensor t9, Tensor t10)",
)
torch.library.define(
    "evogp_cuda::tree_crossover",
    "(int i1, int i2, int i3, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, Tensor t6, Tensor t7)"
    " -> (Tensor t8, Tensor t9, Tensor t10)",
)
torch.library.define(
    "evogp_cuda::tree_evaluate",
    "(int i1, int i2, int i3, int i4, Tensor t1, Tensor t2, Tensor t3, Tensor t4) -> Tensor t5",
)
torch.library.define(
    "evogp_cuda::tree_SR_fitness",
    "(int i1, int i2, int i3, int i4, int i5, bool b1, Tensor t1, Tensor t2, Tensor t3, Tensor t4, Tensor t5, int i6)"
    " -> Tensor t6",
)
torch.library.define(
    "evogp_hip::tree_generate_offset",
    "(int pop_size, int gp_len, int var_len, int out_len, int const_samples_len, float out_prob, float const_prob,"
    " Tensor keys, Tensor depth2leaf_probs, Tensor roulette_funcs, Tensor const_samples, int tree_index_offset)"
    " -> (Tensor value, Tensor node_type, Tensor subtree_size)",
)
torch.library.define(
    "evogp_hip::tree_batch_evaluate",
    "(int pop_size, int data_points, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type,"
    " Tensor subtree_size, Tensor variables) -> Tensor results",
)


torch.library.define(
    "evogp_hip::tree_generate_masked",
    "(int pop_size, int gp_len, int var_len, int out_len, int const_samples_len, float out_prob, float const_prob,"
    " Tensor keys, Tensor depth2leaf_probs, Tensor roulette_funcs, Tensor const_samples, int tree_index_offset,"
    " Tensor active_word, int active_below) -> (Tensor value, Tensor node_type, Tensor subtree_size)",
)
torch.library.define(
    "evogp_hip::breed_default",
    "(int pop_size, int gp_len, int n_elite, int n_surv, Tensor value, Tensor node_type, Tensor subtree_size,"
    " Tensor order, Tensor rnd, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size,"
    " bool want_decisions) -> (Tensor value, Tensor node_type, Tensor subtree_size, Tensor decisions)",
)


torch.library.define(
    "evogp_hip::breed_default_rows",
    "(int pop_size, int gp_len, int n_elite, int n_surv, Tensor value, Tensor node_type, Tensor subtree_size,"
    " Tensor order, Tensor rnd, int mutate_below, Tensor donor_value, Tensor donor_type, Tensor donor_size,"
    " int row_begin, int row_count) -> (Tensor value, Tensor node_type, Tensor subtree_size)",
)


torch.library.define(
    "evogp_hip::tree_batch_argmax_count",
    "(int pop_size, int data_points, int gp_len, int var_len, int out_len, Tensor value, Tensor node_type,"
    " Tensor subtree_size, Tensor variables, Tensor labels) -> Tensor counts",
)


# ---- helpers ----------------------------------------------------------------------------------
def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _check_tensor(t: torch.Tensor, shape, name: str, dtype=None) -> None:
    _check(t.is_cuda and t.is_contiguous(), f"{name} must be a contiguous CUDA tensor")
    _check(tuple(t.shape) == tuple(shape), f"{name} must have shape {list(shape)}, but got shape {list(t.shape)}")
    if dtype is not None:
        _check(t.dtype == dtype, f"expected scalar type {dtype} for {name} but found {t.dtype}")


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _check_sizes_common(pop_size: int, gp_len: int) -> None:
    _check(pop_size > 0, f"pop_size must larger than 0, but got {pop_size}")
    _check(0 < gp_len <= MAX_STACK, f"gp_len must be in range (0, {MAX_STACK}], but got {gp_len}")


def _keys_u32(keys: torch.Tensor) -> torch.Tensor:
    # the reference reads keys.data_ptr<unsigned int>() (torch_wrapper.cu:76); accept the signed
    # integer dtypes defensively (same 32-bit patterns)
    if keys.dtype == torch.uint32:
        return keys
    _check(keys.dtype in (torch.int32, torch.int64), f"keys must be uint32/int32/int64, got {keys.dtype}")
    return (keys.to(torch.int64) & 0xFFFFFFFF).to(torch.uint32).contiguous()


def _generate(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys, depth2leaf_probs,
              roulette_funcs, const_samples, tree_index_offset, active_word=None, active_below=0):
    _check_sizes_common(pop_size, gp_len)
    _check(var_len > 0, f"var_len must larger than 0, but got {var_len}")
    _check(out_len > 0, f"out_len must larger than 0, but got {out_len}")
    _check(const_samples_len > 0, f"const_samples_len must larger than 0, but got {const_samples_len}")
    _check(0 <= out_prob <= 1, f"out_prob must be in range [0, 1], but got {out_prob}")
    _check(0 <= const_prob <= 1, f"const_prob must be in range [0, 1], but got {const_prob}")
    _check_tensor(keys, (2,), "keys")
    _check_tensor(depth2leaf_probs, (MAX_FULL_DEPTH,), "depth2leaf_probs", torch.float32)
    _check_tensor(roulette_funcs, (FUNC_END,), "roulette_funcs", torch.float32)
    _check_tensor(const_samples, (const_samples_len,), "const_samples", torch.float32)
    _check(0 <= tree_index_offset < 2**32, "tree_index_offset must fit in 32 bits")
    keys = _keys_u32(keys)
    dev = keys.device
    with torch.cuda.device(dev):
        value = torch.empty((pop_size, gp_len), dtype=torch.float32, device=dev)
        ntype = torch.empty((pop_size, gp_len), dtype=torch.int16, device=dev)
        size = torch.empty((pop_size, gp_len), dtype=torch.int16, device=dev)
        if active_word is not None:
            _check_tensor(active_word, (pop_size,), "active_word", torch.int32)
        rc = _lib_h.evogp_hip_generate_masked(
            pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob,
            keys.data_ptr(), depth2leaf_probs.data_ptr(), roulette_funcs.data_ptr(), const_samples.data_ptr(),
            value.data_ptr(), ntype.data_ptr(), size.data_ptr(), tree_index_offset,
            active_word.data_ptr() if active_word is not None else None, active_below, _stream(dev))
    _lib.check(rc, "tree_generate")
    return value, ntype, size


# ---- implementations --------------------------------------------------------------------------
@torch.library.impl("evogp_cuda::tree_generate", "CUDA")
def tree_generate(i1, i2, i3, i4, i5, f1, f2, t1, t2, t3, t4):
    return _generate(i1, i2, i3, i4, i5, f1, f2, t1, t2, t3, t4, 0)


@torch.library.impl("evogp_hip::tree_generate_offset", "CUDA")
def tree_generate_offset(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys,
                         depth2leaf_probs, roulette_funcs, const_samples, tree_index_offset):
    return _generate(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys,
                     depth2leaf_probs, roulette_funcs, const_samples, tree_index_offset)


@torch.library.impl("evogp_hip::tree_generate_masked", "CUDA")
def tree_generate_masked(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys,
                         depth2leaf_probs, roulette_funcs, const_samples, tree_index_offset, active_word, active_below):
    """Rows of trees with active_word[n] >= active_below are left uninitialised."""
    _check(0 <= active_below < 2**32, "active_below must fit in 32 bits")
    return _generate(pop_size, gp_len, var_len, out_len, const_samples_len, out_prob, const_prob, keys,
                     depth2leaf_probs, roulette_funcs, const_samples, tree_index_offset, active_word, active_below)


@torch.library.impl("evogp_hip::breed_default", "CUDA")
def breed_default(pop_size, gp_len, n_elite, n_surv, value, ntype, size, order, rnd, mutate_below, donor_value,
                  donor_type, donor_size, want_decisions):
    _check_sizes_common(pop_size, gp_len)
    _check(0 <= n_elite <= pop_size, f"n_elite must be in [0, pop_size], but got {n_elite}")
    _check(0 < n_surv <= pop_size, f"n_surv must be in (0, pop_size], but got {n_surv}")
    _check_forest(pop_size, gp_len, value, ntype, size)
    n_new = pop_size - n_elite
    _check(order.is_cuda and order.is_contiguous() and order.dtype == torch.int32 and order.dim() == 1
           and order.shape[0] >= max(n_elite, n_surv), "order must be a contiguous int32 CUDA vector of >= max(n_elite, n_surv) entries")
    _check_tensor(rnd, (6, n_new), "rnd", torch.int32)
    _check(0 <= mutate_below < 2**32, "mutate_below must fit in 32 bits")
    for t, nm, dt in ((donor_value, "donor_value", torch.float32), (donor_type, "donor_type", torch.int16),
                      (donor_size, "donor_size", torch.int16)):
        _check_tensor(t, (n_new, gp_len), nm, dt)
    dev = value.device
    shp = (pop_size, gp_len)
    with torch.cuda.device(dev):
        ov = torch.empty(shp, dtype=torch.float32, device=dev)
        ot = torch.empty(shp, dtype=torch.int16, device=dev)
        osz = torch.empty(shp, dtype=torch.int16, device=dev)
        dec = torch.empty((n_new, 6) if want_decisions else (0, 6), dtype=torch.int32, device=dev)
        rc = _lib_h.evogp_hip_breed_default(
            pop_size, gp_len, n_elite, n_surv, value.data_ptr(), ntype.data_ptr(), size.data_ptr(), order.data_ptr(),
            rnd.data_ptr(), mutate_below, donor_value.data_ptr(), donor_type.data_ptr(), donor_size.data_ptr(),
            ov.data_ptr(), ot.data_ptr(), osz.data_ptr(), dec.data_ptr() if want_decisions else None, _stream(dev))
    _lib.check(rc, "breed_default")
    return ov, ot, osz, dec


@torch.library.impl("evogp_hip::tree_batch_argmax_count", "CUDA")
def tree_batch_argmax_count(pop_size, data_points, gp_len, var_len, out_len, value, ntype, size, variables, labels):
    """counts[t] = #rows whose arg-max output (as torch.argmax(clip(softmax(.))) sees it) equals the int32 label."""
    _check_sizes_common(pop_size, gp_len)
    _check(data_points > 0 and var_len > 0, "data_points and var_len must be larger than 0")
    _check(2 <= out_len <= 16, f"out_len must be in [2, 16], but got {out_len}")
    _check_forest(pop_size, gp_len, value, ntype, size)
    _check_tensor(variables, (data_points, var_len), "variables", torch.float32)
    _check_tensor(labels, (data_points,), "labels", torch.int32)
    dev = value.device
    with torch.cuda.device(dev):
        counts = torch.empty((pop_size,), dtype=torch.int32, device=dev)
        rc = _lib_h.evogp_hip_batch_argmax_count(pop_size, data_points, gp_len, var_len, out_len, value.data_ptr(),
                                                 ntype.data_ptr(), size.data_ptr(), variables.data_ptr(), labels.data_ptr(),
                                                 counts.data_ptr(), _stream(dev))
    _lib.check(rc, "tree_batch_argmax_count")
    return counts


@torch.library.impl("evogp_hip::breed_default_rows", "CUDA")
def breed_default_rows(pop_size, gp_len, n_elite, n_surv, value, ntype, size, order, rnd, mutate_below, donor_value,
                       donor_type, donor_size, row_begin, row_count):
    """Rows [row_begin, row_begin + row_count) of the next generation.  The donor arrays are aligned with the OFFSPRING rows
    of the range: they may cover the whole range (row_count rows; the rows of elites are never read) or only its
    offspring (row_count minus the elite rows at the head of the range) — then no padding copy is needed."""
    _check_sizes_common(pop_size, gp_len)
    _check(0 <= n_elite <= pop_size and 0 < n_surv <= pop_size, "n_elite / n_surv out of range")
    _check(0 <= row_begin and 0 < row_count and row_begin + row_count <= pop_size, "row range out of the population")
    # value / type / size: the whole population, or only the trees `order` names (a sharded run's survivor table)
    table_rows = value.shape[0] if value.dim() == 2 else 0
    _check(table_rows > 0, "value must be a (rows, gp_len) tensor")
    _check_forest(table_rows, gp_len, value, ntype, size)
    _check(order.is_cuda and order.is_contiguous() and order.dtype == torch.int32 and order.dim() == 1
           and order.shape[0] >= max(n_elite, n_surv), "order must be a contiguous int32 CUDA vector of >= max(n_elite, n_surv) entries")
    _check_tensor(rnd, (6, pop_size - n_elite), "rnd", torch.int32)
    head = max(0, min(row_begin + row_count, n_elite) - row_begin)  # elite rows at the head of the range
    drows = donor_value.shape[0] if donor_value.dim() == 2 else -1
    _check(drows in (row_count, row_count - head), f"donor arrays must have {row_count} or {row_count - head} rows, but got {drows}")
    for t, nm, dt in ((donor_value, "donor_value", torch.float32), (donor_type, "donor_type", torch.int16),
                      (donor_size, "donor_size", torch.int16)):
        _check_tensor(t, (drows, gp_len), nm, dt)
    skip = head if drows == row_count - head else 0  # the engine indexes donors by (row - row_begin)
    dev = value.device
    shp = (row_count, gp_len)
    with torch.cuda.device(dev):
        ov = torch.empty(shp, dtype=torch.float32, device=dev)
        ot = torch.empty(shp, dtype=torch.int16, device=dev)
        osz = torch.empty(shp, dtype=torch.int16, device=dev)
        rc = _lib_h.evogp_hip_breed_default_table(
            pop_size, table_rows, gp_len, n_elite, n_surv, value.data_ptr(), ntype.data_ptr(), size.data_ptr(), order.data_ptr(),
            rnd.data_ptr(), mutate_below, donor_value.data_ptr() - skip * gp_len * 4, donor_type.data_ptr() - skip * gp_len * 2,
            donor_size.data_ptr() - skip * gp_len * 2, ov.data_ptr(), ot.data_ptr(), osz.data_ptr(), None, row_begin, row_count,
            _stream(dev))
    _lib.check(rc, "breed_default_rows")
    return ov, ot, osz


@torch.library.impl("evogp_cuda::tree_mutate", "CUDA")
def tree_mutate(pop_size, gp_len, value_ori, type_ori, size_ori, mutate_indices, value_new, type_new, size_new):
    _check_sizes_common(pop_size, gp_len)
    shp = (pop_size, gp_len)
    _check_tensor(value_ori, shp, "value_ori", torch.float32)
    _check_tensor(type_ori, shp, "type_ori", torch.int16)
    _check_tensor(size_ori, shp, "subtree_size_ori", torch.int16)
    _check_tensor(mutate_indices, (pop_size,), "mutateIndices", torch.int32)
    _check_tensor(value_new, shp, "value_new", torch.float32)
    _check_tensor(type_new, shp, "type_new", torch.int16)
    _check_tensor(size_new, shp, "subtree_size_new", torch.int16)
    dev = value_new.device
    with torch.cuda.device(dev):
        value = torch.empty(shp, dtype=torch.float32, device=dev)
        ntype = torch.empty(shp, dtype=torch.int16, device=dev)
        size = torch.empty(shp, dtype=torch.int16, device=dev)
        rc = _lib_h.evogp_hip_mutate(
            pop_size, gp_len, value_ori.data_ptr(), type_ori.data_ptr(), size_ori.data_ptr(),
            mutate_indices.data_ptr(), value_new.data_ptr(), type_new.data_ptr(), size_new.data_ptr(),
            value.data_ptr(), ntype.data_ptr(), size.data_ptr(), _stream(dev))
    _lib.check(rc, "tree_mutate")
    return value, ntype, size


@torch.library.impl("evogp_cuda::tree_crossover", "CUDA")
def tree_crossover(pop_size_ori, pop_size_new, gp_len, value_ori, type_ori, size_ori, left_idx, right_idx,
                   left_node_idx, right_node_idx):
    _check(pop_size_ori > 0, f"pop_size_ori must larger than 0, but got {pop_size_ori}")
    _check(pop_size_new > 0, f"pop_size_new must larger than 0, but got {pop_size_new}")
    _check(0 < gp_len <= MAX_STACK, f"gp_len must be in range (0, {MAX_STACK}], but got {gp_len}")
    shp = (pop_size_ori, gp_len)
    _check_tensor(value_ori, shp, "value_ori", torch.float32)
    _check_tensor(type_ori, shp, "type_ori", torch.int16)
    _check_tensor(size_ori, shp, "subtree_size_ori", torch.int16)
    for t, nm in ((left_idx, "left_idx"), (right_idx, "right_idx"), (left_node_idx, "left_node_idx"),
                  (right_node_idx, "right_node_idx")):
        _check_tensor(t, (pop_size_new,), nm, torch.int32)
    dev = value_ori.device
    out = (pop_size_new, gp_len)
    with torch.cuda.device(dev):
        value = torch.empty(out, dtype=torch.float32, device=dev)
        ntype = torch.empty(out, dtype=torch.int16, device=dev)
        size = torch.empty(out, dtype=torch.int16, device=dev)
        rc = _lib_h.evogp_hip_crossover(
            pop_size_ori, pop_size_new, gp_len, value_ori.data_ptr(), type_ori.data_ptr(), size_ori.data_ptr(),
            left_idx.data_ptr(), right_idx.data_ptr(), left_node_idx.data_ptr(), right_node_idx.data_ptr(),
            value.data_ptr(), ntype.data_ptr(), size.data_ptr(), _stream(dev))
    _lib.check(rc, "tree_crossover")
    return value, ntype, size


def _check_forest(pop_size, gp_len, value, ntype, size):
    shp = (pop_size, gp_len)
    _check_tensor(value, shp, "value", torch.float32)
    _check_tensor(ntype, shp, "type", torch.int16)
    _check_tensor(size, shp, "subtree_size", torch.int16)


@torch.library.impl("evogp_cuda::tree_evaluate", "CUDA")
def tree_evaluate(pop_size, gp_len, var_len, out_len, value, ntype, size, variables):
    _check_sizes_common(pop_size, gp_len)
    _check(var_len > 0, f"var_len must larger than 0, but got {var_len}")
    _check(out_len > 0, f"out_len must larger than 0, but got {out_len}")
    _check_forest(pop_size, gp_len, value, ntype, size)
    _check_tensor(variables, (pop_size, var_len), "variables", torch.float32)
    dev = value.device
    with torch.cuda.device(dev):
        results = torch.empty((pop_size, out_len), dtype=torch.float32, device=dev)
        rc = _lib_h.evogp_hip_evaluate(pop_size, gp_len, var_len, out_len, value.data_ptr(), ntype.data_ptr(),
                                       size.data_ptr(), variables.data_ptr(), results.data_ptr(), _stream(dev))
    _lib.check(rc, "tree_evaluate")
    return results


@torch.library.impl("evogp_cuda::tree_SR_fitness", "CUDA")
def tree_sr_fitness(pop_size, data_points, gp_len, var_len, out_len, use_mse, value, ntype, size, variables, labels,
                    kernel_type):
    _check_sizes_common(pop_size, gp_len)
    _check(var_len > 0, f"var_len must larger than 0, but got {var_len}")
    _check(out_len > 0, f"out_len must larger than 0, but got {out_len}")
    _check(data_points > 0, f"data_points must larger than 0, but got {data_points}")
    _check(0 <= kernel_type <= 4, f"kernel_type must be in 0..4, but got {kernel_type}")
    _check_forest(pop_size, gp_len, value, ntype, size)
    _check_tensor(variables, (data_points, var_len), "variables", torch.float32)
    _check_tensor(labels, (data_points, out_len), "labels", torch.float32)
    dev = value.device
    with torch.cuda.device(dev):
        fitness = torch.empty((pop_size,), dtype=torch.float32, device=dev)
        rc = _lib_h.evogp_hip_sr_fitness(pop_size, data_points, gp_len, var_len, out_len, int(bool(use_mse)),
                                         value.data_ptr(), ntype.data_ptr(), size.data_ptr(), variables.data_ptr(),
                                         labels.data_ptr(), fitness.data_ptr(), kernel_type, _stream(dev))
    _lib.check(rc, "tree_SR_fitness")
    return fitness


@torch.library.impl("evogp_hip::tree_batch_evaluate", "CUDA")
def tree_batch_evaluate(pop_size, data_points, gp_len, var_len, out_len, value, ntype, size, variables):
    _check_sizes_common(pop_size, gp_len)
    _check(var_len > 0 and out_len > 0 and data_points > 0, "var_len, out_len and data_points must be positive")
    _check_forest(pop_size, gp_len, value, ntype, size)
    _check_tensor(variables, (data_points, var_len), "variables", torch.float32)
    dev = value.device
    with torch.cuda.device(dev):
        results = torch.empty((pop_size, data_points, out_len), dtype=torch.float32, device=dev)
        rc = _lib_h.evogp_hip_batch_evaluate(pop_size, data_points, gp_len, var_len, out_len, value.data_ptr(),
                                             ntype.data_ptr(), size.data_ptr(), variables.data_ptr(),
                                             results.data_ptr(), _stream(dev))
    _lib.check(rc, "tree_batch_evaluate")
    return results
