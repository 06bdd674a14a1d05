"""GPU-side test helper: call the C ABI (include/evogp_hip.h; the test entries of include/evogp_hip_debug.h) directly with device pointers.
numpy in, numpy out; torch is only used to own device memory."""
import numpy as np
import torch

from evogp_amd import _lib

L = _lib.lib
DEV = "cuda:0"


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(DEV)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _out3(pop, gp_len, poison=True):
    # poison the outputs so that "every byte is written" is actually tested
    v = torch.full((pop, gp_len), float("nan"), dtype=torch.float32, device=DEV) if poison else torch.empty((pop, gp_len), dtype=torch.float32, device=DEV)
    t = torch.full((pop, gp_len), -7, dtype=torch.int16, device=DEV)
    s = torch.full((pop, gp_len), -7, dtype=torch.int16, device=DEV)
    return v, t, s


def _np3(v, t, s):
    torch.cuda.synchronize()
    return v.cpu().numpy(), t.cpu().numpy(), s.cpu().numpy()


def generate(pop, gp_len, var_len, out_len, out_prob, const_prob, keys, d2l, rou, cs, offset=0, expect=0):
    k = dev(keys, np.uint32); d = dev(d2l, np.float32); r = dev(rou, np.float32); c = dev(cs, np.float32)
    v, t, s = _out3(pop, gp_len)
    rc = L.evogp_hip_generate(pop, gp_len, var_len, out_len, c.shape[0], out_prob, const_prob, k.data_ptr(), d.data_ptr(),
                              r.data_ptr(), c.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), offset, _stream())
    assert rc == expect, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def mutate(value, type_, size, idx, nvalue, ntype, nsize):
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(idx, np.int32),
         dev(nvalue, np.float32), dev(ntype, np.int16), dev(nsize, np.int16)]
    v, t, s = _out3(pop, gp_len)
    rc = L.evogp_hip_mutate(pop, gp_len, *[x.data_ptr() for x in a], v.data_ptr(), t.data_ptr(), s.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def crossover(value, type_, size, li, ri, ln, rn):
    pop, gp_len = value.shape
    n = len(li)
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(li, np.int32), dev(ri, np.int32),
         dev(ln, np.int32), dev(rn, np.int32)]
    v, t, s = _out3(n, gp_len)
    rc = L.evogp_hip_crossover(pop, n, gp_len, *[x.data_ptr() for x in a], v.data_ptr(), t.data_ptr(), s.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def evaluate(value, type_, size, X, out_len):
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(X, np.float32)]
    res = torch.full((pop, out_len), 12345.0, dtype=torch.float32, device=DEV)
    rc = L.evogp_hip_evaluate(pop, gp_len, a[3].shape[1], out_len, *[x.data_ptr() for x in a], res.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return res.cpu().numpy()


def sr_fitness(value, type_, size, X, y, use_mse=True, kernel_type=0, func_mask=0, forget=True):
    """func_mask != 0: evogp_hip_sr_fitness_hinted with that function-set mask (what evogp_amd.tree.Forest.SR_fitness and bench.py call).
    Without a mask the engine chooses its program compiler by what the last call on a forest of this shape observed; `forget` drops
    those observations first (include/evogp_hip_debug.h), so that the call looks at ITS forest and a test's kernels do not depend on the
    tests before it.  tests/test_gpu_learned.py is about the observations themselves."""
    pop, gp_len = value.shape
    if forget and not func_mask:
        assert L.evogp_hip_debug_forget_function_classes() == 0
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(X, np.float32), dev(y, np.float32)]
    D, var_len = a[3].shape
    fit = torch.full((pop,), 12345.0, dtype=torch.float32, device=DEV)
    if func_mask:
        rc = L.evogp_hip_sr_fitness_hinted(pop, D, gp_len, var_len, a[4].shape[1], int(use_mse), *[x.data_ptr() for x in a],
                                           fit.data_ptr(), kernel_type, func_mask, _stream())
    else:
        rc = L.evogp_hip_sr_fitness(pop, D, gp_len, var_len, a[4].shape[1], int(use_mse), *[x.data_ptr() for x in a],
                                    fit.data_ptr(), kernel_type, _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return fit.cpu().numpy()


def batch_evaluate(value, type_, size, X, out_len):
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(X, np.float32)]
    D, var_len = a[3].shape
    res = torch.full((pop, D, out_len), 12345.0, dtype=torch.float32, device=DEV)
    rc = L.evogp_hip_batch_evaluate(pop, D, gp_len, var_len, out_len, *[x.data_ptr() for x in a], res.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return res.cpu().numpy()


def generate_masked(pop, gp_len, var_len, out_len, out_prob, const_prob, keys, d2l, rou, cs, active_word, active_below, offset=0):
    """-> (value, type, size) with the rows of inactive trees still holding the poison pattern."""
    k = dev(keys, np.uint32); d = dev(d2l, np.float32); r = dev(rou, np.float32); c = dev(cs, np.float32)
    aw = dev(active_word, np.int32)
    v, t, s = _out3(pop, gp_len)
    rc = L.evogp_hip_generate_masked(pop, gp_len, var_len, out_len, c.shape[0], out_prob, const_prob, k.data_ptr(), d.data_ptr(),
                                     r.data_ptr(), c.data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), offset,
                                     aw.data_ptr(), int(active_below), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def breed_default(value, type_, size, order, rnd, mutate_below, n_elite, n_surv, dvalue, dtype_, dsize):
    """-> ((value, type, size) of the next generation, decisions int32[n_new][6])"""
    pop, gp_len = value.shape
    n_new = pop - n_elite
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(order, np.int32), dev(rnd, np.int32)]
    d = [dev(dvalue, np.float32), dev(dtype_, np.int16), dev(dsize, np.int16)]
    v, t, s = _out3(pop, gp_len)
    dec = torch.full((n_new, 6), -9, dtype=torch.int32, device=DEV)
    rc = L.evogp_hip_breed_default(pop, gp_len, n_elite, n_surv, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(),
                                   a[3].data_ptr(), a[4].data_ptr(), int(mutate_below), d[0].data_ptr(), d[1].data_ptr(),
                                   d[2].data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), dec.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    out = _np3(v, t, s)
    return out, dec.cpu().numpy()


def breed_lists(value, type_, size, elite_rows, parent_rows, rnd, mutate_below, dvalue, dtype_, dsize, pop=None, row_begin=0, row_count=None):
    """evogp_hip_breed_lists: the breeding pass under any selection (separate elite / parent lists, parents may repeat)
    -> ((value, type, size) of rows [row_begin, row_begin + row_count), decisions int32[row_count][6])"""
    table_rows, gp_len = value.shape
    pop = table_rows if pop is None else pop
    row_count = pop if row_count is None else row_count
    n_elite, n_surv = len(elite_rows), len(parent_rows)
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(elite_rows if n_elite else [0], np.int32),
         dev(parent_rows, np.int32), dev(rnd, np.int32)]
    d = [dev(dvalue, np.float32), dev(dtype_, np.int16), dev(dsize, np.int16)]     # row_count rows: row k belongs to row row_begin + k
    v, t, s = _out3(row_count, gp_len)
    dec = torch.full((row_count, 6), -9, dtype=torch.int32, device=DEV)
    rc = L.evogp_hip_breed_lists(pop, table_rows, gp_len, n_elite, n_surv, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(),
                                 a[3].data_ptr() if n_elite else None, a[4].data_ptr(), a[5].data_ptr(), int(mutate_below), d[0].data_ptr(),
                                 d[1].data_ptr(), d[2].data_ptr(), v.data_ptr(), t.data_ptr(), s.data_ptr(), dec.data_ptr(), row_begin,
                                 row_count, _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    out = _np3(v, t, s)
    return out, dec.cpu().numpy()


def batch_argmax_count(value, type_, size, X, labels, out_len):
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(X, np.float32), dev(labels, np.int32)]
    D, var_len = a[3].shape
    cnt = torch.full((pop,), -5, dtype=torch.int32, device=DEV)
    rc = L.evogp_hip_batch_argmax_count(pop, D, gp_len, var_len, out_len, *[x.data_ptr() for x in a], cnt.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return cnt.cpu().numpy()


def evaluate_prepared(value, type_, size, X, out_len, steps=1):
    """-> (results of the prepared forward pass, number of trees it left to the stack interpreter)"""
    import ctypes

    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(X, np.float32)]
    var_len = a[3].shape[1]
    nbytes = L.evogp_hip_evaluate_workspace_bytes(pop, gp_len)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    rc = L.evogp_hip_evaluate_prepare(pop, gp_len, var_len, out_len, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), ws.data_ptr(),
                                      ctypes.c_size_t(nbytes), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    left = int(ws[nbytes - 64:nbytes - 60].view(torch.int32).item())
    res = torch.full((pop, out_len), 12345.0, dtype=torch.float32, device=DEV)
    for _ in range(steps):
        rc = L.evogp_hip_evaluate_prepared(pop, gp_len, var_len, out_len, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), ws.data_ptr(),
                                           1 if left else 0, a[3].data_ptr(), res.data_ptr(), _stream())
        assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return res.cpu().numpy(), left


def structural_mutate_given(value, type_, size, mode, given, inner_is_offset=False, skip_rows=0):
    """Delete (mode 0) / Hoist (mode 1) with the reference's own draws: given int32 [pop][3] = {mutates, node, child number / inner position}"""
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16)]
    gd = dev(given, np.int32)
    v, t, s = _out3(pop, gp_len)
    rc = L.evogp_hip_debug_structural_mutate_given(pop, gp_len, mode, int(inner_is_offset), skip_rows, gd.data_ptr(), *[x.data_ptr() for x in a],
                                                   v.data_ptr(), t.data_ptr(), s.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def insert_mutate_given(value, type_, size, given, fresh, skip_rows=0):
    """Insert with the reference's own draws: given int32 [pop][3] = {mutates, node, position inside the fresh tree}; fresh = (value, type, size), row n for tree n"""
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16), dev(fresh[0], np.float32), dev(fresh[1], np.int16), dev(fresh[2], np.int16)]
    gd = dev(given, np.int32)
    v, t, s = _out3(pop, gp_len)
    rc = L.evogp_hip_debug_insert_mutate_given(pop, gp_len, skip_rows, gd.data_ptr(), *[x.data_ptr() for x in a], v.data_ptr(), t.data_ptr(), s.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    return _np3(v, t, s)


def point_mutate_given(value, type_, size, mode, target, u, var_idx, const_idx, out_idx, roulettes, consts, input_len, output_len, modify_output=False,
                       fix_roulette=False, skip_rows=0):
    """the point mutations with the reference's own per-node draws (all [pop][gp_len]); -> new values"""
    pop, gp_len = value.shape
    a = [dev(value, np.float32), dev(type_, np.int16), dev(size, np.int16)]
    tg = dev(target, np.uint8)
    ud = dev(u, np.float32) if u is not None else None
    vi = dev(var_idx, np.int32) if var_idx is not None else None
    ci = dev(const_idx, np.int32)
    oi = dev(out_idx, np.int32) if out_idx is not None else None
    rs = [dev(r, np.float32) for r in roulettes] if roulettes is not None else [None] * 3
    cs = dev(consts, np.float32)
    out = torch.full((pop, gp_len), float("nan"), dtype=torch.float32, device=DEV)
    ptr = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
    rc = L.evogp_hip_debug_point_mutate_given(pop, gp_len, mode, int(modify_output), int(fix_roulette), skip_rows, input_len, output_len, cs.shape[0],
                                              tg.data_ptr(), ptr(ud), ptr(vi), ci.data_ptr(), ptr(oi), *[x.data_ptr() for x in a], *[ptr(r) for r in rs],
                                              cs.data_ptr(), out.data_ptr(), _stream())
    assert rc == 0, L.evogp_hip_error_string(rc)
    torch.cuda.synchronize()
    return out.cpu().numpy()
