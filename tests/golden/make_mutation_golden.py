#!/usr/bin/env python3
"""Golden vectors for the structural / point mutation operators: the REFERENCE's own Python operators
(/root/reference/src/evogp/algorithm/mutation/*.py) executed in this container on the CPU, with every random number they
draw recorded.

The reference hard-codes device="cuda" and reaches its kernels through torch.ops.evogp_cuda.*; here
  * torch's factory functions and Tensor.to are patched so that "cuda" means the CPU,
  * the five ops run on the CPU oracle (tests/cpu_ops.py; the oracle is pinned bit-for-bit to the reference's device code by
    tests/test_oracle_vs_ref.py),
  * torch.rand / torch.randint / the reference's randint are wrapped and their results logged in call order.
Each case stores the input forest, the descriptor arguments, the log and the reference's result in
tests/golden/mutation_<case>.npz.  tests/mutation_replay.py turns a log into the arguments of this repository's
`apply` methods; tests/test_mutation_parity.py (CPU) and tests/test_gpu_mutation_parity.py (GPU) compare bit for bit.

    python tests/golden/make_mutation_golden.py          (needs /root/reference; never runs on the GPU box)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src"

# ---- "cuda" means the CPU ---------------------------------------------------------------------------------------------------
def _is_cuda(d):
    return d is not None and str(d).startswith("cuda")


def _decuda(fn):
    def wrapped(*a, **k):
        if _is_cuda(k.get("device")):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for _name in ("arange", "rand", "randint", "zeros", "ones", "empty", "tensor", "full", "randn", "zeros_like", "ones_like", "empty_like", "as_tensor"):
    setattr(torch, _name, _decuda(getattr(torch, _name)))
_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    a = tuple("cpu" if (isinstance(x, (str, torch.device)) and _is_cuda(x)) else x for x in a)
    if _is_cuda(k.get("device")):
        k["device"] = "cpu"
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to
torch.Tensor.cuda = lambda self, *a, **k: self

# ---- the op namespace on the CPU oracle, then the reference package under its own name ----------------------------------------
import cpu_ops  # noqa: E402

cpu_ops.register()
for _m in [m for m in sys.modules if m == "evogp" or m.startswith("evogp.")]:
    del sys.modules[_m]
sys.modules["evogp.evogp_cuda"] = types.ModuleType("evogp.evogp_cuda")
sys.path.insert(0, REF)
import evogp  # noqa: E402  (the reference)

assert evogp.__file__.startswith(REF), evogp.__file__
from evogp.algorithm import mutation as ref_mut  # noqa: E402
from evogp.tree import Forest, GenerateDescriptor  # noqa: E402
import evogp.tree.utils as ref_utils  # noqa: E402

# ---- recording ------------------------------------------------------------------------------------------------------------------
LOG = []
_depth = [0]
_rand0, _randint0, _ref_randint0 = torch.rand, torch.randint, ref_utils.randint


def _rec(tag, fn):
    def wrapped(*a, **k):
        _depth[0] += 1
        try:
            r = fn(*a, **k)
        finally:
            _depth[0] -= 1
        if _depth[0] == 0:
            LOG.append((tag, r.detach().clone()))
        return r
    return wrapped


torch.rand = _rec("rand", _rand0)
torch.randint = _rec("torch_randint", _randint0)
_ref_randint = _rec("ref_randint", _ref_randint0)
for _mod in list(sys.modules.values()):
    if _mod is not None and getattr(_mod, "__name__", "").startswith("evogp.") and getattr(_mod, "randint", None) is _ref_randint0:
        _mod.randint = _ref_randint


def forest_np(f):
    return f.batch_node_value.numpy().copy(), f.batch_node_type.numpy().copy(), f.batch_subtree_size.numpy().copy()


CASES = {
    # name: (operator factory taking the descriptor, descriptor kwargs, population, seed)
    "hoist": (lambda d: ref_mut.HoistMutation(0.6), dict(using_funcs=["+", "-", "*", "/", "sin", "neg", "if"], output_len=1), 240, 1),
    "hoist_mo": (lambda d: ref_mut.HoistMutation(0.9), dict(using_funcs=["+", "*", "max"], output_len=3), 200, 2),
    "delete": (lambda d: ref_mut.DeleteMutation(0.7), dict(using_funcs=["+", "-", "*", "/", "sin", "neg", "if"], output_len=1), 240, 3),
    "delete_capped": (lambda d: ref_mut.DeleteMutation(0.7, max_mutatable_size=7), dict(using_funcs=["+", "-", "if", "abs"], output_len=2), 240, 4),
    "insert": (lambda d: ref_mut.InsertMutation(0.6, d.update(max_layer_cnt=2)), dict(using_funcs=["+", "-", "*", "/", "sin", "neg", "if"], output_len=1), 240, 5),
    "insert_mo": (lambda d: ref_mut.InsertMutation(0.8, d.update(max_layer_cnt=3)), dict(using_funcs=["+", "*", "neg"], output_len=2), 200, 6),
    "single_point": (lambda d: ref_mut.SinglePointMutation(0.7, d), dict(using_funcs=["+", "-", "*", "/", "sin", "neg", "if"], output_len=1), 240, 7),
    "single_point_mo": (lambda d: ref_mut.SinglePointMutation(0.7, d, modify_output=True), dict(using_funcs=["+", "-", "sin", "if"], output_len=4), 240, 8),
    "single_point_binary_only": (lambda d: ref_mut.SinglePointMutation(0.8, d), dict(using_funcs=["+", "-", "*", "/"], output_len=1), 240, 9),
    "multi_point": (lambda d: ref_mut.MultiPointMutation(0.7, d, mutation_intensity=0.5), dict(using_funcs=["+", "-", "*", "/", "sin", "neg", "if"], output_len=1), 240, 10),
    "multi_point_mo": (lambda d: ref_mut.MultiPointMutation(0.7, d, mutation_intensity=0.6, modify_output=True), dict(using_funcs=["+", "*", "abs"], output_len=3), 200, 11),
    "single_const": (lambda d: ref_mut.SingleConstMutation(0.7, d), dict(using_funcs=["+", "-", "*", "/", "sin"], output_len=1), 240, 12),
    "multi_const": (lambda d: ref_mut.MultiConstMutation(0.7, d, mutation_intensity=0.5), dict(using_funcs=["+", "-", "*", "/", "sin"], output_len=2), 240, 13),
}
COMMON = dict(max_tree_len=64, input_len=4, max_layer_cnt=4, const_samples=[-1.0, 0.0, 1.0, 0.5, 2.0])


def main():
    for name, (make, dk, pop, seed) in CASES.items():
        torch.manual_seed(seed)
        kw = dict(COMMON, **dk)
        desc = GenerateDescriptor(**kw)
        forest = Forest.random_generate(pop, desc)
        before = forest_np(forest)
        op = make(desc)
        del LOG[:]
        out = op(Forest(forest.input_len, forest.output_len, forest.batch_node_value.clone(), forest.batch_node_type.clone(),
                        forest.batch_subtree_size.clone()))
        after = forest_np(out)
        log = list(LOG)
        arrays = {"in_value": before[0], "in_type": before[1], "in_size": before[2], "out_value": after[0], "out_type": after[1], "out_size": after[2]}
        tags = []
        for i, (tag, t) in enumerate(log):
            arrays[f"log{i}"] = t.numpy()
            tags.append(tag)
        params = {k: getattr(op, k) for k in ("mutation_rate", "max_mutatable_size", "mutation_intensity", "modify_output") if hasattr(op, k)}
        meta = dict(case=name, operator=type(op).__name__, descriptor=kw, params=params, tags=tags, pop=pop, seed=seed)
        if hasattr(op, "descriptor"):
            meta["op_descriptor"] = dict(kw, max_layer_cnt=2 if name == "insert" else 3 if name == "insert_mo" else kw["max_layer_cnt"])
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"mutation_{name}.npz"), **arrays)
        changed = int((np.asarray(before[0]).view(np.uint32) != np.asarray(after[0]).view(np.uint32)).any(1).sum())
        print(f"{name}: {type(op).__name__}, {len(log)} draws {tags}, {changed} of {pop} trees changed")


if __name__ == "__main__":
    main()
