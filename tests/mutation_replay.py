"""Replay a recorded run of a REFERENCE mutation operator (tests/golden/mutation_*.npz, written by
tests/golden/make_mutation_golden.py) through this repository's operators: the reference's draws, which it makes for the
mutating trees only, are scattered into the population-sized arguments of `apply`."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    return sorted(f[len("mutation_"):-4] for f in os.listdir(GOLD) if f.startswith("mutation_") and f.endswith(".npz"))


def load(case):
    z = np.load(os.path.join(GOLD, f"mutation_{case}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    log = [z[f"log{i}"] for i in range(len(meta["tags"]))]
    return z, meta, log


def replay(case, device):
    """-> (result Forest of this repository's operator, the reference's (value, type, size))"""
    from evogp_amd.algorithm import (DeleteMutation, HoistMutation, InsertMutation, MultiConstMutation, MultiPointMutation,
                                     SingleConstMutation, SinglePointMutation)
    from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

    set_default_device(device)
    z, meta, log = load(case)
    dev = torch.device(device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    dk = meta["descriptor"]
    desc = GenerateDescriptor(**dk)
    forest = Forest(dk["input_len"], dk["output_len"], t(z["in_value"]), t(z["in_type"]), t(z["in_size"]))
    pop, L = z["in_value"].shape
    par = meta["params"]
    rate = par["mutation_rate"]
    kind = meta["operator"]
    mask = t(log[0]) < rate
    if kind == "DeleteMutation":
        mask = mask & (forest.batch_subtree_size[:, 0] > 1)
    idx = torch.nonzero(mask).squeeze(1)

    def per_tree(a, fill=0):
        """a draw made for the mutating trees -> one entry per tree of the population"""
        a = t(a)
        full = torch.full((pop,) + tuple(a.shape[1:]), fill, dtype=a.dtype, device=dev)
        full[idx] = a
        return full

    if kind == "HoistMutation":
        out = HoistMutation(rate).apply(forest, mask, per_tree(log[1]), per_tree(log[2]))
    elif kind == "DeleteMutation":
        op = DeleteMutation(rate, par.get("max_mutatable_size"))
        out = op.apply(forest, mask, per_tree(log[1]), nth_childs=per_tree(log[2]))
    elif kind == "InsertMutation":
        op = InsertMutation(rate, GenerateDescriptor(**meta["op_descriptor"]))
        out = op.apply(forest, mask, per_tree(log[1]), t(log[2]).to(torch.uint32), new_positions=per_tree(log[3]))
    elif kind in ("SinglePointMutation", "MultiPointMutation"):
        modify = par.get("modify_output", False)
        if kind == "SinglePointMutation":
            op = SinglePointMutation(rate, desc, modify_output=modify)
            targets = op.targets(forest, mask, per_tree(log[1]))
            rest = log[2:]
        else:
            op = MultiPointMutation(rate, desc, par["mutation_intensity"], modify_output=modify)
            targets = op.targets(forest, mask, per_tree(log[1], fill=2.0))   # (m, 1) uniforms: one per tree
            rest = log[2:]
        names = ["u_uf", "u_bf", "u_tf"] + (["out_idx"] if modify else []) + ["var_idx", "const_idx"]
        assert len(rest) == len(names), (case, len(rest))
        draws = {}
        for name, a in zip(names, rest):
            a = t(a)
            full = torch.zeros((pop, L), dtype=a.dtype, device=dev)
            full[targets] = a          # the reference draws one number per target, in row-major order of the mutating trees
            draws[name] = full
        draws.setdefault("out_idx", None)
        out = op.apply(forest, targets, draws)
    elif kind == "SingleConstMutation":
        op = SingleConstMutation(rate, desc)
        targets = op.targets(forest, mask, per_tree(log[1]))
        out = op.apply(forest, targets, per_tree(log[2])[:, None].expand(pop, L))
    elif kind == "MultiConstMutation":
        op = MultiConstMutation(rate, desc, par["mutation_intensity"])
        targets = op.targets(forest, mask, per_tree(log[1], fill=2.0))
        full = torch.zeros((pop, L), dtype=torch.int64, device=dev)
        full[targets] = t(log[2])
        out = op.apply(forest, targets, full)
    else:
        raise AssertionError(kind)
    return out, (z["out_value"], z["out_type"], z["out_size"])


def assert_same(out, want, case):
    got = (out.batch_node_value.cpu().numpy(), out.batch_node_type.cpu().numpy(), out.batch_subtree_size.cpu().numpy())
    assert np.array_equal(got[2][:, 0], want[2][:, 0]), f"{case}: tree lengths differ"
    live = np.arange(got[2].shape[1])[None, :] < got[2][:, :1].astype(np.int64)   # the reference leaves the tails undefined
    for name, a, b in zip(("value", "type", "size"), got, want):
        a = a.view(np.uint32) if a.dtype == np.float32 else a
        b = b.view(np.uint32) if b.dtype == np.float32 else b
        bad = np.argwhere((a != b) & live)
        assert bad.size == 0, f"{case}: {name} differs at tree {bad[0][0]} node {bad[0][1]} ({len(bad)} entries)"
