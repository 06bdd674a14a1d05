"""Encoding tables and small helpers of the Forest tensors.

The numeric values are the wire format shared with the kernels (reference:
src/evogp/cuda/defs.h:10-57, mirrored in src/evogp/tree/utils.py:14-136); the helper semantics
follow src/evogp/tree/utils.py:261-310.
"""
from __future__ import annotations

import torch

DELTA = 1e-9
MAXVAL = 1e9
MAX_STACK = 1024
MAX_FULL_DEPTH = 10


class NType:
    """Node types (low 7 bits) and the output-node flag (bit 7)."""

    VAR = 0
    CONST = 1
    UFUNC = 2
    BFUNC = 3
    TFUNC = 4
    TYPE_MASK = 0x7F
    OUT_NODE = 1 << 7
    UFUNC_OUT = UFUNC + OUT_NODE
    BFUNC_OUT = BFUNC + OUT_NODE
    TFUNC_OUT = TFUNC + OUT_NODE


# function id -> user-facing name; ids 0 ternary, 1..13 binary, 14..28 unary
FUNCS_NAMES = [
    "if",
    "+", "-", "*", "/", "loose_div", "pow", "loose_pow", "max", "min", "<", ">", "<=", ">=",
    "sin", "cos", "tan", "sinh", "cosh", "tanh", "log", "loose_log", "exp", "inv", "loose_inv", "neg", "abs",
    "sqrt", "loose_sqrt",
]


class Func:
    TF_START = 0
    BF_START = 1
    UF_START = 14
    END = 29


for _i, _n in enumerate(
    ["IF", "ADD", "SUB", "MUL", "DIV", "LOOSE_DIV", "POW", "LOOSE_POW", "MAX", "MIN", "LT", "GT", "LE", "GE",
     "SIN", "COS", "TAN", "SINH", "COSH", "TANH", "LOG", "LOOSE_LOG", "EXP", "INV", "LOOSE_INV", "NEG", "ABS",
     "SQRT", "LOOSE_SQRT"]
):
    setattr(Func, _n, _i)

FUNCS = list(range(Func.END))
FUNCS_DISPLAY = list(FUNCS_NAMES)


def func_arity(func_id: int) -> int:
    return 3 if func_id < Func.BF_START else (2 if func_id < Func.UF_START else 1)


_DEVICE = None


def default_device() -> torch.device:
    """The device Forest tensors live on.  The reference hard-codes "cuda" (tree/utils.py:280-285);
    on a GPU-less host (CPU-only unit tests of the host logic) tensors stay on the CPU and any
    attempt to run an op fails loudly because only device kernels are registered."""
    global _DEVICE
    if _DEVICE is None:
        _DEVICE = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    return _DEVICE


def set_default_device(device) -> None:
    """Pin the device (e.g. ``cuda:LOCAL_RANK`` in a multi-process run)."""
    global _DEVICE
    _DEVICE = torch.device(device)


def dict2prob(prob_dict) -> torch.Tensor:
    """{function name: weight} -> normalised probability vector over the 29 function ids."""
    assert len(prob_dict) > 0, "Empty probability dictionary"
    prob = torch.zeros(Func.END)
    for name, weight in prob_dict.items():
        assert name in FUNCS_NAMES, f"Unknown function name: {name}, total functions are {FUNCS_NAMES}"
        prob[FUNCS_NAMES.index(name)] = weight
    return prob / prob.sum()


def check_tensor(x, device=None):
    """Tensor on ``device`` (a Forest passes the device of its own tensors, so operands follow the population in a
    multi-GPU process; default: the default device), detached (non-tensors are converted to float32)."""
    device = default_device() if device is None else device
    if not isinstance(x, torch.Tensor):
        return torch.tensor(x, dtype=torch.float32, device=device)
    return x.to(device).detach().requires_grad_(False)


def randint(size, low, high, dtype=torch.int32, device=None, requires_grad=False):
    """Uniform integers in [low, high) drawn as floor(low + U*(high-low)); low/high may be tensors."""
    device = default_device() if device is None else device
    u = torch.rand(size, device=device, requires_grad=requires_grad)
    return (low + u * (high - low)).to(dtype=dtype)


def str_tree(value, node_type, subtree_size) -> str:
    """Prefix listing of the live nodes of one tree."""
    out = []
    for i in range(int(subtree_size[0])):
        t = int(node_type[i]) & NType.TYPE_MASK
        if t == NType.VAR:
            out.append(f"x[{int(value[i])}]")
        elif t == NType.CONST:
            out.append(f"{float(value[i]):.2f}")
        else:
            out.append(FUNCS_NAMES[decode_func(value[i], node_type[i])[0]])
    return " ".join(out)


def decode_func(value, node_type):
    """(function id, output index or -1) of a function node."""
    if int(node_type) & NType.OUT_NODE:
        bits = torch.as_tensor(value, dtype=torch.float32).view(torch.int32).item()
        return bits & 0xFFFF, (bits >> 16) & 0xFFFF
    return int(value), -1
