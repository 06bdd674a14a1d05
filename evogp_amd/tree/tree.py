"""Tree — a single-tree view over three rows of a Forest.

API follows src/evogp/tree/tree.py:9-140 (``forward``, ``SR_fitness``, ``to_forest``,
``random_generate``); presentation helpers (``to_infix``, ``to_sympy_expr``) are small
re-implementations, ``to_png`` is out of scope.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .descriptor import GenerateDescriptor
from .utils import FUNCS_DISPLAY, NType, check_tensor, decode_func, func_arity

_SR_MODES_TREE = {"hybrid parallel": 3, "data parallel": 1, "tree parallel": 2, "auto": 4}  # tree.py:94-101


class Tree:
    def __init__(self, input_len, output_len, node_value: Tensor, node_type: Tensor, subtree_size: Tensor):
        self.input_len = input_len
        self.output_len = output_len
        self.max_tree_len = node_value.shape[0]
        for name, t in (("node_value", node_value), ("node_type", node_type), ("subtree_size", subtree_size)):
            assert t.shape == (self.max_tree_len,), f"{name} shape should be {self.max_tree_len}, but got {t.shape}"
        self.node_value = node_value
        self.node_type = node_type
        self.subtree_size = subtree_size

    @staticmethod
    def random_generate(descriptor: GenerateDescriptor) -> "Tree":
        from .forest import Forest

        return Forest.random_generate(pop_size=1, descriptor=descriptor)[0]

    def forward(self, x: Tensor) -> Tensor:
        """x: (input_len,) or (batch, input_len) -> (output_len,) or (batch, output_len)."""
        x = check_tensor(x, self.node_value.device)
        assert x.dim() <= 2, f"x dim should be <= 2, but got {x.dim()}"
        squeeze = x.dim() == 1
        if squeeze:
            x = x.unsqueeze(0)
        assert x.shape[1] == self.input_len, f"x shape should be {self.input_len}, but got {x.shape[1]}"
        # one tree, many rows == the non-replicating batch op with pop = 1 (the reference replicates
        # the tree `batch` times and calls tree_evaluate, tree.py:56-72)
        res = torch.ops.evogp_hip.tree_batch_evaluate(
            1, x.shape[0], self.max_tree_len, self.input_len, self.output_len,
            self.node_value[None, :].contiguous(), self.node_type[None, :].contiguous(),
            self.subtree_size[None, :].contiguous(), x.contiguous().to(torch.float32),
        )[0]
        return res[0] if squeeze else res

    def SR_fitness(self, inputs: Tensor, labels: Tensor, use_MSE: bool = True, execute_mode: str = "auto") -> Tensor:
        inputs, labels = check_tensor(inputs, self.node_value.device), check_tensor(labels, self.node_value.device)
        assert execute_mode in _SR_MODES_TREE, (
            f"execute_mode should be one of {list(_SR_MODES_TREE)}, but got {execute_mode}")
        n = inputs.shape[0]
        assert inputs.shape == (n, self.input_len), (
            f"inputs shape should be ({n}, {self.input_len}), but got {inputs.shape}")
        assert labels.shape == (n, self.output_len), (
            f"outputs shape should be ({n}, {self.output_len}), but got {labels.shape}")
        return torch.ops.evogp_cuda.tree_SR_fitness(
            1, n, self.max_tree_len, self.input_len, self.output_len, use_MSE,
            self.node_value[None, :].contiguous(), self.node_type[None, :].contiguous(),
            self.subtree_size[None, :].contiguous(), inputs.contiguous(), labels.contiguous(),
            _SR_MODES_TREE[execute_mode],
        )

    def to_forest(self):
        from .forest import Forest

        return Forest(self.input_len, self.output_len, self.node_value[None, :], self.node_type[None, :],
                      self.subtree_size[None, :])

    # ---- presentation -------------------------------------------------------------------------
    def _nodes(self):
        n = int(self.subtree_size[0])
        return (self.node_value[:n].cpu(), self.node_type[:n].cpu())

    def to_infix(self) -> str:
        """Infix string of the expression (single-output view; OUT flags are shown as out[i]:)."""
        values, types = self._nodes()
        stack = []
        for v, t in zip(reversed(list(values)), reversed(list(types))):
            base = int(t) & NType.TYPE_MASK
            if base == NType.VAR:
                stack.append(f"x{int(v)}")
            elif base == NType.CONST:
                stack.append(f"{float(v):.2f}")
            else:
                fid, out = decode_func(v, t)
                args = [stack.pop() for _ in range(func_arity(fid) if base != NType.TFUNC else 3)]
                name = FUNCS_DISPLAY[fid] if fid < len(FUNCS_DISPLAY) else f"f{fid}"
                if base == NType.BFUNC and name in ("+", "-", "*", "/", "<", ">", "<=", ">="):
                    s = f"({args[0]} {name} {args[1]})"
                else:
                    s = f"{name}({', '.join(args)})"
                stack.append(s if out < 0 else f"out[{out}]:{s}")
        return stack.pop() if stack else ""

    def to_sympy_expr(self, symbol_names=None):
        """sympy expression of a single-output tree (arithmetic/elementary functions only)."""
        import sympy as sp

        values, types = self._nodes()
        names = symbol_names or [f"x{i}" for i in range(self.input_len)]
        syms = sp.symbols(names)
        if not isinstance(syms, (list, tuple)):
            syms = [syms]
        table = {
            "if": lambda a, b, c: sp.Piecewise((b, a > 0), (c, True)), "+": lambda a, b: a + b, "-": lambda a, b: a - b,
            "*": lambda a, b: a * b, "/": lambda a, b: a / b, "loose_div": lambda a, b: a / b, "pow": sp.Pow,
            "loose_pow": lambda a, b: sp.Pow(sp.Abs(a), b), "max": sp.Max, "min": sp.Min, "<": sp.Lt, ">": sp.Gt,
            "<=": sp.Le, ">=": sp.Ge, "sin": sp.sin, "cos": sp.cos, "tan": sp.tan, "sinh": sp.sinh, "cosh": sp.cosh,
            "tanh": sp.tanh, "log": sp.log, "loose_log": lambda a: sp.log(sp.Abs(a)), "exp": sp.exp,
            "inv": lambda a: 1 / a, "loose_inv": lambda a: 1 / a, "neg": lambda a: -a, "abs": sp.Abs, "sqrt": sp.sqrt,
            "loose_sqrt": lambda a: sp.sqrt(sp.Abs(a)),
        }
        stack = []
        for v, t in zip(reversed(list(values)), reversed(list(types))):
            base = int(t) & NType.TYPE_MASK
            if base == NType.VAR:
                stack.append(syms[int(v)])
            elif base == NType.CONST:
                stack.append(sp.Float(float(v)))
            else:
                fid, _ = decode_func(v, t)
                args = [stack.pop() for _ in range(func_arity(fid))]
                stack.append(table[FUNCS_DISPLAY[fid]](*args))
        return stack.pop()

    def __str__(self):
        return self.to_infix()

    __repr__ = __str__
