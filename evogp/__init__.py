"""``evogp`` — the reference's import root, served by the MI355X engine.

Scripts written against EMI-Group/evogp (``from evogp.tree import Forest, GenerateDescriptor``, ``from evogp.algorithm
import ...``, ``from evogp.problem import SymbolicRegression``, ``from evogp.pipeline import StandardPipeline``,
``import evogp.evogp_cuda``; /root/reference/src/evogp/__init__.py, example/*.py) run unmodified: every sub-package is the
``evogp_amd`` one under the reference's name, and importing this package loads the HIP engine and registers
``torch.ops.evogp_cuda.*`` (the reference's ``__init__`` does that through ``import evogp.evogp_cuda``)."""
import sys as _sys

import evogp_amd as _impl
from evogp_amd import algorithm, pipeline, problem, tree  # noqa: F401

for _name in ("tree", "algorithm", "problem", "pipeline"):
    _pkg = getattr(_impl, _name)
    _sys.modules[f"{__name__}.{_name}"] = _pkg
    # sub-modules scripts reach by path (e.g. ``from evogp.tree.utils import ...``)
    for _full, _mod in list(_sys.modules.items()):
        if _full.startswith(f"evogp_amd.{_name}.") and _mod is not None:
            _sys.modules[__name__ + _full[len("evogp_amd"):]] = _mod

from . import evogp_cuda  # noqa: E402,F401

__version__ = _impl.__version__
