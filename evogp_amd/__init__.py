"""evogp_amd — MI355X (gfx950) native tree-evaluation and genetic-operation engine behind the
EvoGP operator boundary (``torch.ops.evogp_cuda.*``) and the ``evogp.tree / algorithm / problem /
pipeline`` API surface.  See DESIGN.md and INTEGRATION.md."""
from . import ops  # noqa: F401  (loads evogp_amd/lib/libevogp_hip.so and registers the ops)
from . import tree, algorithm, problem, pipeline  # noqa: F401

__version__ = "0.1.0"
