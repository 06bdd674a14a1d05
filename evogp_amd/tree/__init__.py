"""evogp_amd.tree — the tensorised population layer (reference: src/evogp/tree/__init__.py:1-8)."""
from .. import ops as _ops  # registers torch.ops.evogp_cuda.* (fails loudly if the HIP engine is missing)
from .descriptor import GenerateDescriptor
from .tree import Tree
from .forest import Forest
from .combined import CombinedForest, CombinedTree
from .utils import MAX_STACK, randint, NType, set_default_device, default_device

__all__ = ["GenerateDescriptor", "Tree", "Forest", "CombinedForest", "CombinedTree", "MAX_STACK", "randint", "NType", "set_default_device",
           "default_device"]
