"""``torch.ops.evogp_cuda.*`` — the reference's operator boundary, registered from C++.

Loading ``evogp_amd/lib/libevogp_torch.so`` (csrc/torch_ops.cpp) runs its static registrars, exactly as importing the
reference's extension module does (src/evogp/tree/__init__.py:2, torch_wrapper.cu:287-307):

* ``TORCH_LIBRARY(evogp_cuda)`` with the reference's five schemas verbatim (torch_wrapper.cu:294-298) and
  ``TORCH_LIBRARY_IMPL(evogp_cuda, CUDA)`` — ``tree_generate / tree_mutate / tree_crossover / tree_evaluate /
  tree_SR_fitness`` — so code written against them runs unchanged.  ROCm builds of PyTorch dispatch HIP tensors on the
  ``CUDA`` key; there is deliberately no CPU implementation (the reference has none, torch_wrapper.cu:301) and no fallback;
* the extra ops of this engine in the ``evogp_hip`` namespace (no counterpart in the reference):
  ``tree_generate_offset`` (tree-index offset for sharded populations), ``tree_generate_masked`` and ``breed_default`` /
  ``breed_default_rows`` (the default generation step in two launches), ``tree_batch_evaluate``
  (non-replicating ``Forest.batch_forward``), ``tree_batch_argmax_count`` (fused classification accuracy).

Every implementation validates like the reference's wrapper, allocates its outputs and calls the C ABI of
include/evogp_hip.h (``libevogp_hip.so``, loaded by ``_lib``) on torch's current stream; no Python runs per call.
"""
from __future__ import annotations

import os

import torch

from . import _lib  # the engine itself: fails loudly when the HIP library is missing

MAX_STACK = 1024
MAX_FULL_DEPTH = 10
FUNC_END = 29

TORCH_LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libevogp_torch.so")

if not os.path.exists(TORCH_LIB_PATH):
    raise ImportError(
        f"{TORCH_LIB_PATH} is missing: build the engine first (python -c 'import __graft_entry__ as g; g.build()'  or  "
        "make -C evogp_amd/csrc).  evogp_amd has no Python or CPU fallback for its operators.")
torch.ops.load_library(TORCH_LIB_PATH)
