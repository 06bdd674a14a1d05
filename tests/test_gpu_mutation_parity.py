"""GPU: the structural / point mutation operators through the real kernels (tree_crossover, tree_mutate, tree_generate)
against recorded runs of the REFERENCE's Python operators — see tests/test_mutation_parity.py and
tests/golden/make_mutation_golden.py.  Bit-for-bit on the live prefix of every tree."""
import pytest

import mutation_replay as mr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", mr.cases())
def test_operator_reproduces_the_reference_on_the_device(case):
    import torch

    assert torch.cuda.is_available()
    import evogp_amd  # noqa: F401

    out, want = mr.replay(case, "cuda:0")
    assert out.batch_node_value.is_cuda
    mr.assert_same(out, want, case)
