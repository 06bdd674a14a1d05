"""CPU: this repository's structural / point mutation operators against the REFERENCE's own Python operators
(hoist.py:43-75, insert.py:45-85, delete.py:44-105, single_point.py:43-126, multi_point.py:46-143,
single_const.py:39-98, multi_const.py:43-95).  The golden files hold a recorded run of each reference operator — input
population, every random number it drew, its result (tests/golden/make_mutation_golden.py).  Here the same draws go into
`apply` with the ops on the TEST-ONLY oracle back end; tests/test_gpu_mutation_parity.py does the same on the GPU through
the real kernels.  Bit-for-bit on the live prefix of every tree."""
import pytest

import cpu_ops
import mutation_replay as mr


@pytest.fixture(scope="module", autouse=True)
def _cpu_ops():
    cpu_ops.register()
    from evogp_amd.tree import default_device, set_default_device

    old = default_device()
    yield
    set_default_device(old)


@pytest.mark.parametrize("case", mr.cases())
def test_operator_reproduces_the_reference(case):
    out, want = mr.replay(case, "cpu")
    mr.assert_same(out, want, case)
