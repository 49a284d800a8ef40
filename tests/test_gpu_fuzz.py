"""A short run of the seeded differential campaign (tests/fuzz_parity.py) over every kernel family: random shapes and
argument combinations against exact integer references (int8 kernels), the numpy oracle (quantizer) and float64 products
(floating-point activations).  The campaign prints the arguments of every mismatch."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family,cases", [("gemm", 250), ("conv", 250), ("gemm_w8", 120), ("conv_w8", 120),
                                          ("conv_general", 150), ("quantizer", 80), ("layers", 60), ("layers_wa", 60), ("pool", 100), ("activation", 100), ("small_kernels", 200)])
def test_random_campaign(family, cases):
    import fuzz_parity
    assert fuzz_parity.run([family], cases, 20240) == 0
