"""The tracer's own forms of sqrt(x) and 1 / sqrt(x) (rt_device_math.hpp: the compiler's square-root sequence without
its range scaling, and a reciprocal refined from it) must give the bits of the IEEE operations they replace - the
parity of hit distances and normals with the CPU checker rests on that.  rt_selftest_math compares the two forms on
the GPU over 5 classes of 2^26 arguments each; the sixth count is the 3d sky colour's division by 100 in three
instructions against the IEEE division, over every float it is used on (exhaustive)."""
import pytest

pytestmark = pytest.mark.gpu


def test_exact_sqrt_and_reciprocal_sqrt_have_ieee_bits(gpu_ctx):
    assert gpu_ctx.selftest_math(26) == [0, 0, 0, 0, 0, 0]
    assert gpu_ctx.selftest_math(20, seed=99) == [0, 0, 0, 0, 0, 0]
