"""The order-independent reduction accumulator of the Krylov solvers (csrc/fg_internal.h FgDacc).

The reference's dot products are cuBLAS calls in a fixed order (cg_solver_kernel.cu:277,317), so a replayed state steps to
the same result (envs/fluid_env.py:1320-1363).  Here every workgroup contributes one partial; the partials are split exactly
into 42-bit fixed-point words and added with integer atomics, so the sum must not depend on their order.  The CPU half runs the
very split / read-back code the kernels run (fg_dacc_host_sum); the GPU half lets the hardware schedule the contributions."""
import ctypes
import math

import numpy as np
import pytest

from fluidgym_amd import _lib as L

D = ctypes.POINTER(ctypes.c_double)


def host_sum(values, plain=0.0):
    v = np.ascontiguousarray(values, np.float64)
    out = ctypes.c_double()
    L.check(L.load().fg_dacc_host_sum(v.ctypes.data_as(D), v.size, plain, ctypes.byref(out)))
    return out.value


def _mixed(rng, n):
    """fp32-valued partials (what fg_block_sum hands over) over 25 orders of magnitude, both signs."""
    mag = 10.0 ** rng.uniform(-18, 7, n)
    return (rng.standard_normal(n) * mag).astype(np.float32).astype(np.float64)


def test_sum_is_exact_and_order_independent():
    rng = np.random.default_rng(0)
    for n in (1, 7, 256, 20000):
        v = _mixed(rng, n)
        ref = math.fsum(v)                       # correctly rounded exact sum
        s0 = host_sum(v)
        assert abs(s0 - ref) <= 2.0 ** -93 * n + abs(ref) * 2.0 ** -52, (n, s0, ref)
        for _ in range(5):
            assert host_sum(rng.permutation(v)) == s0          # bit-identical in any order
    # cancellation: the partial sums pass through 1e7 and come back to 1e-12
    v = np.array([1.0e7, 3.0e-12, -1.0e7, 2.5e-12], np.float64)
    assert host_sum(v) == host_sum(v[::-1]) == 5.5e-12
    # double-valued contributions (53 bits) split exactly as well
    w = rng.standard_normal(1000) * 10.0 ** rng.uniform(-10, 10, 1000)
    assert abs(host_sum(w) - math.fsum(w)) <= 2.0 ** -93 * 1000 + abs(math.fsum(w)) * 2.0 ** -52


def test_plain_part_and_reset():
    assert host_sum([], plain=0.0) == 0.0
    assert host_sum([], plain=3.25) == 3.25                    # a scalar parked with acc_st reads back unchanged
    assert host_sum([1.5, -0.25], plain=2.0) == 3.25
    assert math.isnan(host_sum([1.0], plain=float("nan")))


def test_non_finite_poisons_and_out_of_range_saturates():
    for bad in (float("nan"), float("inf"), -float("inf")):
        assert math.isnan(host_sum([1.0, bad, 2.0]))
    # finite contributions beyond the window saturate one by one: a diverging-but-finite solve (|r.r| ~ 1e23 and above) keeps
    # reading as a large FINITE residual -- "not converged", which return-best and the retry ladder handle, not "not finite"
    cap = 2.0 ** 75 * (1.0 - 2.0 ** -30)
    for big in (2.0 ** 75, 1.0e23, 3.0e25, 1.0e300):
        s = host_sum([1.0, big, 2.0])
        assert math.isfinite(s) and abs(s - min(big, cap)) <= 3.0 + big * 2.0 ** -52, (big, s)
        assert host_sum([2.0, 1.0, big]) == s                          # still order-independent
        assert host_sum([-big]) == -host_sum([big])
    assert host_sum([1.0e23] * 4096) == host_sum([1.0e23] * 4096)       # many large ones: no integer wrap (2^20 contributions fit)
    assert math.isfinite(host_sum([2.0 ** 80] * 65536))
    assert host_sum([2.0 ** 74, -(2.0 ** 74), 1.0]) == 1.0         # the largest admitted magnitude
    assert host_sum([2.0 ** -92]) == 2.0 ** -92 and host_sum([2.0 ** -94]) == 0.0   # unit of the last word (rounded to nearest)


@pytest.mark.gpu
def test_device_atomics_reproduce_the_host_sum_bit_for_bit():
    rng = np.random.default_rng(1)
    v = _mixed(rng, 200_000)
    reps = 8
    out = (ctypes.c_double * reps)()
    L.check(L.load().fg_dacc_device_sum(v.ctypes.data_as(D), v.size, 0.75, reps, out, None))
    expect = host_sum(v, plain=0.75)
    assert all(x == expect for x in out), (list(out), expect)
    v[1234] = float("inf")
    L.check(L.load().fg_dacc_device_sum(v.ctypes.data_as(D), v.size, 0.0, 2, out, None))
    assert math.isnan(out[0]) and math.isnan(out[1])
