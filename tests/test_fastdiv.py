"""The host's check of a grid extent before the look-ups may replace `q / extent` by a multiplication with the rounded reciprocal
and one exact-residual correction (csrc/vpt_fastdiv.h; the device side is csrc/vpt_trace_common.h: to_unit, whose GPU test is
tests/test_gpu_edge.py::test_quotient_by_checked_reciprocal_is_the_division).  Host only: no GPU needed."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture()
def lib(pkg):
    lib = pkg.load_library()
    lib.vpt_test_fast_div_ok.argtypes = [C.c_float, C.c_float]
    lib.vpt_test_fast_div_ok.restype = C.c_int
    return lib


def _rcp(d):
    return float(np.float32(1.0) / np.float32(d))


def test_grid_extents_pass(lib):
    # the bench scenes' extents (dragon, fireball, cloud), small and awkward ones, the ends of the admitted range
    for d in (1, 2, 3, 7, 77, 185, 300, 511, 704, 1024, 1216, 4095, 65535, 65536):
        assert lib.vpt_test_fast_div_ok(float(d), _rcp(d)) == 1, d


def test_a_wrong_reciprocal_is_caught(lib):
    # (one ulp off can still pass -- the correction step absorbs it; the check decides on the bits, not on the recipe)
    for d in (3, 300, 1216):
        r = np.float32(_rcp(d))
        assert lib.vpt_test_fast_div_ok(float(d), float(r * np.float32(1.0 + 2.0 ** -10))) == 0, d
        assert lib.vpt_test_fast_div_ok(float(d), float(r * np.float32(1.0 - 2.0 ** -10))) == 0, d


def test_outside_the_admitted_range(lib):
    # below 1 / above 2^16 the device's guard on |q| no longer keeps every intermediate normal: not offered
    for d in (0.0, 0.5, 65537.0, 1e9, float("nan"), float("inf"), -3.0):
        assert lib.vpt_test_fast_div_ok(d, 1.0) == 0, d


def test_check_agrees_with_a_numpy_restatement_on_a_sample(lib):
    """the same sequence in numpy (float64 products of binary32 values are exact; the residual is exactly representable) on the
    significands most likely to break it: those whose quotient lies closest to a rounding boundary"""
    rng = np.random.default_rng(5)
    for d in (3.0, 77.0, 300.0, 1216.0):
        q = (np.float32(1.0) + rng.integers(0, 1 << 23, 200000).astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)
        r = np.float32(1.0) / np.float32(d)
        y = (q * r).astype(np.float32)
        e = (q.astype(np.float64) - np.float64(d) * y.astype(np.float64))
        assert np.array_equal(e, e.astype(np.float32).astype(np.float64))            # exact residual
        exact = y.astype(np.float64) + e * np.float64(r)                                # (53 bits: enough away from float ties here)
        u = exact.astype(np.float32)
        ref = (q / np.float32(d)).astype(np.float32)
        tie = np.abs(exact - u.astype(np.float64)) == np.abs(np.spacing(u).astype(np.float64)) * 0.5
        assert np.array_equal(u[~tie], ref[~tie])
        assert lib.vpt_test_fast_div_ok(d, float(r)) == 1
