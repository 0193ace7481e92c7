"""The exact-libm layer (csrc/rl_glibc_math.h): glibc 2.35's float routines restated for the device.

CPU part: tools/check_glibc_math.cc compiled with g++ and run on reduced sweeps against this host's libm
(the full 2^32-input sweeps take minutes: run the tool by hand).  GPU part: the DEVICE evaluates the
functions on sampled and edge-case inputs and must return the host libm's bits."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

import helpers

ROOT = helpers.ROOT


def test_restatement_matches_host_libm_on_a_strided_sweep(tmp_path):
    exe = str(tmp_path / "check_glibc_math")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-pthread", "-DRL_CHECK_STRIDE=251",
                           os.path.join(ROOT, "tools", "check_glibc_math.cc"), "-o", exe])
    out = subprocess.run([exe, "expf", "logf", "sinf", "cosf", "sincos", "tanf", "acosf", "asinf", "atanf", "quick"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "ALL BIT-EXACT" in out.stdout


FUNCS = {0: np.sin, 1: np.cos, 2: np.tan, 3: np.arccos, 4: np.arcsin, 6: np.exp, 7: np.log, 9: np.sin, 10: np.cos, 11: np.sqrt}


def _libm():
    return C.CDLL("libm.so.6")


def _host(fn, x, y=None):
    """Host libm through ctypes, one call per element (numpy's own loops may use SIMD kernels that round differently)."""
    m = _libm()
    names = {0: "sinf", 1: "cosf", 2: "tanf", 3: "acosf", 4: "asinf", 5: "atan2f", 6: "expf", 7: "logf", 8: "powf", 9: "sinf", 10: "cosf", 11: "sqrtf"}
    f = getattr(m, names[fn])
    f.restype = C.c_float
    f.argtypes = [C.c_float] * (2 if fn in (5, 8) else 1)
    if fn in (5, 8):
        return np.array([f(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    return np.array([f(float(a)) for a in x], np.float32)


@pytest.mark.gpu
def test_device_math_is_bit_identical_to_host_libm(gpu_lib):
    rng = np.random.RandomState(0)
    n = 20000
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-30, 1e-20, 3.1415927, 1.5707964, 0.7853982, 6.2831855, 87.0, -87.0, 88.8,
                     -104.0, 1e30, 0.99999994, 1.0000001, 0.975, 0.6744, np.inf, -np.inf, np.nan, 2.0, 120.0, 1e5], np.float32)
    cases = {
        0: np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-200, 200, n // 4), edge]),
        1: np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-200, 200, n // 4), edge]),
        2: np.concatenate([rng.uniform(0, np.pi, n), rng.uniform(-50, 50, n // 4), edge]),
        3: np.concatenate([rng.uniform(-1, 1, n), 1 - 10.0 ** rng.uniform(-7, 0, n // 4), edge]),
        4: np.concatenate([rng.uniform(-1, 1, n), edge]),
        6: np.concatenate([rng.uniform(-20, 5, n), rng.uniform(-110, 90, n // 4), edge]),
        7: np.concatenate([10.0 ** rng.uniform(-38, 38, n), rng.uniform(0, 2, n), edge]),
        9: np.concatenate([rng.uniform(0, 6.2832, n), edge]),
        10: np.concatenate([rng.uniform(0, 6.2832, n), edge]),
        11: np.concatenate([10.0 ** rng.uniform(-38, 38, n), edge]),
    }
    for fn, x in cases.items():
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        assert gpu_lib.RaylibAMD_EvalDeviceMath(fn, x.ctypes.data_as(C.POINTER(C.c_float)), None, len(x), out.ctypes.data_as(C.POINTER(C.c_float))) == 1
        want = _host(fn, x)
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert same.all(), "fn %d: %d of %d differ, e.g. x=%r device=%r host=%r" % (
            fn, (~same).sum(), len(x), x[~same][:3], out[~same][:3], want[~same][:3])
    # two-argument functions
    xa = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 50, n // 2), rng.uniform(-1e-6, 1e-6, 100), edge[:20]]).astype(np.float32)
    ya = np.concatenate([np.full(n // 2, 5.0), np.full(n // 2, 2.2), rng.uniform(0.3, 1.5, n // 2), rng.uniform(-3, 3, 100), np.full(20, 5.0)]).astype(np.float32)
    for fn, (x, y) in {8: (xa, ya), 5: (rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, n).astype(np.float32))}.items():
        out = np.zeros_like(x)
        assert gpu_lib.RaylibAMD_EvalDeviceMath(fn, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)), len(x),
                                                out.ctypes.data_as(C.POINTER(C.c_float))) == 1
        want = _host(fn, x, y)
        same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
        assert same.all(), "fn %d: %d differ" % (fn, (~same).sum())
    # IEEE division on the device
    x = rng.uniform(-10, 10, n).astype(np.float32); y = rng.uniform(-3, 3, n).astype(np.float32)
    out = np.zeros_like(x)
    gpu_lib.RaylibAMD_EvalDeviceMath(12, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)), n, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out.view(np.uint32), (x / y).astype(np.float32).view(np.uint32))


@pytest.mark.gpu
def test_short_exact_reciprocal_and_square_root_on_every_float(gpu_lib):
    """normalize, 1 / tan, 1 / (1 + ...), the ray's 1 / d: on the device `1.0f / x` and `sqrtf(x)` run as short sequences (csrc/rl_glibc_math.h
    rcp1_ / sqrtf_: v_rcp + one Newton step, v_rsq + one residual step, inside a range guard) instead of the compiler's 36- and 57-cycle IEEE
    expansions.  They must BE those expansions' results: all 2^32 bit patterns are compared on the device, inside the product library."""
    for which, name in ((0, "1.0f / x"), (1, "sqrtf(x)"), (2, "a / b with RN(1 / b) in hand (div_by_)"), (3, "the triangle test's short barycentric form")):
        bad, first = C.c_uint64(1), C.c_uint64(0)
        assert gpu_lib.RaylibAMD_VerifyExactMath(which, C.byref(bad), C.byref(first)) == 1
        print("%s: %d of 2^32 inputs differ from the IEEE expansion" % (name, bad.value))
        assert bad.value == 0, "%s: %d mismatches, first at bits 0x%08x" % (name, bad.value, first.value)
    # and through the array hook, against the host: 1 / x and sqrt are correctly rounded on both sides
    rng = np.random.RandomState(3)
    x = np.concatenate([rng.uniform(-4, 4, 5000), 10.0 ** rng.uniform(-44, 38, 5000), -(10.0 ** rng.uniform(-44, 38, 2000)),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1.17549435e-38, 3.4028235e38, 1.7014118e38, 8.5e37, 2 ** -100, 2 ** -101, 2 ** -102])]).astype(np.float32)
    out = np.zeros_like(x)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    with np.errstate(all="ignore"):
        assert gpu_lib.RaylibAMD_EvalDeviceMath(14, fp(x), None, len(x), fp(out)) == 1
        assert helpers.same(out, (np.float32(1.0) / x).astype(np.float32)).all()
        assert gpu_lib.RaylibAMD_EvalDeviceMath(15, fp(x), None, len(x), fp(out)) == 1
        assert helpers.same(out, np.sqrt(x).astype(np.float32)).all()
