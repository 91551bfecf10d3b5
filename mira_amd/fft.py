"""Host-side mirror of the reference's src/fft.rs for bn256::Fr, running on the GPU.

Same function names and argument meaning: arrays are (2^log_n, 4) uint64 Montgomery field
elements, transformed natural order -> natural order.  The reference works in place on a slice
and panics on bad sizes (src/fft.rs:13,65); here the functions return the transformed array and
raise AssertionError with the reference's message.
"""
import ctypes

import numpy as np

from . import _lib

FR_S = 28


def _prep(a, log_n):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).copy()
    assert log_n <= FR_S, f"k={log_n} should no larger than F::S={FR_S}"
    assert len(a) == 1 << log_n, f"assertion failed: n == 1 << log_n ({len(a)} vs {1 << log_n})"
    return a


def get_omega_or_inv(k, is_inverse, lib=None):
    """src/fft.rs:12-23"""
    assert k <= FR_S, f"k={k} should no larger than F::S={FR_S}"
    lib = lib or _lib.load()
    out = np.empty(4, dtype=np.uint64)
    lib.check(lib.c.mira_get_omega_or_inv(k, int(bool(is_inverse)), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def best_fft(a, omega, log_n, lib=None):
    """src/fft.rs:51-115"""
    lib = lib or _lib.load()
    a = _prep(a, log_n)
    omega = np.ascontiguousarray(omega, dtype=np.uint64)
    lib.check(lib.c.mira_ntt_bn256_fr(a.ctypes.data_as(ctypes.c_void_p), log_n, omega.ctypes.data_as(ctypes.c_void_p)))
    return a


def _call(name, a, log_n, lib):
    lib = lib or _lib.load()
    a = _prep(a, log_n)
    lib.check(getattr(lib.c, name)(a.ctypes.data_as(ctypes.c_void_p), log_n))
    return a


def fft(a, log_n, lib=None):
    """src/fft.rs:160-162"""
    return _call("mira_fft_bn256_fr", a, log_n, lib)


def ifft(a, log_n, lib=None):
    """src/fft.rs:165-174"""
    return _call("mira_ifft_bn256_fr", a, log_n, lib)


def _log2(a):
    n = len(np.asarray(a).reshape(-1, 4))
    assert n and n & (n - 1) == 0, "assertion failed: a.len().is_power_of_two()"
    return n.bit_length() - 1


def coset_fft(a, lib=None):
    """src/fft.rs:178-185"""
    return _call("mira_coset_fft_bn256_fr", a, _log2(a), lib)


def coset_ifft(a, lib=None):
    """src/fft.rs:189-196"""
    return _call("mira_coset_ifft_bn256_fr", a, _log2(a), lib)


def fft_device(d_a, log_n, lib=None):
    lib = lib or _lib.load()
    lib.check(lib.c.mira_fft_bn256_fr_device(ctypes.c_void_p(d_a), log_n))


def ifft_device(d_a, log_n, lib=None):
    lib = lib or _lib.load()
    lib.check(lib.c.mira_ifft_bn256_fr_device(ctypes.c_void_p(d_a), log_n))
