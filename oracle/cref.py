"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_build/liboracle.so (oracle.c).

Arrays are numpy uint64: field elements (n, 4), affine points (n, 8), Montgomery form,
exactly the in-memory layout the reference hands to best_multiexp / best_fft.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

FIELD_FQ, FIELD_FR = 0, 1
CURVE_BN256, CURVE_GRUMPKIN = 0, 1


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_init()
        for name in ("oracle_msm_naive", "oracle_msm_pippenger", "oracle_commit", "oracle_best_fft", "oracle_fft",
                     "oracle_ifft", "oracle_coset_fft", "oracle_coset_ifft", "oracle_is_on_curve", "oracle_num_threads"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a if shape is None else a.reshape(shape)


def num_threads():
    return lib().oracle_num_threads()


def set_threads(n):
    """Default pool size of the `threads=0` entry points (the reference's rayon pool)."""
    lib().oracle_set_num_threads(int(n))


def to_mont(field, a):
    a = _u64(a).reshape(-1, 4); out = np.empty_like(a)
    lib().oracle_f_to_mont(field, _p(a), _p(out), ctypes.c_size_t(len(a)))
    return out


def from_mont(field, a):
    a = _u64(a).reshape(-1, 4); out = np.empty_like(a)
    lib().oracle_f_from_mont(field, _p(a), _p(out), ctypes.c_size_t(len(a)))
    return out


def f_mul(field, a, b):
    a, b = _u64(a), _u64(b); out = np.empty(4, dtype=np.uint64)
    lib().oracle_f_mul(field, _p(a), _p(b), _p(out)); return out


def f_inv(field, a):
    a = _u64(a); out = np.empty(4, dtype=np.uint64)
    lib().oracle_f_inv(field, _p(a), _p(out)); return out


def generator(curve):
    out = np.empty(8, dtype=np.uint64); lib().oracle_generator(curve, _p(out)); return out


def is_on_curve(curve, pt):
    return bool(lib().oracle_is_on_curve(curve, _p(_u64(pt))))


def ec_add(curve, a, b):
    out = np.empty(8, dtype=np.uint64); lib().oracle_ec_add(curve, _p(_u64(a)), _p(_u64(b)), _p(out)); return out


def ec_mul(curve, k_mont, pt):
    out = np.empty(8, dtype=np.uint64); lib().oracle_ec_mul(curve, _p(_u64(k_mont)), _p(_u64(pt)), _p(out)); return out


def msm_naive(curve, scalars, bases):
    scalars, bases = _u64(scalars).reshape(-1, 4), _u64(bases).reshape(-1, 8)
    assert len(scalars) == len(bases)
    out = np.empty(8, dtype=np.uint64)
    lib().oracle_msm_naive(curve, _p(scalars), _p(bases), ctypes.c_size_t(len(scalars)), _p(out)); return out


def msm_pippenger(curve, scalars, bases, threads=0):
    scalars, bases = _u64(scalars).reshape(-1, 4), _u64(bases).reshape(-1, 8)
    assert len(scalars) == len(bases)
    out = np.empty(8, dtype=np.uint64)
    lib().oracle_msm_pippenger(curve, _p(scalars), _p(bases), ctypes.c_size_t(len(scalars)), threads, _p(out)); return out


class TooLongInput(Exception):
    pass


def commit(curve, ck, v, threads=0):
    """CommitmentKey::commit, src/commitment.rs:78-87"""
    ck, v = _u64(ck).reshape(-1, 8), _u64(v).reshape(-1, 4)
    out = np.empty(8, dtype=np.uint64)
    rc = lib().oracle_commit(curve, _p(ck), ctypes.c_size_t(len(ck)), _p(v), ctypes.c_size_t(len(v)), threads, _p(out))
    if rc:
        raise TooLongInput(f"Can't commit too long input: input len: {len(v)}, but limit is {len(ck)}")
    return out


def get_omega_or_inv(k, is_inverse):
    out = np.empty(4, dtype=np.uint64); lib().oracle_get_omega_or_inv(ctypes.c_uint32(k), int(is_inverse), _p(out)); return out


def _ntt(fn, a, log_n, threads):
    a = _u64(a).reshape(-1, 4).copy()
    assert len(a) == 1 << log_n
    getattr(lib(), fn)(_p(a), ctypes.c_uint32(log_n), threads); return a


def fft(a, log_n, threads=0): return _ntt("oracle_fft", a, log_n, threads)
def ifft(a, log_n, threads=0): return _ntt("oracle_ifft", a, log_n, threads)
def coset_fft(a, log_n, threads=0): return _ntt("oracle_coset_fft", a, log_n, threads)
def coset_ifft(a, log_n, threads=0): return _ntt("oracle_coset_ifft", a, log_n, threads)


def best_fft(a, omega, log_n, threads=0):
    a = _u64(a).reshape(-1, 4).copy()
    lib().oracle_best_fft(_p(a), _p(_u64(omega)), ctypes.c_uint32(log_n), threads); return a


def synth_scalars(curve, n, seed=0x4D495241, kind=0):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().oracle_synth_scalars(curve, ctypes.c_size_t(n), ctypes.c_uint64(seed), kind, _p(out)); return out


def synth_bases(curve, n, seed=0x42415345):
    out = np.empty((n, 8), dtype=np.uint64)
    lib().oracle_synth_bases(curve, ctypes.c_size_t(n), ctypes.c_uint64(seed), _p(out)); return out


def fold_witness(field, w1, w2, r):
    """RelaxedPlonkWitness::fold, W part (src/plonk/mod.rs:1099-1110)"""
    w1, w2 = _u64(w1).reshape(-1, 4), _u64(w2).reshape(-1, 4)
    out = np.empty_like(w1)
    lib().oracle_fold_witness(field, _p(w1), _p(w2), _p(_u64(r)), ctypes.c_size_t(len(w1)), _p(out)); return out


def fold_error(field, e, cross_terms, r):
    """RelaxedPlonkWitness::fold, E part (src/plonk/mod.rs:1118-1131)"""
    e = _u64(e).reshape(-1, 4).copy()
    terms = [_u64(t).reshape(-1, 4) for t in cross_terms]
    ptrs = (ctypes.c_void_p * len(terms))(*[t.ctypes.data for t in terms])
    lib().oracle_fold_error(field, _p(e), ptrs, ctypes.c_size_t(len(terms)), _p(_u64(r)), ctypes.c_size_t(len(e))); return e


def graph_eval(field, code, num_calculations, constants, rotations, columns, challenges, num_rows):
    """GraphEvaluator::evaluate for rows 0..num_rows (src/polynomial/graph_evaluator.rs:361-390).
    `code` etc. in the flattened layout of include/mira_gpu.h; `columns` = list of host arrays:
    (n, 4) uint64 field columns or 1-D uint8/bool selector columns (None = unresolved index)."""
    code = np.ascontiguousarray(code, dtype=np.uint32)
    constants = _u64(constants).reshape(-1, 4)
    rotations = np.ascontiguousarray(rotations, dtype=np.int32)
    challenges = _u64(challenges).reshape(-1, 4)
    keep, ptrs, kinds = [], [], []
    for c in columns:
        if c is None:
            ptrs.append(None); kinds.append(0); continue
        a = np.asarray(c)
        if a.dtype == np.uint64:
            a = np.ascontiguousarray(a).reshape(-1, 4); kinds.append(0)
        else:
            a = np.ascontiguousarray(a, dtype=np.uint8); kinds.append(1)
        keep.append(a); ptrs.append(a.ctypes.data)
    cp = (ctypes.c_void_p * max(1, len(ptrs)))(*ptrs)
    kinds = np.ascontiguousarray(kinds + [0], dtype=np.uint32)
    out = np.zeros((num_rows, 4), dtype=np.uint64)
    rc = lib().oracle_graph_eval(field, _p(code), ctypes.c_size_t(len(code)), ctypes.c_uint32(num_calculations), _p(constants), ctypes.c_uint32(len(constants)),
                                 _p(rotations), ctypes.c_uint32(len(rotations)), cp, _p(kinds), ctypes.c_uint32(len(ptrs)),
                                 _p(challenges), ctypes.c_uint32(len(challenges)), ctypes.c_size_t(num_rows), _p(out))
    if rc != 0:
        raise ValueError("oracle_graph_eval: malformed graph")
    return out


def pow_tree(field, leaves, weights, point_stride=0):
    """tree_reduce with node = left + right * weights[p][height] (src/nifs/protogalaxy/poly/mod.rs:131-166, 263-290).
    leaves (n or points*n, 4), weights (points, levels, 4); returns (points, 4)."""
    leaves = _u64(leaves).reshape(-1, 4)
    weights = _u64(weights)
    points, levels = weights.shape[0], weights.shape[1]
    n = 1 << levels
    out = np.zeros((points, 4), dtype=np.uint64)
    lib().oracle_pow_tree(field, _p(leaves), ctypes.c_size_t(n), ctypes.c_size_t(point_stride), _p(weights), ctypes.c_uint32(levels), ctypes.c_uint32(points), _p(out))
    return out
