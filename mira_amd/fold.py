"""Host-side mirror of the folding step that follows the commits in one NIFS fold
(reference src/plonk/mod.rs): `RelaxedPlonkWitness::fold` (:1097-1134) on device-resident
vectors, and the commitment side of `RelaxedPlonkInstance::fold` (:986-999, 1049-1053).

Field ids: FIELD_FQ = 0 (bn256::Fq, the scalar field of Grumpkin), FIELD_FR = 1 (bn256::Fr).
Vectors are (n, 4) uint64 Montgomery field elements."""
import ctypes

import numpy as np

from . import _lib

FIELD_FQ, FIELD_FR = 0, 1


def _u64(a, w):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, w)


def fold_witness_device(field, d_out, d_w1, d_w2, r, n, lib=None):
    """out[i] = w1[i] + r * w2[i] on device pointers (src/plonk/mod.rs:1099-1110)."""
    lib = lib or _lib.load()
    r = _u64(r, 4)
    lib.check(lib.c.mira_fold_witness_device(field, ctypes.c_void_p(d_out), ctypes.c_void_p(d_w1), ctypes.c_void_p(d_w2),
                                             r.ctypes.data_as(ctypes.c_void_p), n))


def fold_error_device(field, d_e, d_cross_terms, r, n, lib=None):
    """e[i] += sum_k r^(k+1) * cross_terms[k][i], in place (src/plonk/mod.rs:1118-1131)."""
    lib = lib or _lib.load()
    r = _u64(r, 4)
    ptrs = (ctypes.c_void_p * len(d_cross_terms))(*d_cross_terms)
    lib.check(lib.c.mira_fold_error_device(field, ctypes.c_void_p(d_e), ptrs, len(d_cross_terms), r.ctypes.data_as(ctypes.c_void_p), n))


def fold_relaxed_witness_device(field, d_w_out, d_w1, d_w2, n_w, d_e_out, d_e, d_cross_terms, r, n, lib=None):
    """`RelaxedPlonkWitness::fold` in one submission (src/plonk/mod.rs:1097-1134): w_out = w1 + r w2 over n_w elements and
    e_out = e + sum_k r^(k+1) cross_terms[k] over n (d_e_out may be d_e)."""
    lib = lib or _lib.load()
    r = _u64(r, 4)
    ptrs = (ctypes.c_void_p * max(1, len(d_cross_terms)))(*d_cross_terms)
    lib.check(lib.c.mira_fold_relaxed_witness_device(field, ctypes.c_void_p(d_w_out), ctypes.c_void_p(d_w1), ctypes.c_void_p(d_w2), n_w,
                                                     ctypes.c_void_p(d_e_out), ctypes.c_void_p(d_e), ptrs, len(d_cross_terms),
                                                     r.ctypes.data_as(ctypes.c_void_p), n))


def fold_witness(field, w1, w2, r, lib=None):
    """Host-array convenience around fold_witness_device (uploads, folds, downloads)."""
    lib = lib or _lib.load()
    w1, w2 = _u64(w1, 4), _u64(w2, 4)
    assert len(w1) == len(w2), "zip_eq: witness vectors differ in length"
    n = len(w1)
    if n == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    d1, d2 = lib.alloc(n * 32), lib.alloc(n * 32)
    lib.upload(d1, w1); lib.upload(d2, w2)
    fold_witness_device(field, d1, d1, d2, r, n, lib)
    out = lib.download(d1, (n, 4))
    lib.free(d1); lib.free(d2)
    return out


def fold_error(field, e, cross_terms, r, lib=None):
    lib = lib or _lib.load()
    e = _u64(e, 4)
    n = len(e)
    terms = [_u64(t, 4) for t in cross_terms]
    assert all(len(t) == n for t in terms), "cross terms differ in length from E"
    if n == 0:
        return e.copy()
    de = lib.alloc(n * 32); lib.upload(de, e)
    dts = []
    for t in terms:
        p = lib.alloc(n * 32); lib.upload(p, t); dts.append(p)
    fold_error_device(field, de, dts, r, n, lib)
    out = lib.download(de, (n, 4))
    for p in [de] + dts:
        lib.free(p)
    return out


def g1_mul_add(curve, acc, scalar, point, lib=None):
    """acc + scalar * point (affine): `*W1 + best_multiexp(&[*r], &[W2])` of
    RelaxedPlonkInstance::fold (src/plonk/mod.rs:992-993)."""
    lib = lib or _lib.load()
    acc, scalar, point = _u64(acc, 8), _u64(scalar, 4), _u64(point, 8)
    out = np.empty(8, dtype=np.uint64)
    lib.check(lib.c.mira_g1_mul_add(curve, acc.ctypes.data_as(ctypes.c_void_p), scalar.ctypes.data_as(ctypes.c_void_p),
                                    point.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def g1_lincomb(curve, acc, scalars, points, lib=None):
    """acc + sum_i scalars[i] * points[i] (affine): `E_commit + sum_k r^(k+1) T_k` of
    RelaxedPlonkInstance::fold (src/plonk/mod.rs:1049-1053) with one shared chain of doublings."""
    lib = lib or _lib.load()
    acc, scalars, points = _u64(acc, 8), _u64(scalars, 4), _u64(points, 8)
    assert len(scalars) == len(points)
    out = np.empty(8, dtype=np.uint64)
    lib.check(lib.c.mira_g1_lincomb(curve, acc.ctypes.data_as(ctypes.c_void_p), scalars.ctypes.data_as(ctypes.c_void_p),
                                    points.ctypes.data_as(ctypes.c_void_p), len(scalars), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def fold_instance_commitments(curve, w1_commits, w2_commits, r, e_commit, cross_term_commits, lib=None):
    """The commitments of `RelaxedPlonkInstance::fold` in one call (src/plonk/mod.rs:986-999, 1049-1053):
    returns ([W1_i + r W2_i], E + sum_k r^(k+1) T_k) as affine points.  `r` in Montgomery form, like g1_mul_add's scalar."""
    lib = lib or _lib.load()
    w1, w2 = _u64(w1_commits, 8).reshape(-1, 8), _u64(w2_commits, 8).reshape(-1, 8)
    assert len(w1) == len(w2), "zip: the instances differ in their number of W commitments"
    t = _u64(cross_term_commits, 8).reshape(-1, 8)
    r, e = _u64(r, 4), _u64(e_commit, 8)
    w_out, e_out = np.empty((len(w1), 8), dtype=np.uint64), np.empty(8, dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.check(lib.c.mira_g1_fold_commitments(curve, vp(r), vp(w1), vp(w2), len(w1), vp(e), vp(t), len(t), vp(w_out), vp(e_out)))
    return w_out, e_out
