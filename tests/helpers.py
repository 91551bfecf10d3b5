"""Shared test helpers: conversions between Python ints and the (n,4)/(n,8) uint64 Montgomery
arrays of the C ABI, and reconstruction of the golden MSM inputs."""
import json
import os

import numpy as np

from oracle import cref as C
from oracle import pyref as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def base_mod(cid):
    return P.CURVES[cid].p


def scalar_mod(cid):
    return P.CURVES[cid].r


def scalar_field_id(cid):      # oracle field ids: 0 = Fq, 1 = Fr
    return C.FIELD_FR if cid == P.CURVE_BN256 else C.FIELD_FQ


def base_field_id(cid):
    return C.FIELD_FQ if cid == P.CURVE_BN256 else C.FIELD_FR


def ints_to_mont(vals, mod):
    """canonical ints -> (n,4) uint64 Montgomery"""
    return np.array([P.limbs4(P.to_mont(v, mod)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def mont_to_ints(arr, mod):
    return [P.from_mont(P.from_limbs4(row), mod) for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


def point_to_arr(pt, cid):
    """affine tuple / None -> (8,) uint64 Montgomery, identity = zeros"""
    if pt is None:
        return np.zeros(8, dtype=np.uint64)
    m = base_mod(cid)
    return np.array(P.limbs4(P.to_mont(pt[0], m)) + P.limbs4(P.to_mont(pt[1], m)), dtype=np.uint64)


def arr_to_point(arr, cid):
    arr = np.asarray(arr, dtype=np.uint64).reshape(8)
    if not arr.any():
        return None
    m = base_mod(cid)
    return (P.from_mont(P.from_limbs4(arr[:4]), m), P.from_mont(P.from_limbs4(arr[4:]), m))


def neg_point_arr(arr, cid):
    pt = arr_to_point(arr, cid)
    return point_to_arr(P.ec_neg(pt, P.CURVES[cid]), cid)


def golden_msm_case(cid, case):
    """Rebuild (scalars, bases, expected) arrays of one tests/golden/msm_vectors.json case."""
    n = case["n"]
    sm = scalar_mod(cid)
    sc = C.synth_scalars(cid, n, seed=case["scalar_seed"])
    bs = C.synth_bases(cid, n, seed=case["base_seed"])
    if case["edits"]:
        for idx, hv in case["edits"]["scalars"].items():
            sc[int(idx)] = ints_to_mont([int(hv, 16)], sm)[0]
        bs[6] = bs[5]
        bs[7] = 0
        bs[9] = neg_point_arr(bs[8], cid)
    exp = case["result"]
    expected = point_to_arr(None if exp is None else (int(exp[0], 16), int(exp[1], 16)), cid)
    return sc, bs, expected
