"""GPU parity of ProtoGalaxy's polynomial pipeline through the C ABI."""
import ctypes
import random

import numpy as np
import pytest

from helpers import ints_to_mont, mont_to_ints
from harness import protogalaxy as PG
from oracle import cref as C
from oracle import pyref as P
from test_protogalaxy import MOD, build_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,satisfied,num_traces", [(6, True, 1), (9, False, 2), (10, True, 3)])
def test_compute_F_G_K_vs_oracle(gpu_lib, k, satisfied, num_traces):
    ints, traces, S, dev, ptrs = build_case(gpu_lib, k, seed=40 + k, satisfied=satisfied, num_traces=num_traces)
    rng = random.Random(7 * k)
    betas = [rng.getrandbits(250) % MOD for _ in range(16)]
    alpha, delta, f_alpha = (rng.getrandbits(250) % MOD for _ in range(3))
    f = PG.compute_F(betas, delta, S, dev[0], lib=gpu_lib)
    assert f == P.pg_compute_F(betas, delta, ints, traces[0])
    assert (not any(f)) == satisfied
    bs = PG.beta_stroke(betas, alpha, delta)
    g = PG.compute_G(S, bs, dev[0], dev[1:], lib=gpu_lib)
    assert g == P.pg_compute_G(ints, bs, traces[0], traces[1:], S.max_degree())
    assert PG.compute_K(S, f_alpha, bs, dev[0], dev[1:], lib=gpu_lib) == P.pg_compute_K(ints, f_alpha, bs, traces[0], traces[1:], S.max_degree())
    for p in ptrs:
        gpu_lib.free(p)


def test_tree_reduce_2p22_properties(gpu_lib):
    """2^22 leaves (two kernel rounds), 16 challenges: all weights one -> plain sums; a single
    non-zero leaf -> that leaf times the product of the weights its index selects."""
    levels, points = 22, 16
    n = 1 << levels
    leaves = C.synth_scalars(0, n, seed=123)
    d = gpu_lib.alloc(leaves.nbytes); gpu_lib.upload(d, leaves)
    out = np.zeros((points, 4), dtype=np.uint64)
    one = C.to_mont(C.FIELD_FR, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    w = np.repeat(one, points * levels, axis=0)
    gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), n, 0, w.ctypes.data_as(ctypes.c_void_p), points, out.ctypes.data_as(ctypes.c_void_p)))
    assert (out == out[0]).all()
    # cross-check the sum through linearity: sum(leaves) = sum(first half) + sum(second half)
    halves = np.zeros((2, 4), dtype=np.uint64)
    for h in range(2):
        gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d + h * (n // 2) * 32), n // 2, 0, w.ctypes.data_as(ctypes.c_void_p), 1,
                                                              halves[h].ctypes.data_as(ctypes.c_void_p)))
    assert (sum(mont_to_ints(halves, MOD)) % MOD) == mont_to_ints(out[:1], MOD)[0]
    sample = 1 << 16
    small = np.zeros((1, 4), dtype=np.uint64)
    gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), sample, 0, w.ctypes.data_as(ctypes.c_void_p), 1, small.ctypes.data_as(ctypes.c_void_p)))
    assert mont_to_ints(small, MOD) == [sum(mont_to_ints(leaves[:sample], MOD)) % MOD]
    # the C restatement of the tree on all 2^22 leaves, random weights, shared and per-point leaves
    rng = random.Random(9)
    wr = C.synth_scalars(0, points * levels, seed=77).reshape(points, levels, 4)
    gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), n, 0, wr.ctypes.data_as(ctypes.c_void_p), points, out.ctypes.data_as(ctypes.c_void_p)))
    assert (out == C.pow_tree(1, leaves, wr)).all()
    q = n // 4                                                  # four points, each with its own quarter as leaves
    gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), q, q, wr[:4, : levels - 2].copy().ctypes.data_as(ctypes.c_void_p), 4, out.ctypes.data_as(ctypes.c_void_p)))
    assert (out[:4] == C.pow_tree(1, leaves, wr[:4, : levels - 2].copy(), point_stride=q)).all()
    # single non-zero leaf
    wi = [[rng.getrandbits(250) % MOD for _ in range(levels)] for _ in range(points)]
    wm = ints_to_mont([x for row in wi for x in row], MOD)
    idx = 0b1011001110001111010101
    z = np.zeros((n, 4), dtype=np.uint64); z[idx] = leaves[idx]
    gpu_lib.upload(d, z)
    gpu_lib.check(gpu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), n, 0, wm.ctypes.data_as(ctypes.c_void_p), points, out.ctypes.data_as(ctypes.c_void_p)))
    leaf = mont_to_ints(leaves[idx:idx + 1], MOD)[0]
    for p in range(points):
        want = leaf
        for j in range(levels):
            if (idx >> j) & 1:
                want = want * wi[p][j] % MOD
        assert mont_to_ints(out[p:p + 1], MOD) == [want]
    gpu_lib.free(d)


def test_basic_lagrange_kat(gpu_lib):
    """basic_lagrange_test (src/polynomial/lagrange.rs:115-127): the reference's four constants
    through the GPU library's get_omega_or_inv."""
    from helpers import load_golden
    kat = load_golden("ref_kats.json")["basic_lagrange_test"]
    assert PG.eval_lagrange_poly_for_cyclic_group(kat["X"], kat["log_n"], gpu_lib) == [int(v) for v in kat["output_decimal"]]
