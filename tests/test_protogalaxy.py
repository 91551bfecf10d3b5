"""ProtoGalaxy's polynomial pipeline (compute_F / compute_G / compute_K,
reference src/nifs/protogalaxy/poly/mod.rs) -- the device path under the test-only emulation
against the Python-integer restatement, plus the properties the reference's own tests assert
(:417-500: F and G vanish on satisfied traces, not otherwise)."""
import ctypes
import random

import numpy as np
import pytest

from helpers import ints_to_mont, load_golden, mont_to_ints
from mira_amd import _lib
from harness import graph_evaluator as G
from harness import protogalaxy as PG
from oracle import cref as C
from oracle import pyref as P

MOD = P.R_MOD


def build_case(lib, k, seed, satisfied, num_traces=2):
    """A small circuit: 1 selector, 1 fixed column, 3 advice columns, 2 gates
         gate 0:  s * (a * b - c)                (degree 2 in the witness)
         gate 1:  s * (a + f - b') * ch0         (rotation, fixed column, challenge)
    Satisfied traces set c = a * b and b' = a + f on selected rows."""
    rng = random.Random(seed)
    rows = 1 << k
    sel = [rng.random() < 0.7 for _ in range(rows)]
    fix = [rng.getrandbits(250) % MOD for _ in range(rows)]
    s, f, a, b, c = G.Polynomial(0), G.Polynomial(1), G.Polynomial(2), G.Polynomial(3), G.Polynomial(4)
    gates = [G.Product(s, G.Sum(G.Product(a, b), G.Negated(c))),
             G.Product(G.Product(s, G.Sum(G.Sum(a, f), G.Negated(G.Polynomial(3, 1)))), G.Challenge(0))]

    def trace():
        av = [rng.getrandbits(250) % MOD for _ in range(rows)]
        bv = [0] * rows
        if satisfied:                                           # b[r + 1] = a[r] + f[r] all the way round is
            bv[0] = rng.getrandbits(250) % MOD                  # over-determined: make it hold on selected rows
            for r in range(rows):
                nxt = (av[r] + fix[r]) % MOD
                if r + 1 < rows:
                    bv[r + 1] = nxt if sel[r] else rng.getrandbits(250) % MOD
                elif sel[r] and nxt != bv[0]:
                    sel[r] = False                              # drop the wrap-around constraint
        else:
            bv = [rng.getrandbits(250) % MOD for _ in range(rows)]
        cv = [(x * y) % MOD if satisfied else rng.getrandbits(250) % MOD for x, y in zip(av, bv)]
        return dict(challenges=[rng.getrandbits(250) % MOD], W=[av + bv + cv])
    traces = [trace() for _ in range(num_traces + 1)]
    structure_ints = dict(k=k, gates=[g.to_tuple() for g in gates], selectors=[sel], fixed=[fix], num_advice=3)
    ptrs = []

    def up(arr):
        p = lib.alloc(max(1, arr.nbytes)); lib.upload(p, arr); ptrs.append(p); return p
    S = PG.Structure(k, gates, [up(np.array(sel, dtype=np.uint8))], [up(ints_to_mont(fix, MOD))], 3)
    dev = [PG.Trace(t["challenges"], [(up(ints_to_mont(t["W"][0], MOD)), len(t["W"][0]))]) for t in traces]
    return structure_ints, traces, S, dev, ptrs


def test_lagrange_helpers_match_oracle(emu_lib):
    for log_n in (1, 2, 3):
        assert PG.iter_cyclic_subgroup(log_n, emu_lib) == P.pg_cyclic_subgroup(log_n)
        for X in (P.pg_cyclic_subgroup(log_n)[1], 12345, P.FR_ZETA):
            assert PG.eval_lagrange_poly_for_cyclic_group(X, log_n, emu_lib) == P.pg_lagrange(X, log_n)
            assert PG.eval_vanish_polynomial(log_n, X) == P.pg_vanish(log_n, X)
    # correctness_for_cyclic_element (lagrange.rs:92-112): on a domain element exactly one polynomial is 1
    w = P.pg_cyclic_subgroup(2)
    assert PG.eval_lagrange_poly_for_cyclic_group(w[3], 2, emu_lib) == [0, 0, 0, 1]
    # basic_lagrange_test (lagrange.rs:115-127): the reference's four constants
    kat = load_golden("ref_kats.json")["basic_lagrange_test"]
    assert PG.eval_lagrange_poly_for_cyclic_group(kat["X"], kat["log_n"], emu_lib) == [int(v) for v in kat["output_decimal"]]


def test_tree_reduce_and_lincomb_kernels(emu_lib):
    """mira_pow_tree_reduce_device / mira_lincomb_device against their definitions, both fields,
    shared and per-point leaves, one to three kernel rounds"""
    rng = random.Random(4)
    for field, mod in ((1, P.R_MOD), (0, P.P_MOD)):
        for levels, points, shared in ((0, 2, True), (3, 3, True), (9, 2, False), (12, 2, True), (13, 1, False)):
            n = 1 << levels
            leaves = [[rng.getrandbits(256) % mod for _ in range(n)] for _ in range(1 if shared else points)]
            w = [[rng.getrandbits(256) % mod for _ in range(levels)] for _ in range(points)]
            flat = ints_to_mont([v for row in leaves for v in row], mod)
            d = emu_lib.alloc(flat.nbytes); emu_lib.upload(d, flat)
            out = np.zeros((points, 4), dtype=np.uint64)
            wm = ints_to_mont([x for row in w for x in row] or [0], mod)
            emu_lib.check(emu_lib.c.mira_pow_tree_reduce_device(field, ctypes.c_void_p(d), n, 0 if shared else n, wm.ctypes.data_as(ctypes.c_void_p), points,
                                                                  out.ctypes.data_as(ctypes.c_void_p)))
            emu_lib.free(d)
            for p in range(points):
                lv = leaves[0 if shared else p]
                want = 0
                for i, v in enumerate(lv):
                    wt = 1
                    for j in range(levels):
                        if (i >> j) & 1:
                            wt = wt * w[p][j] % mod
                    want = (want + v * wt) % mod
                assert mont_to_ints(out[p:p + 1], mod) == [want], (field, levels, p)
        vecs = [[rng.getrandbits(256) % mod for _ in range(77)] for _ in range(3)]
        co = [rng.getrandbits(256) % mod for _ in range(3)]
        ptrs = []
        for v in vecs:
            a = ints_to_mont(v, mod); p_ = emu_lib.alloc(a.nbytes); emu_lib.upload(p_, a); ptrs.append(p_)
        dout = emu_lib.alloc(77 * 32)
        emu_lib.check(emu_lib.c.mira_lincomb_device(field, ctypes.c_void_p(dout), (ctypes.c_void_p * 3)(*ptrs), ints_to_mont(co, mod).ctypes.data_as(ctypes.c_void_p), 3, 77))
        got = mont_to_ints(emu_lib.download(dout, (77, 4)), mod)
        assert got == [sum(c * v[i] for c, v in zip(co, vecs)) % mod for i in range(77)]
        for p_ in ptrs + [dout]:
            emu_lib.free(p_)
    with pytest.raises(_lib.MiraError, match="power-of-two"):
        d = emu_lib.alloc(96)
        emu_lib.c.mira_pow_tree_reduce_device.restype = ctypes.c_int
        emu_lib.check(emu_lib.c.mira_pow_tree_reduce_device(1, ctypes.c_void_p(d), 3, 0, out.ctypes.data_as(ctypes.c_void_p), 1, out.ctypes.data_as(ctypes.c_void_p)))


@pytest.mark.parametrize("k,satisfied", [(2, True), (3, False), (4, True)])
def test_compute_F(emu_lib, k, satisfied):
    ints, traces, S, dev, ptrs = build_case(emu_lib, k, seed=10 + k, satisfied=satisfied)
    rng = random.Random(k)
    betas = [rng.getrandbits(250) % MOD for _ in range(12)]
    delta = rng.getrandbits(250) % MOD
    got = PG.compute_F(betas, delta, S, dev[0], lib=emu_lib)
    assert got == P.pg_compute_F(betas, delta, ints, traces[0])
    assert (not any(got)) == satisfied                           # zero_f / non_zero_f (:417-455)
    for p in ptrs:
        emu_lib.free(p)


@pytest.mark.parametrize("k,satisfied,num_traces", [(2, True, 1), (3, False, 2), (3, True, 3)])
def test_compute_G_and_K(emu_lib, k, satisfied, num_traces):
    ints, traces, S, dev, ptrs = build_case(emu_lib, k, seed=20 + k, satisfied=satisfied, num_traces=num_traces)
    rng = random.Random(100 + k)
    betas = [rng.getrandbits(250) % MOD for _ in range(12)]
    alpha, delta = rng.getrandbits(250) % MOD, rng.getrandbits(250) % MOD
    bs = PG.beta_stroke(betas, alpha, delta)
    md = S.max_degree()
    assert md == 2                                               # a * b, and (a + f - b') * ch0; selectors and fixed columns count 0
    got = PG.compute_G(S, bs, dev[0], dev[1:], lib=emu_lib)
    want = P.pg_compute_G(ints, bs, traces[0], traces[1:], md)
    assert got == want
    assert len(got) == 1 << (num_traces * md).bit_length()
    if satisfied:
        # G interpolates the traces' own sums over the Lagrange domain: zero there for satisfied traces
        log_dom = (num_traces).bit_length()
        for X in P.pg_cyclic_subgroup(log_dom)[: num_traces + 1]:
            assert sum(c * pow(X, i, MOD) for i, c in enumerate(got)) % MOD == 0
        # zero_g (:457-478): accumulator and trace are one and the same satisfied trace -> G == 0
        assert not any(PG.compute_G(S, bs, dev[0], [dev[0]], lib=emu_lib))
    else:
        assert any(got)                                          # non_zero_g (:480-500)
    f_alpha = rng.getrandbits(250) % MOD
    assert PG.compute_K(S, f_alpha, bs, dev[0], dev[1:], lib=emu_lib) == P.pg_compute_K(ints, f_alpha, bs, traces[0], traces[1:], md)
    with pytest.raises(ValueError, match="0 traces"):
        PG.compute_G(S, bs, dev[0], [], lib=emu_lib)
    for p in ptrs:
        emu_lib.free(p)
