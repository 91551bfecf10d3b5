"""cross-term evaluation.  CPU suite: the graph builder against the
reference's rules, the two oracles against each other, and the kernel under the test-only
emulation.  Cases follow the reference's own tests (src/polynomial/graph_evaluator.rs:446-640)."""
import random

import numpy as np
import pytest

from graph_cases import MODS, gate_like_expression, mock_data, oracle_columns, random_expression
from helpers import ints_to_mont, mont_to_ints
from mira_amd import _lib, commitment as cm
from harness import graph_evaluator as G
from oracle import cref as C
from oracle import pyref as P


def direct(expr, ints, num_rows, mod):
    return [P.eval_expression(expr.to_tuple(), ints, r, num_rows, mod) for r in range(num_rows)]


# ---- graph construction (host logic) ----------------------------------------------------------
def test_default_constants_and_final_store():
    ge = G.GraphEvaluator.new(G.Constant(77))
    assert ge.constants == [0, 1, 2, 77]                       # graph_evaluator.rs:186-187
    assert ge.calculations == [(G.OP_STORE, (G.SRC_CONSTANT, 3))]


def test_simplifications():
    a, b = G.Polynomial(0), G.Polynomial(1)
    sa, sb = (G.SRC_INTERMEDIATE, 0), (G.SRC_INTERMEDIATE, 1)
    calcs = lambda e: G.GraphEvaluator.new(e).calculations
    assert calcs(G.Product(a, G.Constant(0))) == [(G.OP_STORE, (G.SRC_COLUMN, 0, 0)), (G.OP_STORE, (G.SRC_CONSTANT, 0))]
    assert calcs(G.Product(G.Constant(1), a))[-1] == (G.OP_STORE, sa)
    assert calcs(G.Product(a, G.Constant(2)))[1] == (G.OP_DOUBLE, sa)
    assert calcs(G.Product(a, a))[1] == (G.OP_SQUARE, sa)
    assert calcs(G.Product(b, a))[2] == (G.OP_MUL, sa, sb) or calcs(G.Product(b, a))[2] == (G.OP_MUL, sb, sa)
    assert calcs(G.Sum(a, G.Negated(b)))[2] == (G.OP_SUB, sa, sb)
    assert calcs(G.Sum(G.Constant(0), G.Negated(a)))[1] == (G.OP_NEGATE, sa)
    assert calcs(G.Sum(a, G.Negated(G.Constant(0))))[-1] == (G.OP_STORE, sa)
    assert calcs(G.Scaled(a, 0)) == [(G.OP_STORE, (G.SRC_CONSTANT, 0))]
    assert calcs(G.Scaled(a, 1))[-1] == (G.OP_STORE, sa)
    ge = G.GraphEvaluator.new(G.Scaled(a, 9))
    assert ge.constants[3] == 9 and ge.calculations[1] == (G.OP_MUL, sa, (G.SRC_CONSTANT, 3))
    assert G.GraphEvaluator.new(G.Negated(G.Constant(5)), G.FIELD_FR).constants[3] == P.R_MOD - 5


def test_operand_order_and_sharing():
    a, b = G.Polynomial(3, 1), G.Polynomial(4, -1)
    e = G.Sum(G.Product(b, a), G.Product(b, a))                # the same product twice -> one calculation
    ge = G.GraphEvaluator.new(e)
    assert ge.rotations == [-1, 1]                             # in order of first use (b is visited first)
    muls = [c for c in ge.calculations if c[0] == G.OP_MUL]
    assert len(muls) == 1 and muls[0][1] <= muls[0][2]         # smaller source first, :311-321
    assert ge.calculations[-2][0] == G.OP_ADD and ge.calculations[-2][1] == ge.calculations[-2][2]


# ---- the reference's evaluation tests, on the emulated kernel ----------------------------------
@pytest.mark.parametrize("field", [0, 1])
def test_reference_cases(emu_lib, field):
    mod = MODS[field]
    rnd = random.Random(11 + field)
    lhs, rhs = rnd.getrandbits(256) % mod, rnd.getrandbits(256) % mod
    ev = lambda e, data=None: mont_to_ints(G.GraphEvaluator.new(e, field).evaluate(data or {}, lib=emu_lib), mod)
    assert ev(G.Constant(lhs)) == [lhs]                                              # constant
    assert ev(G.Sum(G.Constant(lhs), G.Constant(rhs))) == [(lhs + rhs) % mod]        # sum_const
    assert ev(G.Product(G.Constant(lhs), G.Constant(rhs))) == [lhs * rhs % mod]      # product_const
    assert ev(G.Negated(G.Constant(lhs))) == [(-lhs) % mod]                          # neg_const
    assert ev(G.Challenge(0), dict(challenges=[lhs])) == [lhs]                       # challenge
    # poly: 2 selectors, 2 fixed, 2 advice columns of 2 rows; rotations wrap (rem_euclid)
    ints, arrs = mock_data(field, 2, 2, 2, 2, 0, seed=5)
    for col in range(6):
        for rot in (0, 1, -1, 2, -3):
            got = ev(G.Polynomial(col, rot), arrs)
            want = direct(G.Polynomial(col, rot), ints, 2, mod)
            assert got == want
    sel = [int(v) for v in ints["selectors"][0]]
    assert ev(G.Polynomial(0, 0), arrs) == sel                                       # bool -> ONE / ZERO
    assert ev(G.Polynomial(4, 1), arrs) == [ints["advice"][0][1], ints["advice"][0][0]]


@pytest.mark.parametrize("field,seed", [(0, 1), (1, 2), (1, 3)])
def test_random_trees_three_ways(emu_lib, field, seed):
    """device VM (emulated) == C restatement of Calculation::evaluate == direct tree evaluation"""
    mod, n = MODS[field], 37
    rng = random.Random(seed)
    ints, arrs = mock_data(field, n, 2, 3, 5, 3, seed=100 + seed)
    for _ in range(6):
        e = random_expression(rng, 6, 10, 3)
        ge = G.GraphEvaluator.new(e, field)
        code, consts, rots = ge.flatten()
        want = direct(e, ints, n, mod)
        assert mont_to_ints(C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), ints_to_mont(ints["challenges"], mod), n), mod) == want
        assert mont_to_ints(ge.evaluate(arrs, lib=emu_lib), mod) == want


def test_batch_equals_one_by_one(emu_lib):
    """mira_graph_eval_batch: the cross-term graphs of a fold step over the same columns in one
    submission (18 graphs: two launches), an empty batch, a graph that reads a missing column."""
    field, n = 1, 29
    mod = MODS[field]
    rng = random.Random(77)
    ints, arrs = mock_data(field, n, 1, 2, 4, 2, seed=177)
    ptrs, cols = [], []
    for s_ in arrs["selectors"]:
        p = emu_lib.alloc(s_.nbytes); emu_lib.upload(p, s_); ptrs.append(p); cols.append((p, G.COL_BOOL))
    for f in list(arrs["fixed"]) + list(arrs["advice"]):
        p = emu_lib.alloc(f.nbytes); emu_lib.upload(p, f); ptrs.append(p); cols.append((p, G.COL_FIELD))
    evs = [G.GraphEvaluator.new(gate_like_expression(rng, rng.choice([1, 3, 8]), 5, 7, 2) if k % 2 else random_expression(rng, 5, 7, 2), field) for k in range(18)]
    want = [mont_to_ints(ev.evaluate(arrs, lib=emu_lib), mod) for ev in evs]
    d = emu_lib.alloc(len(evs) * n * 32)
    G.GraphEvaluator.evaluate_batch_device(evs, cols, ints["challenges"], n, [d + k * n * 32 for k in range(len(evs))], lib=emu_lib)
    got = emu_lib.download(d, (len(evs), n, 4))
    assert [mont_to_ints(got[k], mod) for k in range(len(evs))] == want
    G.GraphEvaluator.evaluate_batch_device([], cols, ints["challenges"], n, [], lib=emu_lib)
    with pytest.raises(_lib.MiraError):
        G.GraphEvaluator.evaluate_batch_device(evs[:2], cols[:2] + [None] * (len(cols) - 2), ints["challenges"], n, [d, d + n * 32], lib=emu_lib)
    for p in ptrs + [d]:
        emu_lib.free(p)


def flat_graph(calcs):
    words = []
    for op, srcs in calcs:
        words.append(op | ((len(srcs) - 2) << 8 if op == G.OP_HORNER else 0))
        words += [(k << 29) | p for k, p in srcs]
    return np.array(words, dtype=np.uint32)


def run_flat(lib, field, code, ncalc, consts, rots, cols, chal, n):
    g = _lib.MiraGraph(code.ctypes.data, len(code), ncalc, len(consts), consts.ctypes.data, rots.ctypes.data, len(rots), 0)
    import ctypes
    carr = (_lib.MiraEvalColumn * max(1, len(cols)))()
    ptrs = []
    for k, c in enumerate(cols):
        if c is None:
            continue
        a = np.ascontiguousarray(c)
        p = lib.alloc(a.nbytes); lib.upload(p, a); ptrs.append(p)
        carr[k].d_data, carr[k].kind = p, (G.COL_BOOL if a.dtype == np.uint8 else G.COL_FIELD)
    out = lib.alloc(max(1, n) * 32)
    try:
        lib.check(lib.c.mira_graph_eval_device(field, ctypes.byref(g), carr, len(cols), chal.ctypes.data_as(ctypes.c_void_p), len(chal), n, ctypes.c_void_p(out)))
        return lib.download(out, (n, 4))
    finally:
        for p in ptrs + [out]:
            lib.free(p)


def test_horner_and_empty_graph(emu_lib):
    """Calculation::Horner (graph_evaluator.rs:148-155) is part of the instruction set although
    add_expression never emits it; an empty graph evaluates to zero (:386-389)."""
    field, mod, n = 1, P.R_MOD, 9
    ints, arrs = mock_data(field, n, 0, 1, 3, 1, seed=8)
    col = lambda c, r=0: (G.SRC_COLUMN, c | (r << 20))
    calcs = [(G.OP_STORE, [col(1)]), (G.OP_HORNER, [(G.SRC_INTERMEDIATE, 0), (G.SRC_CHALLENGE, 0), col(2), col(3, 1), (G.SRC_CONSTANT, 2)])]
    code, consts, rots = flat_graph(calcs), ints_to_mont([0, 1, 2], mod), np.array([0, 1], dtype=np.int32)
    chal = ints_to_mont(ints["challenges"], mod)
    cols = oracle_columns(arrs)
    want = []
    for r in range(n):
        v, f = ints["advice"][0][r], ints["challenges"][0]
        for part in (ints["advice"][1][r], ints["advice"][2][(r + 1) % n], 2):
            v = (v * f + part) % mod
        want.append(v)
    assert mont_to_ints(C.graph_eval(field, code, 2, consts, rots, cols, chal, n), mod) == want
    assert mont_to_ints(run_flat(emu_lib, field, code, 2, consts, rots, cols, chal, n), mod) == want
    empty = np.zeros(0, dtype=np.uint32)
    assert not run_flat(emu_lib, field, empty, 0, consts, rots, cols, chal, n).any()
    assert not C.graph_eval(field, empty, 0, consts, rots, cols, chal, n).any()


def test_errors(emu_lib):
    """EvalError::{ChallengeIndexOutOfBoundary, ColumnVariableIndexOutOfBoundary} (src/plonk/eval.rs:3-24) and malformed code"""
    ints, arrs = mock_data(1, 4, 1, 1, 1, 1, seed=9)
    with pytest.raises(_lib.MiraError, match="challenge index out of boundary: 1"):
        G.GraphEvaluator.new(G.Challenge(1)).evaluate(arrs, lib=emu_lib)
    with pytest.raises(_lib.MiraError, match="column variable index out of boundary: 3"):
        G.GraphEvaluator.new(G.Polynomial(3)).evaluate(arrs, lib=emu_lib)
    consts, rots, chal = ints_to_mont([0, 1, 2], P.R_MOD), np.zeros(1, dtype=np.int32), ints_to_mont([1], P.R_MOD)
    bad_forward = flat_graph([(G.OP_STORE, [(G.SRC_INTERMEDIATE, 0)])])
    with pytest.raises(_lib.MiraError, match="before it is written"):
        run_flat(emu_lib, 1, bad_forward, 1, consts, rots, [], chal, 4)
    with pytest.raises(_lib.MiraError, match="trailing"):
        run_flat(emu_lib, 1, np.append(flat_graph([(G.OP_STORE, [(G.SRC_CONSTANT, 1)])]), np.uint32(0)), 1, consts, rots, [], chal, 4)
    with pytest.raises(_lib.MiraError, match="unknown calculation"):
        run_flat(emu_lib, 1, np.array([9, 0], dtype=np.uint32), 1, consts, rots, [], chal, 4)


# ---- PlonkEvalDomain and the fused evaluate + commit ------------------------------------------
@pytest.mark.parametrize("num_w", [2, 3, 1])
def test_plonk_domain_index_map(num_w):
    num_advice, num_lookup, rows = 3, 2, 8
    base = 0x10000000
    lens = {2: [(num_advice + 3 * num_lookup) * rows, 2 * num_lookup * rows], 3: [num_advice * rows, 3 * num_lookup * rows, 2 * num_lookup * rows], 1: [num_advice * rows]}[num_w]
    w1 = [(base + 0x100000 * i, l) for i, l in enumerate(lens)]
    w2 = [(base * 2 + 0x100000 * i, l) for i, l in enumerate(lens)]
    dom = G.PlonkEvalDomain(num_advice, num_lookup, [], [1, 2], [3], w1, w2, rows)
    cols = dom.columns()
    assert cols[:3] == [(1, G.COL_BOOL), (2, G.COL_BOOL), (3, G.COL_FIELD)]
    width = num_advice + 5 * num_lookup
    assert len(cols) == 3 + 2 * width
    for idx in range(2 * width):
        loc = P.plonk_advice_location(num_advice, num_lookup, num_w, num_w, idx)
        got = cols[3 + idx]
        if loc is None:
            assert got is None
            continue
        first, i, j = loc
        ws = w1 if first else w2
        if i >= len(ws) or ws[i][1] < (j + 1) * rows:
            assert got is None
        else:
            assert got == (ws[i][0] + j * rows * 32, G.COL_FIELD)


def test_commit_cross_terms_emulated(emu_lib):
    """evaluation + commit spans of commit_cross_terms (src/nifs/vanilla/mod.rs:100-127) on one key"""
    field, cid, mod, rows, num_advice = 1, 0, P.R_MOD, 64, 4
    rng = random.Random(21)
    ints, arrs = mock_data(field, rows, 1, 2, 2 * num_advice, 2, seed=77)
    lib = emu_lib
    up = lambda a: (lambda p: (lib.upload(p, a), p)[1])(lib.alloc(a.nbytes))
    d_sel, d_fix = [up(s) for s in arrs["selectors"]], [up(f) for f in arrs["fixed"]]
    W1 = np.concatenate(arrs["advice"][:num_advice]); W2 = np.concatenate(arrs["advice"][num_advice:])
    d_w1, d_w2 = up(W1), up(W2)
    dom = G.PlonkEvalDomain(num_advice, 0, ints["challenges"], d_sel, d_fix, [(d_w1, len(W1))], [(d_w2, len(W2))], rows)
    exprs = [random_expression(rng, 5, 3 + 2 * num_advice, 2), None, random_expression(rng, 5, 3 + 2 * num_advice, 2)]
    evs = [None if e is None else G.GraphEvaluator.new(e, field) for e in exprs]
    key = cm.CommitmentKey(cid, C.synth_bases(cid, rows, seed=3), lib=lib)
    d_terms, commits = G.commit_cross_terms(key, evs, dom, lib=lib)
    terms = lib.download(d_terms, (3, rows, 4))
    for k, e in enumerate(exprs):
        want = [0] * rows if e is None else direct(e, ints, rows, mod)
        assert mont_to_ints(terms[k], mod) == want
        assert (commits[k] == C.msm_pippenger(cid, ints_to_mont(want, mod), C.synth_bases(cid, rows, seed=3))).all()
    for p in d_sel + d_fix + [d_w1, d_w2, d_terms]:
        lib.free(p)
    key.close()
