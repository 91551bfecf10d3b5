"""GPU parity of the cross-term evaluator through the C ABI: random
expression graphs over 2^12..2^17 rows against the C restatement of Calculation::evaluate, and the
fused evaluate + batched commit against the oracle's Pippenger."""
import random

import numpy as np
import pytest

from graph_cases import MODS, gate_like_expression, oracle_columns, random_expression
from helpers import ints_to_mont, mont_to_ints
from mira_amd import commitment as cm
from harness import graph_evaluator as G
from oracle import cref as C
from oracle import pyref as P

pytestmark = pytest.mark.gpu


def device_columns(lib, arrs):
    ptrs, cols = [], []
    for s in arrs["selectors"]:
        p = lib.alloc(s.nbytes); lib.upload(p, s); ptrs.append(p); cols.append((p, G.COL_BOOL))
    for f in list(arrs["fixed"]) + list(arrs["advice"]):
        p = lib.alloc(f.nbytes); lib.upload(p, f); ptrs.append(p); cols.append((p, G.COL_FIELD))
    return ptrs, cols


def synth_data(field, n, nsel, nfix, nadv, nchal, seed):
    """large columns from the oracle's generator (python-int mock data is too slow at 2^17 rows)"""
    cid = 1 if field == 0 else 0                          # the curve whose scalars live in `field`
    rng = np.random.default_rng(seed)
    sel = [(rng.random(n) < 0.5).astype(np.uint8) for _ in range(nsel)]
    fix = [C.synth_scalars(cid, n, seed=seed * 100 + k) for k in range(nfix)]
    adv = [C.synth_scalars(cid, n, seed=seed * 100 + 50 + k, kind=k % 2) for k in range(nadv)]
    chal = [int(x) for x in mont_to_ints(C.synth_scalars(cid, nchal, seed=seed * 100 + 99), MODS[field])]
    return dict(selectors=sel, fixed=fix, advice=adv, challenges=chal)


@pytest.mark.parametrize("field,log_rows,seed", [(1, 12, 1), (0, 13, 2), (1, 17, 3), (0, 17, 4)])
def test_random_graphs_vs_oracle(gpu_lib, field, log_rows, seed):
    mod, n = MODS[field], 1 << log_rows
    arrs = synth_data(field, n, 2, 3, 7, 3, seed)
    ptrs, cols = device_columns(gpu_lib, arrs)
    rng = random.Random(seed)
    chal = ints_to_mont(arrs["challenges"], mod)
    try:
        singles = []
        for nterms in (1, 5, 24):
            e = gate_like_expression(rng, nterms, 7, 12, 3)
            ge = G.GraphEvaluator.new(e, field)
            code, consts, rots = ge.flatten()
            d = ge.evaluate_device(cols, arrs["challenges"], n, lib=gpu_lib)
            got = gpu_lib.download(d, (n, 4)); gpu_lib.free(d)
            want = C.graph_eval(field, code, ge.num_intermediates, consts, rots, oracle_columns(arrs), chal, n)
            assert (got == want).all(), (nterms, ge.num_intermediates)
            singles.append((ge, got))
            # spot rows against direct evaluation of the tree with Python integers
            ints = dict(selectors=arrs["selectors"], challenges=arrs["challenges"],
                        fixed=[_Lazy(f, mod) for f in arrs["fixed"]], advice=[_Lazy(a, mod) for a in arrs["advice"]])
            for r in (0, 1, n // 2, n - 1):
                assert mont_to_ints(got[r:r + 1], mod)[0] == P.eval_expression(e.to_tuple(), ints, r, n, mod)
        # the same graphs as one submission (mira_graph_eval_batch), 19 of them: two launches
        reps = 19
        d_all = gpu_lib.alloc(reps * n * 32)
        G.GraphEvaluator.evaluate_batch_device([singles[k % 3][0] for k in range(reps)], cols, arrs["challenges"], n, [d_all + k * n * 32 for k in range(reps)], lib=gpu_lib)
        got_all = gpu_lib.download(d_all, (reps, n, 4)); gpu_lib.free(d_all)
        assert all((got_all[k] == singles[k % 3][1]).all() for k in range(reps))
    finally:
        for p in ptrs:
            gpu_lib.free(p)


class _Lazy:
    """column of Montgomery limbs that converts a single row to an int on access"""

    def __init__(self, arr, mod):
        self.arr, self.mod = arr, mod

    def __getitem__(self, r):
        return mont_to_ints(self.arr[r:r + 1], self.mod)[0]

    def __len__(self):
        return len(self.arr)


@pytest.mark.parametrize("cid", [0, 1])
def test_commit_cross_terms_vs_oracle(gpu_lib, cid):
    """one fold step's cross terms at k = 15: evaluate in HBM, commit in one batch, fold into E"""
    from mira_amd import fold as FD
    field = 1 if cid == 0 else 0
    mod, rows, num_advice = MODS[field], 1 << 15, 5
    arrs = synth_data(field, rows, 1, 2, 2 * num_advice, 2, seed=40 + cid)
    lib = gpu_lib
    ptrs, cols = device_columns(lib, dict(selectors=arrs["selectors"], fixed=arrs["fixed"], advice=[]))
    W1, W2 = np.concatenate(arrs["advice"][:num_advice]), np.concatenate(arrs["advice"][num_advice:])
    d_w1, d_w2 = lib.alloc(W1.nbytes), lib.alloc(W2.nbytes)
    lib.upload(d_w1, W1); lib.upload(d_w2, W2)
    dom = G.PlonkEvalDomain(num_advice, 0, arrs["challenges"], [c[0] for c in cols[:1]], [c[0] for c in cols[1:]],
                            [(d_w1, len(W1))], [(d_w2, len(W2))], rows)
    rng = random.Random(5 + cid)
    exprs = [random_expression(rng, 6, 3 + 2 * num_advice, 2) for _ in range(4)] + [None]
    evs = [None if e is None else G.GraphEvaluator.new(e, field) for e in exprs]
    key = cm.CommitmentKey.synthetic(cid, rows, lib=lib)
    bases = key.download()
    d_terms, commits = G.commit_cross_terms(key, evs, dom, lib=lib)
    chal = ints_to_mont(arrs["challenges"], mod)
    want_terms = []
    for k, ev in enumerate(evs):
        if ev is None:
            want_terms.append(np.zeros((rows, 4), dtype=np.uint64)); continue
        code, consts, rots = ev.flatten()
        want_terms.append(C.graph_eval(field, code, ev.num_intermediates, consts, rots, oracle_columns(arrs), chal, rows))
    got_terms = lib.download(d_terms, (len(evs), rows, 4))
    for k in range(len(evs)):
        assert (got_terms[k] == want_terms[k]).all()
        assert (commits[k] == C.msm_pippenger(cid, want_terms[k], bases)).all()
    # and straight into the error-vector fold (src/plonk/mod.rs:1118-1131) without leaving HBM
    e0 = C.synth_scalars(cid, rows, seed=9)
    r = C.synth_scalars(cid, 1, seed=10)[0]
    d_e = lib.alloc(e0.nbytes); lib.upload(d_e, e0)
    FD.fold_error_device(field, d_e, [d_terms + k * rows * 32 for k in range(len(evs))], r, rows, lib=lib)
    assert (lib.download(d_e, (rows, 4)) == C.fold_error(field, e0, want_terms, r)).all()
    for p in ptrs + [d_w1, d_w2, d_terms, d_e]:
        lib.free(p)
    key.close()
