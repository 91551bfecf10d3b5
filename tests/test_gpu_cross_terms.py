"""BASELINE configs[3] on the reference's real graphs: the cross terms of one NIFS fold step at k = 17.

commit_cross_terms (src/nifs/vanilla/mod.rs:79-140) evaluates S.custom_gates_lookup_compressed.grouped()
.iter_from_first() over every row: for the circuits of an IVC step these are the grouped terms of the
homogenised MainGate<5> gate (src/main_gate.rs:543-589) -- the secondary circuit has one such gate (degree 5,
5 cross terms over 15 fixed + 2 x 7 advice columns), the primary two, compressed with a challenge (degree 6,
6 cross terms over 30 fixed + 2 x 14 advice columns).  Both at 2^17 rows, both fields, through the C ABI:

* every cross term bit for bit against the C oracle's walk of the calculation list (all rows) and against the
  direct evaluation of the grouped expression with Python integers (sampled rows);
* oracle-independent: f(W1 + X W2, c1 + X c2) = sum_k X^k T_k at a random X per sampled row, with T_1 .. T_d read
  from the GPU's output -- the relation the folded instances of src/nifs/vanilla/tests.rs:189,228 agree by;
* PlonkEvalDomain's column table and the batched commit of the terms (one point per term = the oracle's MSM).
"""
import random

import numpy as np
import pytest

from helpers import ints_to_mont, mont_to_ints
from mira_amd import commitment as cm
from harness import graph_evaluator as G
from harness import main_gate as MG
from oracle import cref as C
from oracle import pyref as P

pytestmark = pytest.mark.gpu
MODS = {0: P.P_MOD, 1: P.R_MOD}


class _Lazy:
    """column of Montgomery limbs that converts a single row to an int on access"""

    def __init__(self, arr, mod):
        self.arr, self.mod = arr, mod

    def __getitem__(self, r):
        return mont_to_ints(self.arr[r:r + 1], self.mod)[0]

    def __len__(self):
        return len(self.arr)


@pytest.mark.parametrize("cid,gates", [(0, 2), (1, 1), (0, 1), (1, 2)])
def test_main_gate_cross_terms_k17(gpu_lib, cid, gates):
    lib = gpu_lib
    field = 1 if cid == 0 else 0                               # the curve's scalar field
    mod, k = MODS[field], 17
    rows = 1 << k
    cg, ctx = MG.compressed_circuit(5, gates)
    d = cg.degree
    assert d == 4 + gates and len(cg.grouped) == d + 1
    nfix, nadv, nchal = ctx.num_fixed, ctx.num_advice, ctx.num_challenges
    # device-resident instance pair: fixed columns, W1 (accumulator), W2 (the step's witness-like trace), challenges
    d_fix = cm.synth_scalars_device(cid, nfix * rows, seed=0x7100 + cid, lib=lib)
    d_w1 = cm.synth_scalars_device(cid, nadv * rows, seed=0x7200 + cid, lib=lib)
    d_w2 = cm.synth_scalars_device(cid, nadv * rows, seed=0x7300 + cid, kind=1, lib=lib)
    rng = random.Random(0x7400 + 2 * cid + gates)
    chal = [rng.randrange(mod) for _ in range(2 * nchal)]      # [c1.., u1, c2.., u2], src/nifs/vanilla/mod.rs:87-96
    dom = G.PlonkEvalDomain(nadv, 0, chal, [], [d_fix + j * rows * 32 for j in range(nfix)], [(d_w1, nadv * rows)], [(d_w2, nadv * rows)], rows)
    evs = [G.GraphEvaluator.new(t, field) for t in cg.grouped.iter_from_first()]
    assert len(evs) == d
    key = cm.CommitmentKey.synthetic(cid, rows, lib=lib)
    d_terms, commits = G.commit_cross_terms(key, evs, dom, lib=lib)
    try:
        got = lib.download(d_terms, (d, rows, 4))
        fix = lib.download(d_fix, (nfix, rows, 4))
        w1, w2 = lib.download(d_w1, (nadv, rows, 4)), lib.download(d_w2, (nadv, rows, 4))
        cols_host = list(fix) + list(w1) + list(w2)            # eval_column_var's index space: fixed, then both instances' advice
        chal_m = ints_to_mont(chal, mod)
        bases = key.download()
        for t, ev in enumerate(evs):
            code, consts, rots = ev.flatten()
            want = C.graph_eval(field, code, ev.num_intermediates, consts, rots, cols_host, chal_m, rows)
            assert (got[t] == want).all(), (t, ev.num_intermediates)
            assert (commits[t] == C.msm_pippenger(cid, want, bases)).all(), t
        # the same d vectors from d + 1 evaluations of f and one linear combination per term (CrossTermPlan)
        plan = G.CrossTermPlan.from_compressed_gates(cg, ctx, field)
        d_plan, commits_plan = G.commit_cross_terms(key, plan, dom, lib=lib)
        try:
            assert (lib.download(d_plan, (d, rows, 4)) == got).all()
            assert (commits_plan == commits).all()
            assert sum(plan.num_calculations) < sum(ev.num_intermediates for ev in evs)
        finally:
            lib.free(d_plan)
        # sampled rows with Python integers: the grouped expressions directly, and the folding identity
        both = dict(selectors=[], fixed=[_Lazy(c, mod) for c in fix], advice=[_Lazy(c, mod) for c in list(w1) + list(w2)], challenges=chal)
        f = cg.homogeneous.to_tuple()
        terms = [t.to_tuple() for t in cg.grouped.iter()]
        for r in [0, 1, rows // 3, rows - 1] + [rng.randrange(rows) for _ in range(4)]:
            T = [P.eval_expression(terms[0], both, r, rows, mod)] + [mont_to_ints(got[t][r:r + 1], mod)[0] for t in range(d)]
            for t in range(1, d + 1):
                assert T[t] == P.eval_expression(terms[t], both, r, rows, mod)
            X = rng.randrange(mod)
            one = lambda col: mont_to_ints(col[r:r + 1], mod)[0]
            folded = dict(selectors=[], fixed=[{r: one(c)} for c in fix], challenges=[(chal[i] + X * chal[nchal + i]) % mod for i in range(nchal)],
                          advice=[{r: (one(a) + X * one(b)) % mod} for a, b in zip(w1, w2)])
            assert P.eval_expression(f, folded, r, rows, mod) == sum(pow(X, t, mod) * T[t] for t in range(d + 1)) % mod
    finally:
        for p in (d_fix, d_w1, d_w2, d_terms):
            lib.free(p)
        key.close()
