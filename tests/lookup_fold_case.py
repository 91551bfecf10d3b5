"""Shared body of the lookup-argument fold-step tests (CPU: tests/test_lookup_fold.py on the emulation; GPU:
tests/test_gpu_lookup_fold.py through libmira_gpu.so): the reference's `three_rounds_test`
(src/nifs/vanilla/tests.rs:313-340 -> `fold_instances`, :137-244) on the device path.

One NIFS fold of an accumulator with an incoming trace of `FiboCircuitWithLookup` -- THREE witness vectors per instance
(advice | l, t, m | h, g: run_sps_protocol_3) and the index mapping of `PlonkEvalDomain::eval_advice_var` with
num_lookups = 1 (src/plonk/eval.rs:152-228):

  cross terms   every grouped term T_1 .. T_d evaluated on the device over both instances' three vectors
                = the oracle's walk of the calculation list, all rows; sampled rows = the direct evaluation of the grouped
                expression with Python integers, and sum_k X^k T_k = f(folded at X) at a random X
  commit        one batched MSM = the oracle's commitment of every term
  fold          W_i' = W1_i + r W2_i for the three vectors, E' = E + sum_k r^k T_k on the device = the oracle's fold;
                commitments folded on the host side of the library (mira_g1_fold_commitments)
  is_sat_relaxed (src/plonk/mod.rs:505-557)  f_homogeneous(folded W', challenges', u') = E' on every row (device
                evaluation with W2s = []), Com(W_i') = the folded commitments, Com(E') = the folded E commitment,
                and the log-derivative sums of the folded (h, g) agree (:589-621)"""
import random

import numpy as np

from helpers import ints_to_mont, mont_to_ints
from mira_amd import commitment as cm
from mira_amd import fold as FD
from harness import graph_evaluator as G
from harness import lookup as LK
from oracle import cref as C
from oracle import pyref as P

FIELD, CID = 1, 0                       # bn256::Fr, committed on BN256 G1 (the reference's test: G1Affine / Fr)
MOD = P.R_MOD
WIDTHS = (LK.NUM_ADVICE, 3 * LK.NUM_LOOKUPS, 2 * LK.NUM_LOOKUPS)      # columns of W1, W2, W3 (round_sizes / 2^k)


class DeviceTrace:
    """a (relaxed) instance-witness pair with its three witness vectors and E resident on the device"""

    def __init__(self, lib, key, rows, W_ints=None, challenges=(0, 0, 0), u=0):
        self.lib, self.rows = lib, rows
        self.W_host = [ints_to_mont(w, MOD) if w is not None else np.zeros((n * rows, 4), dtype=np.uint64)
                       for w, n in zip(W_ints or [None] * 3, WIDTHS)]
        self.d_W = []
        for w in self.W_host:
            p = lib.alloc(w.nbytes); lib.upload(p, w); self.d_W.append(p)
        self.E_host = np.zeros((rows, 4), dtype=np.uint64)
        self.d_E = lib.alloc(self.E_host.nbytes); lib.upload(self.d_E, self.E_host)
        self.challenges, self.u = list(challenges), u
        # RelaxedPlonkInstance::new: default commitments (the identity); a trace: the commitments of its vectors
        self.W_commits = np.stack([key.commit_device(p, n * rows) if W_ints is not None else np.zeros(8, dtype=np.uint64)
                                   for p, n in zip(self.d_W, WIDTHS)])
        self.E_commit = np.zeros(8, dtype=np.uint64)

    def free(self):
        for p in self.d_W + [self.d_E]:
            self.lib.free(p)


def run_lookup_fold(lib, log_rows, traces, seed, sample_rows=6):
    """Fold the default accumulator with traces[0], the result with traces[1], ...; every step checked as described above.
    traces: LookupTrace objects over 2^log_rows rows."""
    rows = 1 << log_rows
    rng = random.Random(seed)
    cg, ctx, _, _ = LK.compressed_fibo_lookup()
    d = cg.degree
    assert ctx.num_lookups == 1 and ctx.num_fold_vars() == 8 and ctx.num_challenges == 4          # r1, r2, r3, u
    evs = [None if t is None else G.GraphEvaluator.new(t, FIELD) for t in cg.grouped.iter_from_first()]
    assert len(evs) == d and all(ev is not None for ev in evs)
    f_ev = G.GraphEvaluator.new(cg.homogeneous, FIELD)
    key = cm.CommitmentKey.synthetic(CID, 3 * rows, seed=0x4C4B + log_rows, lib=lib)
    bases = key.download()
    t0 = traces[0]
    sel_host = [np.array(s, dtype=np.uint8) for s in t0.selectors]
    fix_host = ints_to_mont([v for col in t0.fixed for v in col], MOD).reshape(LK.NUM_FIXED, rows, 4)
    d_sel, d_fix = [], []
    for s in sel_host:
        p = lib.alloc(s.nbytes); lib.upload(p, s); d_sel.append(p)
    for fcol in fix_host:
        p = lib.alloc(fcol.nbytes); lib.upload(p, np.ascontiguousarray(fcol)); d_fix.append(p)
    acc = DeviceTrace(lib, key, rows)                                            # RelaxedPlonkInstance::new / RelaxedPlonkWitness::new
    live = [acc]
    try:
        for step, tr in enumerate(traces):
            assert tr.selectors == t0.selectors and tr.fixed == t0.fixed        # traces of ONE PlonkStructure
            tr.check_is_sat(cg.compressed) if rows <= 64 else None
            inc = DeviceTrace(lib, key, rows, tr.W, tr.challenges, 1)            # to_relax: u = 1, E = 0
            live.append(inc)
            # the commitments of the incoming vectors are the oracle's
            for i, (w, n) in enumerate(zip(inc.W_host, WIDTHS)):
                assert (inc.W_commits[i] == C.commit(CID, bases, w)).all(), (step, i)
            chal = acc.challenges + [acc.u] + inc.challenges + [inc.u]           # src/nifs/vanilla/mod.rs:90
            dom = G.PlonkEvalDomain(LK.NUM_ADVICE, LK.NUM_LOOKUPS, chal, d_sel, d_fix,
                                    [(p, n * rows) for p, n in zip(acc.d_W, WIDTHS)], [(p, n * rows) for p, n in zip(inc.d_W, WIDTHS)], rows)
            cols = dom.columns()
            assert len(cols) == 2 + 3 + 2 * 8 and all(c is not None for c in cols)
            # the index map itself: variable v of instance i is column v of the vector the reference names
            where = {0: (0, 0), 1: (0, 1), 2: (0, 2), 3: (1, 0), 4: (1, 1), 5: (1, 2), 6: (2, 0), 7: (2, 1)}      # a b out | l t m | h g
            for inst, tr_dev in enumerate((acc, inc)):
                for v, (vec, col) in where.items():
                    assert cols[5 + 8 * inst + v][0] == tr_dev.d_W[vec] + col * rows * 32
            d_terms, commits = G.commit_cross_terms(key, evs, dom, lib=lib)
            try:
                got = lib.download(d_terms, (d, rows, 4))
                host_cols = sel_host + list(fix_host)
                for tr_dev in (acc, inc):
                    flat = np.concatenate(tr_dev.W_host).reshape(8, rows, 4)
                    host_cols += list(flat)
                chal_m = ints_to_mont(chal, MOD)
                for t, ev in enumerate(evs):
                    code, consts, rots = ev.flatten()
                    want = C.graph_eval(FIELD, code, ev.num_intermediates, consts, rots, host_cols, chal_m, rows)
                    assert (got[t] == want).all(), (step, t)
                    assert (commits[t] == C.commit(CID, bases, want)).all(), (step, t)
                # sampled rows with Python integers: the grouped terms directly and the folding identity at a random X
                nch = ctx.num_challenges
                ints = lambda arr: [mont_to_ints(c, MOD) for c in arr]
                var1, var2 = ints(np.concatenate(acc.W_host).reshape(8, rows, 4)), ints(np.concatenate(inc.W_host).reshape(8, rows, 4))
                both = dict(selectors=t0.selectors, fixed=t0.fixed, advice=var1 + var2, challenges=chal)
                terms = [t.to_tuple() for t in cg.grouped.iter()]
                f = cg.homogeneous.to_tuple()
                for r_ in [0, 1, rows - 1] + [rng.randrange(rows) for _ in range(sample_rows)]:
                    T = [P.eval_expression(terms[0], both, r_, rows, MOD)] + [mont_to_ints(got[t][r_:r_ + 1], MOD)[0] for t in range(d)]
                    for t in range(1, d + 1):
                        assert T[t] == P.eval_expression(terms[t], both, r_, rows, MOD), (step, t, r_)
                    X = rng.randrange(MOD)
                    folded = dict(selectors=t0.selectors, fixed=t0.fixed, challenges=[(chal[i] + X * chal[nch + i]) % MOD for i in range(nch)],
                                  advice=[{r_: (a[r_] + X * b[r_]) % MOD} for a, b in zip(var1, var2)])
                    assert P.eval_expression(f, folded, r_, rows, MOD) == sum(pow(X, t, MOD) * T[t] for t in range(d + 1)) % MOD
                # ---- fold (the folding challenge r comes from the random oracle in the reference; any field element here)
                r = rng.randrange(MOD)
                r_m = ints_to_mont([r], MOD)[0]
                new_W_host = []
                for i, n in enumerate(WIDTHS):
                    FD.fold_witness_device(FIELD, acc.d_W[i], acc.d_W[i], inc.d_W[i], r_m, n * rows, lib=lib)           # in place: W1 <- W1 + r W2
                    want_w = C.fold_witness(FIELD, acc.W_host[i], inc.W_host[i], r_m)
                    assert (lib.download(acc.d_W[i], (n * rows, 4)) == want_w).all(), (step, i)
                    new_W_host.append(want_w)
                FD.fold_error_device(FIELD, acc.d_E, [d_terms + t * rows * 32 for t in range(d)], r_m, rows, lib=lib)
                want_e = C.fold_error(FIELD, acc.E_host.copy(), [got[t] for t in range(d)], r_m)
                assert (lib.download(acc.d_E, (rows, 4)) == want_e).all(), step
                new_wc, new_ec = FD.fold_instance_commitments(CID, acc.W_commits, inc.W_commits, r_m, acc.E_commit, commits, lib=lib)
                acc.W_host, acc.E_host, acc.W_commits, acc.E_commit = new_W_host, want_e, np.asarray(new_wc), np.asarray(new_ec)
                acc.challenges = [(a + r * b) % MOD for a, b in zip(acc.challenges, inc.challenges)]
                acc.u = (acc.u + r) % MOD
            finally:
                lib.free(d_terms)
            # ---- is_sat_relaxed on the folded pair
            dom_f = G.PlonkEvalDomain(LK.NUM_ADVICE, LK.NUM_LOOKUPS, acc.challenges + [acc.u], d_sel, d_fix,
                                      [(p, n * rows) for p, n in zip(acc.d_W, WIDTHS)], [], rows)
            d_f = f_ev.evaluate_device(dom_f.columns(), dom_f.challenges, rows, lib=lib)
            try:
                assert (lib.download(d_f, (rows, 4)) == acc.E_host).all(), step                    # f(W', c', u') = E' on every row
            finally:
                lib.free(d_f)
            for i, n in enumerate(WIDTHS):
                assert (key.commit_device(acc.d_W[i], n * rows) == acc.W_commits[i]).all(), (step, i)   # Com(W1 + r W2) = Com(W1) + r Com(W2)
                assert (acc.W_commits[i] == C.commit(CID, bases, acc.W_host[i])).all()
            assert (key.commit_device(acc.d_E, rows) == acc.E_commit).all(), step                  # Com(E + sum r^k T_k) = Com(E) + sum r^k Com(T_k)
            assert (acc.E_commit == C.commit(CID, bases, acc.E_host)).all()
            hg = mont_to_ints(acc.W_host[2], MOD)
            assert (sum(hg[:rows]) - sum(hg[rows:])) % MOD == 0, step                              # is_sat_log_derivative on the folded (h, g)
            live.remove(inc); inc.free()
    finally:
        for tr_dev in live:
            tr_dev.free()
        for p in d_sel + d_fix:
            lib.free(p)
        for ev in evs + [f_ev]:
            ev.close()
        key.close()
    return d
