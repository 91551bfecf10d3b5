"""Development probe: batch evaluation against one-by-one, repeated, reporting which graph differs."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from graph_cases import gate_like_expression
from mira_amd import _lib, commitment as cm
from harness import graph_evaluator as G
lib = _lib.load()
field, n = 1, 1 << 12
rng = random.Random(1)
ncols = 12
d_cols = cm.synth_scalars_device(cm.CURVE_BN256, ncols * n, seed=0x3000)
cols = [(d_cols + j * n * 32, G.COL_FIELD) for j in range(ncols)]
chal = [5, 7, 11]
evs = [G.GraphEvaluator.new(gate_like_expression(rng, t, 7, 12, 3), field) for t in (1, 5, 24)]
single = []
for ev in evs:
    d = ev.evaluate_device(cols, chal, n); single.append(lib.download(d, (n, 4))); lib.free(d)
for reps in (3, 16, 19, 33):
    d_all = lib.alloc(reps * n * 32)
    bad = {}
    for it in range(30):
        G.GraphEvaluator.evaluate_batch_device([evs[k % 3] for k in range(reps)], cols, chal, n, [d_all + k * n * 32 for k in range(reps)])
        got = lib.download(d_all, (reps, n, 4))
        for k in range(reps):
            if not (got[k] == single[k % 3]).all():
                rows = np.nonzero((got[k] != single[k % 3]).any(axis=1))[0]
                bad[k] = bad.get(k, 0) + 1
    print("reps", reps, "bad graphs -> iterations", bad, flush=True)
    lib.free(d_all)
