"""Development probe: wall time of the five-graph cross-term batch at k = 17 (bench.py's gate)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
from harness import graph_evaluator as G
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
nadv, n = 8, 1 << 17
def gate(shift):
    sbox = []
    for i in range(nadv):
        x = G.Sum(G.Polynomial(3 + i), G.Constant(1000 + i + shift))
        x2 = G.Product(x, x)
        sbox.append(G.Product(G.Product(x2, x2), x))
    e = None
    for j in range(nadv):
        acc = None
        for i in range(nadv):
            t = G.Scaled(sbox[i], 17 * j + 3 * i + 2 + shift)
            acc = t if acc is None else G.Sum(acc, t)
        row = G.Product(G.Polynomial(0), G.Sum(acc, G.Negated(G.Polynomial(3 + j, 1))))
        e = row if e is None else G.Sum(G.Product(e, G.Challenge(j % 2)), row)
    return G.Sum(e, G.Product(G.Polynomial(1), G.Polynomial(2, -1)))
d_cols = cm.synth_scalars_device(cm.CURVE_BN256, (nadv + 2) * n, seed=0x3000)
sel = np.ones(n, dtype=np.uint8); sel[::7] = 0
d_sel = lib.alloc(n); lib.upload(d_sel, sel)
cols = [(d_sel, G.COL_BOOL)] + [(d_cols + j * n * 32, G.COL_FIELD) for j in range(nadv + 2)]
chal = [0x1234567 + 977 * j for j in range(2)]
for cnt in (1, 5, 6, 11):
    evs = [G.GraphEvaluator.new(gate(s), G.FIELD_FR) for s in range(cnt)]
    d_out = lib.alloc(cnt * n * 32)
    run = lambda: G.GraphEvaluator.evaluate_batch_device(evs, cols, chal, n, [d_out + i * n * 32 for i in range(cnt)])
    run(); run()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    print("graphs %d: %.3f ms (%.3f per graph)" % (cnt, sorted(ts)[4], sorted(ts)[4] / cnt), flush=True)
    lib.free(d_out)
