"""Development probe: cross-term evaluation kernel time against the number of rows (occupancy)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
from harness import graph_evaluator as G
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
nadv = 8
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
ev = G.GraphEvaluator.new(gate(0), G.FIELD_FR)
chal = [0x1234567 + 977 * j for j in range(2)]
for k in (15, 16, 17, 18, 19, 20):
    n = 1 << k
    d_cols = cm.synth_scalars_device(cm.CURVE_BN256, (nadv + 2) * n, seed=0x3000)
    sel = np.ones(n, dtype=np.uint8); sel[::7] = 0
    d_sel = lib.alloc(n); lib.upload(d_sel, sel)
    cols = [(d_sel, G.COL_BOOL)] + [(d_cols + j * n * 32, G.COL_FIELD) for j in range(nadv + 2)]
    d_out = lib.alloc(n * 32)
    ev.evaluate_device(cols, chal, n, d_out=d_out)
    lib.check(lib.c.mira_set_timing(1))
    ks = []
    for _ in range(5):
        ev.evaluate_device(cols, chal, n, d_out=d_out)
        ks.append(dict(lib.timings())["graph_eval"])
    lib.check(lib.c.mira_set_timing(0))
    ws = []
    for _ in range(5):
        t0 = time.perf_counter(); ev.evaluate_device(cols, chal, n, d_out=d_out); ws.append((time.perf_counter() - t0) * 1e3)
    print("rows 2^%d: kernel %.4f ms, wall %.4f ms, %.1f ns/row-kernel" % (k, sorted(ks)[2], sorted(ws)[2], sorted(ks)[2] * 1e6 / n), flush=True)
    for p in (d_cols, d_sel, d_out): lib.free(p)
