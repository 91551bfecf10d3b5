"""Host side of an instance fold (mira_g1_mul_add, mira_g1_lincomb, mira_g1_fold_commitments): wall time of each call on this
machine's cores.  No device work.  `python tools/host_fold_probe.py`"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, fold as FD
from oracle import cref as C

lib = _lib.load()
for cid in (0, 1):
    pts = C.synth_bases(cid, 16)
    sc = C.synth_scalars(cid, 8, 3)

    def t(name, f):
        f()
        ts = []
        for _ in range(101):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        print("curve %d  %-34s median %.3f ms  min %.3f" % (cid, name, sorted(ts)[50], min(ts)), flush=True)
    t("g1_mul_add", lambda: FD.g1_mul_add(cid, pts[0], sc[1], pts[1], lib=lib))
    for c in (1, 2, 3, 5, 6):
        t("g1_lincomb, %d terms" % c, lambda: FD.g1_lincomb(cid, pts[0], sc[:c], pts[1:1 + c], lib=lib))
    t("fold_commitments 1 W + 6 T", lambda: FD.fold_instance_commitments(cid, pts[:1], pts[1:2], sc[0], pts[2], pts[3:9], lib=lib))
    t("fold_commitments 2 W + 5 T", lambda: FD.fold_instance_commitments(cid, pts[:2], pts[2:4], sc[0], pts[4], pts[5:10], lib=lib))
