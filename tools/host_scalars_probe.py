"""Development probe: commit with host scalars (mira_msm) against the HBM-resident commit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for cid, n, kind, c in ((0, 1 << 22, 0, 16), (0, 1 << 20, 0, 0), (0, 14 << 17, 1, 0), (1, 7 << 17, 1, 0), (0, 1 << 17, 0, 0)):
    lib.check(lib.c.mira_msm_set_window_bits(c))
    key = cm.CommitmentKey.synthetic(cid, n); d = cm.synth_scalars_device(cid, n, kind=kind)
    sc = lib.download(d, (n, 4))
    res = {}
    for name, fn in (("device", lambda: key.commit_device(d, n)), ("host", lambda: key.commit(sc))):
        fn(); fn()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e3)
        res[name] = (sorted(ts)[3], out)
    lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, 1 << 40)
    key.commit(sc)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); out1 = key.commit(sc); ts.append((time.perf_counter() - t0) * 1e3)
    lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, -1)
    print(f"curve {cid} n {n} kind {kind}: device {res['device'][0]:.3f} ms  host chunked {res['host'][0]:.3f} ms (+{(res['host'][0] / res['device'][0] - 1) * 100:.1f} %)  host one copy {sorted(ts)[3]:.3f} ms  same point {bool((res['host'][1] == res['device'][1]).all() and (out1 == res['device'][1]).all())}", flush=True)
    key.close(); lib.free(d)
lib.check(lib.c.mira_msm_set_window_bits(0))
