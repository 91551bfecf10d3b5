"""Development probe: commits over 16-bit shared-bucket tables against the per-window path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
for cid, n, kind in ((0, 1 << 13, 0), (0, 1 << 15, 0), (0, 131072, 0), (1, 131072, 0), (0, 1 << 18, 0), (0, 14 << 17, 1), (1, 7 << 17, 1), (0, 1 << 20, 0), (0, 1 << 22, 0)):
    key = cm.CommitmentKey.synthetic(cid, n); d = cm.synth_scalars_device(cid, n, kind=kind)
    def med(reps=9):
        key.commit_device(d, n); key.commit_device(d, n)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[reps // 2], out
    t0, p0 = med()
    key.precompute(16)
    t1, p1 = med()
    lib.check(lib.c.mira_set_timing(1)); key.commit_device(d, n); st = {a: round(b, 3) for a, b in lib.timings()}; lib.check(lib.c.mira_set_timing(0))
    print(f"curve {cid} n {n} kind {kind}: per-window {t0:.3f} ms  16-bit tables {t1:.3f} ms  same {bool((p0 == p1).all())}  {st}", flush=True)
    key.close(); lib.free(d)
