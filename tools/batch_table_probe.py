"""Development probe: stage timings of the batched cross-term commits of a k = 17 fold step over shared-bucket tables of every width."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
n = 1 << 17
for cid, cnt in ((0, 6), (1, 5)):
    key = cm.CommitmentKey.synthetic(cid, n)
    d = cm.synth_scalars_device(cid, cnt * n, seed=77)
    for c in (0, 10, 11, 12, 13, 14, 15, 16):
        if c:
            key.precompute(c)
            lib.tune(_lib.TUNE_TABLE_WIDTH, c)
        key.commit_batch_device(d, n, cnt)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); key.commit_batch_device(d, n, cnt); ts.append((time.perf_counter() - t0) * 1e3)
        lib.check(lib.c.mira_set_timing(1)); key.commit_batch_device(d, n, cnt); st = {a: round(b, 3) for a, b in lib.timings()}; lib.check(lib.c.mira_set_timing(0))
        print(f"curve {cid} batch {cnt} x 2^17 tables c={c}: {sorted(ts)[3]:.3f} ms  kernels {sum(st.values()):.3f} {st}", flush=True)
    lib.tune(_lib.TUNE_TABLE_WIDTH, -1)
    key.close(); lib.free(d)
