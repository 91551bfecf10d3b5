"""Development probe: MSM throughput over the BASELINE size range (2^16 ... 2^26), both curves,
automatic window width, per-window buckets and fixed-base tables.  Median of 5 calls each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()


def med(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2], r


print("curve log_n  per-window ms  M pairs/s | tables ms  M pairs/s  same point", flush=True)
for cid in (0, 1):
    for log_n in (16, 18, 20, 22, 24) + ((26,) if cid == 0 else ()):
        n = 1 << log_n
        key = cm.CommitmentKey.synthetic(cid, n)
        d = cm.synth_scalars_device(cid, n)
        key.commit_device(d, n)
        t, p = med(lambda: key.commit_device(d, n))
        key.precompute()
        key.commit_device(d, n)
        tt, pt = med(lambda: key.commit_device(d, n))
        print(f"{cid:5d} {log_n:5d} {t * 1e3:13.3f} {n / t / 1e6:10.1f} | {tt * 1e3:9.3f} {n / tt / 1e6:10.1f}  {bool((p == pt).all())}", flush=True)
        key.close(); lib.free(d)
