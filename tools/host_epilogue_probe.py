"""Development probe: the host epilogue of a commit (Horner over the window sums + to_affine) on the box's CPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mira_amd import _lib, commitment as cm
lib = _lib.load()
n = 4096
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
for c in (8, 13, 16):
    part, cc, w = key.commit_partial_device(0, d, n, window_bits=c)
    parts = np.stack([part])
    cm.combine_partials(0, parts, cc, w)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(100): cm.combine_partials(0, parts, cc, w)
        ts.append((time.perf_counter() - t0) / 100 * 1e6)
    print("c=%d W=%d: %.1f us per epilogue (median of 7 x 100; min %.1f max %.1f)" % (cc, w, sorted(ts)[3], min(ts), max(ts)), flush=True)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
