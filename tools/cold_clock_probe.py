"""Development probe: what an idle GPU costs the first work after it -- the k = 17 cross-term batch (6 x 2^17 pairs) and one 2^24 NTT,
timed right after a pause of the given length, against the same call under a sustained load."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm, fft as F
lib = _lib.load()
n = 1 << 17
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, 6 * n, seed=77)
dn = cm.synth_scalars_device(0, 1 << 24, seed=5)
def batch(): key.commit_batch_device(d, n, 6)
def ntt(): F.fft_device(dn, 24)
for name, fn in (("batch 6 x 2^17", batch), ("ntt 2^24", ntt)):
    for _ in range(30): fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    warm = sorted(ts)[10]
    row = []
    for pause in (0.001, 0.01, 0.05, 0.2, 1.0, 3.0):
        cold = []
        for _ in range(5):
            for _ in range(10): fn()
            time.sleep(pause)
            t0 = time.perf_counter(); fn(); cold.append((time.perf_counter() - t0) * 1e3)
        row.append("%gs: %.3f" % (pause, sorted(cold)[2]))
    print("%-16s sustained %.3f ms; first call after a pause of  %s" % (name, warm, "  ".join(row)), flush=True)

# where the penalty sits: the kernels' own event timings of the first call after a pause beside its wall time
lib.check(lib.c.mira_set_timing(1))
for name, fn in (("batch 6 x 2^17", batch), ("ntt 2^24", ntt)):
    for pause in (0.0, 1.0):
        walls, kerns = [], []
        for _ in range(5):
            for _ in range(10): fn()
            if pause: time.sleep(pause)
            t0 = time.perf_counter(); fn(); walls.append((time.perf_counter() - t0) * 1e3)
            kerns.append(sum(ms for _, ms in lib.timings()))
        print("%-16s pause %gs: wall %.3f ms, kernels (events) %.3f ms" % (name, pause, sorted(walls)[2], sorted(kerns)[2]), flush=True)
lib.check(lib.c.mira_set_timing(0))
