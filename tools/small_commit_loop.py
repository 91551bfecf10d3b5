"""Development probe: 200 commits of 131 072 pairs (for a rocprofv3 kernel trace: kernel time vs wall)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
if len(sys.argv) > 2: lib.check(lib.c.mira_msm_set_window_bits(int(sys.argv[2])))      # forced width
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n, kind=kind)
for _ in range(5): key.commit_device(d, n)
t0 = time.perf_counter()
for _ in range(200): key.commit_device(d, n)
print("wall per commit %.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
