"""Development probe: time the 2^24 NTT passes under MIRA_NTT_DEBUG_SKIP variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm, fft as F
lib = _lib.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = cm.synth_scalars_device(0, 1 << k, seed=5)
F.fft_device(d, k)
lib.check(lib.c.mira_set_timing(1))
acc = {}
for _ in range(5):
    F.fft_device(d, k)
    for name, ms in lib.timings():
        acc[name] = acc.get(name, 0) + ms / 5
print("skip=%s" % os.environ.get("MIRA_NTT_DEBUG_SKIP", "0"), {a: round(b, 3) for a, b in acc.items()})
