"""Development probe: time the passes of one NTT size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])   # development: a variant build
from mira_amd import commitment as cm, fft as F
lib = _lib.load()
if os.environ.get("MIRA_PROBE_WAVE"):
    lib.tune(_lib.TUNE_NTT_WAVE, int(os.environ["MIRA_PROBE_WAVE"]))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
WARM, REPS = int(os.environ.get("MIRA_PROBE_WARM", "20")), int(os.environ.get("MIRA_PROBE_REPS", "40"))
d = cm.synth_scalars_device(0, 1 << k, seed=5)
for _ in range(1 + WARM):                      # the first builds the twiddle tables; the rest let the clock settle under the load
    F.fft_device(d, k)
lib.check(lib.c.mira_set_timing(1))
acc = {}
for _ in range(REPS):
    F.fft_device(d, k)
    for name, ms in lib.timings():
        acc[name] = acc.get(name, 0) + ms / REPS
print("k=%d" % k, {a: round(b, 3) for a, b in acc.items()})
