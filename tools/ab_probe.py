"""Development probe: one 2^k MSM under 16-bit windows -- accumulate time, wall, and a digest of the point (compare across
MIRA_PROBE_LIB variants on one box).  usage: MIRA_PROBE_LIB=tools/_variants/x.so python tools/ab_probe.py [k]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
from mira_amd import commitment as cm
lib = _lib.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << k
lib.check(lib.c.mira_msm_set_window_bits(16))
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
for _ in range(3): p = key.commit_device(d, n)
lib.check(lib.c.mira_set_timing(1))
acc, walls = [], []
for _ in range(10):
    t0 = time.perf_counter(); p = key.commit_device(d, n); walls.append((time.perf_counter() - t0) * 1e3)
    acc.append(dict(lib.timings())["accumulate"])
print(f"{os.environ.get('MIRA_PROBE_LIB', 'tree'):32s} k={k} accumulate {sorted(acc)[5]:.4f} ms  wall {sorted(walls)[5]:.4f} ms  point {hashlib.sha1(p.tobytes()).hexdigest()[:12]}", flush=True)
