"""Development probe: per-stage timings of one MSM size under forced window widths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
widths = [int(a) for a in sys.argv[3:]] or [8, 9, 10, 11, 12, 13, 14]
for kv in filter(None, os.environ.get("TUNE", "").split(",")):     # TUNE=12=3,13=2: mira_set_tuning(knob, value)
    k, v = kv.split("="); lib.tune(int(k), int(v))
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n, kind=kind)
for c in widths:
    lib.check(lib.c.mira_msm_set_window_bits(c))
    key.commit_device(d, n)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); key.commit_device(d, n); ts.append((time.perf_counter() - t0) * 1e3)
    lib.check(lib.c.mira_set_timing(1))
    acc = {}
    for _ in range(5):
        key.commit_device(d, n)
        for name, ms in lib.timings():
            acc[name] = acc.get(name, 0) + ms / 5
    lib.check(lib.c.mira_set_timing(0))
    print(f"n {n} kind {kind} c={c}: wall {sorted(ts)[3]:.3f} ms", {a: round(b, 3) for a, b in acc.items()}, flush=True)
