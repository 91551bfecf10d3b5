"""Development probe: the stage timings of one commit per size under MIRA_PROBE_LIB (same-box A/B of library variants built by
tools/build_probe_variants.sh), with a digest of the point.  usage: [MIRA_PROBE_LIB=tools/_variants/x.so] python tools/stage_ab_probe.py"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
name = os.path.basename(os.environ.get("MIRA_PROBE_LIB", "tree"))
for log_n, c in ((17, 0), (20, 16), (22, 16)):
    n = 1 << log_n
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
    lib.check(lib.c.mira_msm_set_window_bits(c))
    for _ in range(20 if c == 0 else 6):
        p = key.commit_device(d, n)
    lib.check(lib.c.mira_set_timing(1))
    acc, walls = {}, []
    for _ in range(15):
        t0 = time.perf_counter(); p = key.commit_device(d, n); walls.append((time.perf_counter() - t0) * 1e3)
        for k_, v in lib.timings():
            acc.setdefault(k_, []).append(v)
    lib.check(lib.c.mira_set_timing(0))
    med = {k_: round(sorted(v)[len(v) // 2] * 1e3, 1) for k_, v in acc.items()}
    print(f"{name:16s} 2^{log_n} c={c:2d} wall {sorted(walls)[7]:.4f} ms  us: {med}  point {hashlib.sha1(p.tobytes()).hexdigest()[:10]}", flush=True)
    key.close(); lib.free(d)
