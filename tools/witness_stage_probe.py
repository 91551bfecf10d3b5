"""Development probe: stage timings of the k = 17 witness commits (witness-like scalars, 14 and 7 columns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
for kv in filter(None, os.environ.get("TUNE", "").split(",")):     # TUNE=12=3,13=2: mira_set_tuning(knob, value)
    k_, v_ = kv.split("="); lib.tune(int(k_), int(v_))
for cid, cols in ((0, 14), (1, 7)):
    n = cols << 17
    key = cm.CommitmentKey.synthetic(cid, n); d = cm.synth_scalars_device(cid, n, seed=0x1000 + cid, kind=1)
    for tables in (0, 13):
        if tables: key.precompute(tables)
        for _ in range(16): key.commit_device(d, n)      # incl. the width trials of the shape
        t0 = time.perf_counter()
        for _ in range(20): key.commit_device(d, n)
        wall = (time.perf_counter() - t0) / 20 * 1e3
        lib.check(lib.c.mira_set_timing(1))
        acc = {}
        for _ in range(10):
            key.commit_device(d, n)
            for name, ms in lib.timings():
                acc[name] = acc.get(name, 0) + ms / 10
        lib.check(lib.c.mira_set_timing(0))
        print("curve %d n=%d tables=%d wall %.4f ms, kernels %.4f ms" % (cid, n, tables, wall, sum(acc.values())), {a: round(b, 4) for a, b in acc.items()}, flush=True)
    key.close(); lib.free(d)
