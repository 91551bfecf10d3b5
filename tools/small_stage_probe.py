"""Development probe: stage timings of one small commit beside its wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
if os.environ.get("MIRA_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [131072]:
    key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
    for _ in range(5): key.commit_device(d, n)
    t0 = time.perf_counter()
    for _ in range(200): key.commit_device(d, n)
    wall = (time.perf_counter() - t0) / 200 * 1e3
    lib.check(lib.c.mira_set_timing(1))
    acc = {}
    for _ in range(20):
        key.commit_device(d, n)
        for name, ms in lib.timings():
            acc[name] = acc.get(name, 0) + ms / 20
    lib.check(lib.c.mira_set_timing(0))
    print("n=%d wall %.4f ms, kernels %.4f ms" % (n, wall, sum(acc.values())), {a: round(b, 4) for a, b in acc.items()}, flush=True)
