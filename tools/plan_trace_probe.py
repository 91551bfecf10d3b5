"""Development probe: the width the planner / the width trials use on each of the first commits of a shape over a fresh key."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib, commitment as cm
lib = _lib.load()
glv = int(os.environ.get("GLV", "0"))
lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, -1 if glv else 0)
for n in [int(a) for a in sys.argv[1:]] or [131072]:
    kind = int(os.environ.get("KIND", "0")); key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n, kind=kind)
    row = []
    for i in range(20):
        t0 = time.perf_counter(); key.commit_device(d, n); dt = (time.perf_counter() - t0) * 1e3
        c, w = ctypes.c_int32(), ctypes.c_int32()
        lib.check(lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
        row.append("%d:%.3f" % (c.value, dt))
    print("n=%d glv=%d  c:ms per commit  " % (n, glv) + " ".join(row), flush=True)
    key.close(); lib.free(d)
