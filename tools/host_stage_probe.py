import os, sys, time
sys.path.insert(0, "/root/repo")
from mira_amd import _lib, commitment as cm
lib = _lib.load()
n = 1 << 22
lib.check(lib.c.mira_msm_set_window_bits(16))
key = cm.CommitmentKey.synthetic(0, n); d = cm.synth_scalars_device(0, n)
sc = lib.download(d, (n, 4))
for name, fn in (("device", lambda: key.commit_device(d, n)), ("host", lambda: key.commit(sc))):
    fn(); fn()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    lib.check(lib.c.mira_set_timing(1)); fn()
    acc = {}
    for nm, ms in lib.timings(): acc[nm] = acc.get(nm, 0) + ms
    lib.check(lib.c.mira_set_timing(0))
    print(name, "wall %.3f" % sorted(ts)[3], "kernels %.3f" % sum(acc.values()), {a: round(b, 3) for a, b in acc.items()}, flush=True)
