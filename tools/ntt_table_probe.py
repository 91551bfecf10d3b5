import os, sys
sys.path.insert(0, "/root/repo")
from mira_amd import _lib, commitment as cm, fft as F
lib = _lib.load()
k = 24
d = cm.synth_scalars_device(0, 1 << k, seed=5)
def run(tag):
    for _ in range(20): F.fft_device(d, k)
    lib.check(lib.c.mira_set_timing(1))
    tot, acc = [], {}
    for _ in range(40):
        F.fft_device(d, k)
        t = dict(lib.timings())
        tot.append(sum(v for a, v in t.items() if a.startswith("ntt_")))
        for a, v in t.items(): acc.setdefault(a, []).append(v)
    lib.check(lib.c.mira_set_timing(0))
    print(tag, "median %.4f ms" % sorted(tot)[20], {a: round(sorted(v)[20], 4) for a, v in acc.items() if a.startswith("ntt_")}, flush=True)
for rep in range(2):
    lib.tune(_lib.TUNE_NTT_FULL_TW_MAX_LOG, -1); run("n-entry table (default)")
    lib.tune(_lib.TUNE_NTT_FULL_TW_MAX_LOG, 0); run("two-table product     ")
