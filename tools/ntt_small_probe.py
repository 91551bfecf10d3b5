import sys, time
sys.path.insert(0, "/root/repo")
from mira_amd import _lib, commitment as cm, fft as F
lib = _lib.load()
for k in (9, 10, 11, 12, 13):
    d = cm.synth_scalars_device(0, 1 << k, seed=5)
    row = []
    for ml, wave in ((-1, -1), (5, 0), (6, 0), (7, 0), (8, 0), (6, 1), (7, 1), (8, 1)):
        if ml > 0 and (k > 3 * ml or k <= ml): 
            continue
        lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, ml); lib.tune(_lib.TUNE_NTT_WAVE, wave)
        for _ in range(30): F.fft_device(d, k)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter(); F.fft_device(d, k); ts.append((time.perf_counter() - t0) * 1e6)
        row.append("line<=2^%d%s: %.1f" % (ml, {-1: "", 0: " wg", 1: " wave"}[wave], sorted(ts)[30]) if ml > 0 else "default: %.1f" % sorted(ts)[30])
    print("2^%d wall us:  " % k + "   ".join(row), flush=True)
    lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1); lib.tune(_lib.TUNE_NTT_WAVE, -1); lib.free(d)
