"""Development probe: NTT pass timings over a range of sizes, wave-level kernel against the
workgroup-level one (MIRA_TUNE_NTT_WAVE), with an optional forced line length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
from mira_amd import commitment as cm, fft as F
lib = _lib.load()
sizes = [int(a) for a in sys.argv[1:]] or [12, 13, 16, 18, 20, 22, 24]
for k in sizes:
    d = cm.synth_scalars_device(0, 1 << k, seed=5)
    for wave in (1, 0):
        lib.tune(_lib.TUNE_NTT_WAVE, wave)
        F.fft_device(d, k)
        lib.check(lib.c.mira_set_timing(1))
        acc, walls = {}, []
        for _ in range(7):
            t0 = time.perf_counter(); F.fft_device(d, k); walls.append((time.perf_counter() - t0) * 1e3)
            for name, ms in lib.timings():
                acc.setdefault(name, []).append(ms)
        lib.check(lib.c.mira_set_timing(0))
        med = {a: round(sorted(b)[len(b) // 2], 4) for a, b in acc.items()}
        print(f"k={k} wave={wave} wall {sorted(walls)[3]:.3f} ms kernels {round(sum(v for a, v in med.items() if a.startswith('ntt_')), 4)} ms {med}", flush=True)
    lib.tune(_lib.TUNE_NTT_WAVE, -1)
    lib.free(d)
