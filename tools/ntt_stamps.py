"""Development probe: phase stamps of k_ntt_wave (a -DNTTW_PROBE_STAMPS build, tools/build_probe_variants.sh): how the
workgroups that share a CU are phased against each other, and how long each phase of an iteration takes.

usage: MIRA_PROBE_LIB=tools/_variants/ntt_stamps.so python tools/ntt_stamps.py [k]
Stamps are s_memrealtime ticks (10 ns).  The buffer holds the LAST launch: pass 3 of the transform.
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mira_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["MIRA_PROBE_LIB"])
from mira_amd import commitment as cm, fft as F
lib = _lib.load()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
d = cm.synth_scalars_device(0, 1 << k, seed=5)
for _ in range(int(os.environ.get("MIRA_PROBE_WARM", "30"))):
    F.fft_device(d, k)
ITERS, SLOTS, WGS = 48, 8, 1024
lib.c.mira_debug_ntt_stamps_clear()
F.fft_device(d, k)
buf = np.zeros(WGS * (ITERS * SLOTS + 1), dtype=np.uint64)
assert lib.c.mira_debug_ntt_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)) == 0
clk = np.zeros(4, dtype=np.uint64)
if hasattr(lib.c, "mira_debug_ntt_clk") and lib.c.mira_debug_ntt_clk(clk.ctypes.data_as(ctypes.c_void_p)) == 0:
    dc, dt = int(clk[2]) - int(clk[0]), (int(clk[3]) - int(clk[1])) / 100.0
    print("workgroup 0 of the last launch: %d s_memtime ticks in %.1f us = %.1f MHz" % (dc, dt, dc / dt if dt else 0))
buf = buf.reshape(WGS, ITERS * SLOTS + 1)
hw = buf[:, 0]
st = buf[:, 1:].reshape(WGS, ITERS, SLOTS).astype(np.int64)
live = st[:, 0, 0] > 0
nwg = int(live.sum())
t0 = st[live][:, 0, 0].min()
st[st < t0] = 0          # iterations of an earlier launch (the passes in front of the last one) that this launch did not reach
print("workgroups with stamps:", nwg, " start spread (us): %.2f" % ((st[live][:, 0, 0].max() - t0) / 100.0))
names = ["fill+barrier", "take+barrier", "rounds", "barrier", "give(finish)", "barrier", "drain+barrier"]
its = (st[live][:, :, 7] > 0).sum(axis=1)
print("iterations per workgroup: min %d max %d" % (its.min(), its.max()))
dur = []
for w in np.nonzero(live)[0]:
    for it in range(ITERS):
        if st[w, it, 7] > 0 and (st[w, it, :] > 0).all():
            dur.append(np.diff(st[w, it, :]))
dur = np.array(dur) / 100.0
print("phase durations, us (mean / median / p90) over %d iterations:" % len(dur))
for i, n in enumerate(names):
    print("  %-14s %6.2f %6.2f %6.2f" % (n, dur[:, i].mean(), np.median(dur[:, i]), np.percentile(dur[:, i], 90)))
print("  iteration      %6.2f" % dur.sum(axis=1).mean())
# workgroups per CU: key = (xcc, se, sh, cu)
key = {}
for w in np.nonzero(live)[0]:
    h = int(hw[w]) & 0xFFFFFFFF
    x = int(hw[w]) >> 32
    cu = (x & 0xF, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF)
    key.setdefault(cu, []).append((w, h & 15, (h >> 4) & 3))
sizes = {}
for cu, v in key.items():
    sizes[len(v)] = sizes.get(len(v), 0) + 1
print("CUs by number of workgroups seen:", sizes, " distinct CUs:", len(key))
# how many workgroups of a CU are in an arithmetic phase (rounds or give) at a time
occ = np.zeros(8)
tot = 0
for cu, v in key.items():
    ws = [w for w, _, _ in v]
    lo = max(st[w, 1, 0] for w in ws)
    hi = min(st[w, its_w - 2, 7] for w, its_w in ((w, int((st[w, :, 7] > 0).sum())) for w in ws))
    if hi <= lo: continue
    grid = np.arange(lo, hi, 5)
    cnt = np.zeros(len(grid), dtype=int)
    for w in ws:
        for it in range(ITERS):
            if st[w, it, 7] == 0: break
            for a, b in ((2, 3), (4, 5)):
                cnt += (grid >= st[w, it, a]) & (grid < st[w, it, b])
    for c in range(8): occ[c] += (cnt == c).sum()
    tot += len(grid)
print("fraction of time with k workgroups of a CU in an arithmetic phase:", {c: round(float(occ[c] / tot), 3) for c in range(5)})
# by wave slot: iterations done, mean iteration time, when the workgroup finished
t_end = max(st[w, int((st[w, :, 7] > 0).sum()) - 1, 7] for w in np.nonzero(live)[0])
print("launch span: %.1f us" % ((t_end - t0) / 100.0))
by_slot = {}
for cu, v in key.items():
    for w, slot, simd in v:
        n = int((st[w, :, 7] > 0).sum())
        by_slot.setdefault(slot, []).append((n, (st[w, n - 1, 7] - st[w, 0, 0]) / 100.0 / n, (st[w, n - 1, 7] - t0) / 100.0))
for slot in sorted(by_slot):
    a = np.array(by_slot[slot])
    print("  wave slot %d: %4d workgroups, iterations %.1f (min %d max %d), us per iteration %.1f, finished at %.1f us (min %.1f max %.1f)" % (
        slot, len(a), a[:, 0].mean(), a[:, 0].min(), a[:, 0].max(), a[:, 1].mean(), a[:, 2].mean(), a[:, 2].min(), a[:, 2].max()))
fin = np.array(sorted((st[w, int((st[w, :, 7] > 0).sum()) - 1, 7] - t0) / 100.0 for w in np.nonzero(live)[0]))
print("workgroups finished at (us): first %.1f, 10%% %.1f, median %.1f, 90%% %.1f, last %.1f" % (fin[0], fin[len(fin) // 10], fin[len(fin) // 2], fin[9 * len(fin) // 10], fin[-1]))
# timeline of one CU
cu, v = sorted(key.items())[len(key) // 2]
print("CU", cu, "workgroups (blockIdx, wave slot, simd):", v)
for w, _, _ in v:
    print("  wg %4d:" % w, " | ".join("%6.1f-%6.1f/%6.1f-%6.1f" % ((st[w, it, 0] - t0) / 100, (st[w, it, 1] - t0) / 100, (st[w, it, 2] - t0) / 100, (st[w, it, 7] - t0) / 100) for it in range(4)))
