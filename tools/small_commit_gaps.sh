#!/bin/bash
# Development: kernel trace of 205 commits of 131 072 pairs (tools/small_commit_loop.py): per commit, the time its kernels run,
# the idle time BETWEEN consecutive kernels of the commit (what a captured hipGraph could at best remove) and the rest of the
# wall (first-launch latency, the copy back, the host epilogue).
export TMPDIR=/tmp
out=$PWD/gpurun_out
rm -rf $out/prof_small
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/prof_small -o s -- python3 tools/small_commit_loop.py > $out/prof_small.txt 2>&1
tail -1 $out/prof_small.txt
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof_small/*kernel_trace.csv") + glob.glob("$out/prof_small/*/*kernel_trace.csv")
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))), key=lambda x: x[0])
# a commit starts at every k_digits launch
starts = [i for i, r in enumerate(rows) if "k_digits" in r[2]]
commits = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])][5:]       # skip the warm-up commits
busy = gaps = span = 0.0
per_gap = {}
for c in commits:
    busy += sum(e - s for s, e, _ in c)
    span += c[-1][1] - c[0][0]
    for (s0, e0, n0), (s1, e1, n1) in zip(c, c[1:]):
        gaps += max(0, s1 - e0)
        k = n0.split("<")[0].split("(")[0][:28] + " -> " + n1.split("<")[0].split("(")[0][:28]
        per_gap.setdefault(k, []).append(max(0, s1 - e0))
n = len(commits)
period = (commits[-1][0][0] - commits[0][0][0]) / (n - 1)
print("commits %d  kernels per commit %.1f" % (n, sum(len(c) for c in commits) / n))
print("per commit: kernels busy %.1f us, gaps between kernels %.1f us, first start to last end %.1f us, period (wall) %.1f us" % (busy / n / 1e3, gaps / n / 1e3, span / n / 1e3, period / 1e3))
for k, v in per_gap.items():
    print("  gap %-62s %.2f us" % (k, sum(v) / len(v) / 1e3))
PY
