#!/bin/bash
# Development: kernel trace of the k = 17 witness commits (tools/witness_stage_probe.py) -- which of the three fix-up kernels takes the time
export TMPDIR=/tmp
out=$PWD/gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_witness -o w -- python3 tools/witness_stage_probe.py > $out/prof_witness.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/prof_witness/*kernel_stats.csv") + glob.glob("$out/prof_witness/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
