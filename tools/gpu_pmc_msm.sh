#!/bin/bash
# PMC passes over the headline bench command for k_accumulate (VALU issue, wave states).
# usage: tools/gpu_pmc_msm.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
out=$PWD/gpurun_out
B="python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmcmsm_${tag}_$i -o p -- $B > /dev/null 2> $out/pmcmsm_${tag}_$i.log || { echo "set $i failed"; tail -3 $out/pmcmsm_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/pmcmsm_${tag}_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_accumulate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:24s} launches {len(v)}  avg {sum(v) / len(v):.0f}")
PY
