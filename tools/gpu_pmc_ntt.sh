#!/bin/bash
# Development: PMC passes over the 2^24 NTT (k_ntt_lines).  usage: tools/gpu_pmc_ntt.sh <tag> [lib]
tag=${1:-x}
export TMPDIR=/tmp
[ -n "$2" ] && export MIRA_PROBE_LIB=$2
out=$PWD/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmcntt_${tag}_$i -o p -- python3 tools/ntt_probe.py 24 > $out/pmcntt_${tag}_$i.log 2>&1 || { echo "set $i failed"; tail -3 $out/pmcntt_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/pmcntt_${tag}_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_ntt_lines" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, "launches", len(v), "avg", sum(v) / len(v))
PY
