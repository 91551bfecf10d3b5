#!/bin/bash
# Development: PMC passes over the 2^24 NTT (FETCH_SIZE and WRITE_SIZE: tools/gpu_profile_all.sh, in passes of their own --
# together they exceed the TCC counter slots and rocprofv3 aborts the process, error 38), wave-level kernel (k_ntt_wave) or, with a second
# argument 0, the workgroup-level one.  usage: tools/gpu_pmc_ntt2.sh <tag> [wave]
tag=${1:-x}
export TMPDIR=/tmp
export MIRA_PROBE_WAVE=${2:-1}
out=$PWD/gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmcntt_${tag}_$i -o p -- python3 tools/ntt_probe.py 24 > $out/pmcntt_${tag}_$i.log 2>&1 || { echo "set $i failed"; tail -3 $out/pmcntt_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$out/pmcntt_${tag}_*/")):
    f = glob.glob(d + "*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    order = {}
    for r in csv.DictReader(open(f[0])):
        if "k_ntt" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in agg.items():
        v.sort()
        npass = 3 if "${2:-1}" != "0" else 2
        tail = v[-npass * 4:]                         # the last four transforms
        per = [sum(x[1] for x in tail[p::npass]) / 4 for p in range(npass)]
        print(k, "per pass", [round(x) for x in per])
PY
