#!/bin/bash
# One gpurun call that regenerates the judged profiles: kernel-trace stats of the headline bench
# command, then FETCH_SIZE and WRITE_SIZE in their own PMC passes, then the NTT stats.
# usage: tools/gpu_profile.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
B="python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o msm -- $B > $out/prof_$tag.json 2> $out/prof_$tag.err && echo stats ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcf_$tag -o msm -- $B > /dev/null 2> $out/pmcf_$tag.err && echo fetch ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcw_$tag -o msm -- $B > /dev/null 2> $out/pmcw_$tag.err && echo write ok &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profntt_$tag -o ntt -- python3 tools/ntt_probe.py 24 > $out/profntt_$tag.txt 2>&1 && echo ntt ok
