#!/bin/bash
# One gpurun call that regenerates EVERY judged profile of a round from the code in the tree:
#   1. GPU parity suite (one pytest process)
#   2. rocprofv3 --kernel-trace --stats of the headline bench command, then FETCH_SIZE and WRITE_SIZE in passes of
#      their own (together they exceed the TCC counter slots: rocprofv3 aborts the process, error 38)
#   3. the same three for the 2^24 NTT (tools/ntt_probe.py 24)
#   4. VALU / wave-state counters of k_accumulate (tools/gpu_pmc_msm.sh) and the NTT's counter sets (tools/gpu_pmc_ntt2.sh)
#   5. python tools/collect_profiles.py <tag> ...  -> profiles/<tag>_*, profiles/pmc_traffic.json (run HERE afterwards:
#      gpurun merges gpurun_out/ back, profiles/ is written in this container)
# Steps are joined with && -- a failed or killed GPU step starts no further one.
# A gpurun call is at most 20 minutes: part a = steps 1-3 and the MSM counters, part b = the NTT counters and the
# full default bench line.
# usage: tools/gpu_profile_all.sh <tag> <a|b> [skip-tests]
tag=${1:-x}
part=${2:-a}
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
B="python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu"
N="python3 tools/ntt_probe.py 24"
if [ "$part" = "b" ]; then
  bash tools/gpu_pmc_ntt2.sh $tag > $out/pmcntt_$tag.txt 2>&1 && echo "ntt pmc ok" &&
  timeout -k 10 600 python3 bench.py > $out/bench_full_$tag.json 2> $out/bench_full_$tag.err && echo "full bench ok"
  exit $?
fi
{ [ -n "$3" ] || { timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_$tag.txt 2>&1; rc=$?; tail -3 $out/pytest_$tag.txt; [ $rc -eq 0 ]; }; } &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o msm -- $B > $out/prof_$tag.json 2> $out/prof_$tag.err && echo "msm stats ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcf_$tag -o msm -- $B > /dev/null 2> $out/pmcf_$tag.err && echo "msm fetch ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcw_$tag -o msm -- $B > /dev/null 2> $out/pmcw_$tag.err && echo "msm write ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/profntt_$tag -o ntt -- $N > $out/profntt_$tag.txt 2>&1 && echo "ntt stats ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmcfntt_$tag -o ntt -- $N > /dev/null 2> $out/pmcfntt_$tag.err && echo "ntt fetch ok" &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmcwntt_$tag -o ntt -- $N > /dev/null 2> $out/pmcwntt_$tag.err && echo "ntt write ok" &&
bash tools/gpu_pmc_msm.sh $tag > $out/pmcmsm_$tag.txt 2>&1 && echo "msm valu ok"
