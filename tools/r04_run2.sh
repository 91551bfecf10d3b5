set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out
python -m pytest tests/test_gpu_graph_jit.py tests/test_gpu_msm.py -x -q -m gpu > $O/r04_b_tests.txt 2>&1 || { tail -30 $O/r04_b_tests.txt; exit 1; }
tail -2 $O/r04_b_tests.txt
python tools/batch_probe.py > $O/r04_b_batch_probe.txt 2>&1
python tools/witness_stage_probe.py > $O/r04_b_witness_probe.txt 2>&1
for c in 13 12 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_b_trace_c$c -o t -- python3 tools/small_commit_loop.py 131072 $c > $O/r04_b_trace_c$c.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_b_trace_wit -o t -- python3 tools/small_commit_loop.py 1835008 0 1 > $O/r04_b_trace_wit.txt 2>&1
grep -v amdgpu.ids $O/r04_b_batch_probe.txt $O/r04_b_witness_probe.txt
