set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fold_step.py::test_fold_step_k17_schedule -x -q -m gpu > gpurun_out/r04_o_tests.txt 2>&1 || { tail -30 gpurun_out/r04_o_tests.txt; exit 1; }
tail -1 gpurun_out/r04_o_tests.txt
O=gpurun_out/r04_o_medium.txt
: > $O
python tools/witness_stage_probe.py 2>&1 | grep "tables=" | sed 's/.*wall/wall/' >> $O
TUNE="17=0" python tools/width_stages.py 131072 0 8 12 13 2>&1 | grep -v amdgpu | sed 's/kind 0//' >> $O
TUNE="17=0" python tools/width_stages.py 8192 0 5 8 2>&1 | grep -v amdgpu | sed 's/kind 0//' >> $O
TUNE="17=0" python tools/width_stages.py 131072 1 8 13 2>&1 | grep -v amdgpu >> $O
python tools/fuzz_msm.py 120 >> $O 2>&1 || true
cat $O | cut -c1-260 | tail -30
