set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fold_step.py::test_fold_step_k17_schedule -x -q -m gpu > gpurun_out/r04_l_tests.txt 2>&1 || { tail -30 gpurun_out/r04_l_tests.txt; exit 1; }
tail -1 gpurun_out/r04_l_tests.txt
O=gpurun_out/r04_l_stages.txt
: > $O
TUNE="17=0" python tools/width_stages.py 131072 0 8 12 13 16 >> $O 2>&1
TUNE="17=0" python tools/width_stages.py 131072 1 8 12 13 >> $O 2>&1
TUNE="17=0" python tools/width_stages.py 4194304 0 16 >> $O 2>&1
python tools/witness_stage_probe.py >> $O 2>&1
python tools/batch_probe.py >> $O 2>&1
grep -v amdgpu $O | cut -c1-330
