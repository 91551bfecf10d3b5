set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_fold_step.py::test_fold_step_k17_schedule tests/test_gpu_fold_cache.py -x -q -m gpu > gpurun_out/r04_p_tests.txt 2>&1 || { tail -30 gpurun_out/r04_p_tests.txt; exit 1; }
tail -1 gpurun_out/r04_p_tests.txt
O=gpurun_out/r04_p_trials.txt
: > $O
python tools/witness_stage_probe.py 2>&1 | grep "tables=0" | sed 's/.*wall/wall/' >> $O
TUNE="18=0" python tools/witness_stage_probe.py 2>&1 | grep "tables=0" | sed 's/.*wall/wall (no trials)/' >> $O
python tools/batch_probe.py 2>&1 | grep "c=0" >> $O
TUNE="18=0" python tools/batch_probe.py 2>&1 | grep "c=0" | sed 's/^/(no trials) /' >> $O
python tools/plan_calibrate.py 2>&1 | grep -v amdgpu | awk '{print $1, $2, $3}' >> $O
cat $O | cut -c1-240
