set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py tests/test_gpu_lookup_fold.py tests/test_gpu_fold_step.py::test_fold_step_k17_schedule -x -q -m gpu > gpurun_out/r04_h_tests.txt 2>&1 || { tail -30 gpurun_out/r04_h_tests.txt; exit 1; }
tail -1 gpurun_out/r04_h_tests.txt
python tools/glv_probe.py --calibrate > gpurun_out/r04_h_glv_calibrate.txt 2>&1
python tools/glv_probe.py 12 15 17 19 20 22 > gpurun_out/r04_h_glv_compare.txt 2>&1
grep -v amdgpu gpurun_out/r04_h_glv_calibrate.txt gpurun_out/r04_h_glv_compare.txt
