set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fold_step.py tests/test_gpu_dist.py tests/test_gpu_msm.py -x -q -m gpu -s > gpurun_out/r04_f_tests.txt 2>&1 || { tail -40 gpurun_out/r04_f_tests.txt; exit 1; }
grep -E "passed|failed|largest-size" gpurun_out/r04_f_tests.txt
