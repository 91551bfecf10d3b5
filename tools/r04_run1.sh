set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r04_a_gpu_msm_tests.txt 2>&1 || { tail -30 gpurun_out/r04_a_gpu_msm_tests.txt; exit 1; }
tail -3 gpurun_out/r04_a_gpu_msm_tests.txt
O=gpurun_out/r04_a_width_stages.txt
: > $O
for T in "12=1" "12=2" "12=3" "12=4"; do echo "TUNE $T" >> $O; TUNE=$T python tools/width_stages.py 131072 0 8 12 13 16 >> $O 2>&1; done
for T in "12=3,13=1" "12=3,13=3" "12=3,14=0" "12=3,14=0,13=2"; do echo "TUNE $T" >> $O; TUNE=$T python tools/width_stages.py 131072 0 12 13 16 >> $O 2>&1; done
for T in "12=1" "12=3" "12=3,14=1" "12=3,13=2"; do echo "TUNE $T" >> $O; TUNE=$T python tools/width_stages.py 1048576 0 13 16 >> $O 2>&1; done
echo "2^22" >> $O; TUNE="12=3" python tools/width_stages.py 4194304 0 16 >> $O 2>&1
grep -v amdgpu.ids $O
