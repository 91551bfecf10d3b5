set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_d_width_stages.txt
python -m pytest tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r04_d_tests.txt 2>&1 || { tail -30 gpurun_out/r04_d_tests.txt; exit 1; }
tail -1 gpurun_out/r04_d_tests.txt
: > $O
for T in "" "16=1000000000" "13=1" "12=4"; do echo "TUNE $T" >> $O; TUNE=$T python tools/width_stages.py 131072 0 8 12 13 16 >> $O 2>&1; done
for T in "" "16=1000000000"; do echo "TUNE $T" >> $O; TUNE=$T python tools/width_stages.py 1048576 0 13 16 >> $O 2>&1;  TUNE=$T python tools/width_stages.py 4194304 0 16 >> $O 2>&1; done
grep -v amdgpu.ids $O
