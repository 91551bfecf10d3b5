set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_k_reduce_phases.txt
: > $O
for L in "" tools/_variants/notree.so tools/_variants/nophasea.so; do
  for T in "" "14=0,13=2" "14=0,13=1" "14=1,13=2" "14=1,13=1"; do
    echo "LIB=$L TUNE=$T" >> $O
    MIRA_PROBE_LIB=$L TUNE=$T python tools/width_stages.py 131072 0 13 16 2>&1 | grep -v amdgpu | sed 's/.*accumulate/acc/' >> $O
  done
done
cat $O
