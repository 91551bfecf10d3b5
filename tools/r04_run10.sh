set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_n_heavy_span.txt
: > $O
for L in "" tools/_variants/span10.so tools/_variants/span14.so; do
  echo "LIB=$L" >> $O
  MIRA_PROBE_LIB=$L python tools/witness_stage_probe.py 2>&1 | grep "tables=0" | sed 's/.*wall/wall/' >> $O
  MIRA_PROBE_LIB=$L TUNE="17=0" python tools/width_stages.py 131072 0 8 12 13 2>&1 | grep -v amdgpu | sed 's/kind 0//' >> $O
  MIRA_PROBE_LIB=$L TUNE="17=0" python tools/width_stages.py 8192 0 5 8 2>&1 | grep -v amdgpu | sed 's/kind 0//' >> $O
  MIRA_PROBE_LIB=$L TUNE="17=0" python tools/width_stages.py 131072 1 8 13 2>&1 | grep -v amdgpu >> $O
done
cat $O | cut -c1-260
