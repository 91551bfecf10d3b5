set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_m_trace_wit -o t -- python3 tools/small_commit_loop.py 1835008 0 1 > $O/r04_m_trace_wit.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_m_trace_c13 -o t -- python3 tools/small_commit_loop.py 131072 13 > $O/r04_m_trace_c13.txt 2>&1
grep "wall per" $O/r04_m_trace_wit.txt $O/r04_m_trace_c13.txt
