cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_edge1; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o e1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/edge_trace.py run > $O/run.txt 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/edge_trace.py report $(find $O -name "*kernel_trace.csv" | head -1) | tee $O/report.txt
tail -2 $O/run.txt
python3 tools/edge_latency.py 2>&1 | head -2
find $O -name "*kernel_trace.csv" -delete
