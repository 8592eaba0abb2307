cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_edge_engine_gpu.py tests/test_recon_gpu.py tests/test_syndrome_form_gpu.py -m gpu -q -x > gpurun_out/g20_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g20_tests.log; tail -3 gpurun_out/g20_tests.log
python3 tools/edge_latency.py 2>&1 | tee gpurun_out/g20_latency.txt
O=$GRAFT_REPO_ROOT/gpurun_out/prof_edge1; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o e1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/edge_trace.py run > $O/run.txt 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/edge_trace.py report $(find $O -name "*kernel_trace.csv" | head -1) | tee $O/report.txt
tail -1 $O/run.txt
find $O -name "*kernel_trace.csv" -delete
python3 tools/latency.py 2>&1 | tail -12 | tee gpurun_out/g20_session_latency.txt
