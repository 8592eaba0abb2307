# Round-3 profiles: one rocprofv3 run per workload, the program directly after `--` (tools/README.md lists what each run is for).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r03; mkdir -p $O
B="python3 $GRAFT_REPO_ROOT/bench.py"
FIXED="--steps 20 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config3 --no-config5 --no-cpu --no-fer-deep"
C5="--steps 2 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/fixed50 -o fixed50 --output-format csv -- $B $FIXED > $O/fixed50.json 2> $O/fixed50.err; echo "fixed50 rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/fixed50_fetch -o f --output-format csv -- $B $FIXED > /dev/null 2> $O/fixed50_fetch.err; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/fixed50_write -o w --output-format csv -- $B $FIXED > /dev/null 2> $O/fixed50_write.err; echo "write rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/config5 -o config5 --output-format csv -- $B $C5 > $O/config5.json 2> $O/config5.err; echo "config5 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/config3 -o config3 --output-format csv -- $GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 256 -r 5 > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/edge -o edge --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/edge_latency.py > $O/edge.txt 2> $O/edge.err; echo "edge rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/fixed50_fetch $O/fixed50_write $O/r03_fixed50_pmc_hbm_traffic.json "python3 bench.py $FIXED" 4096 1 65536 235925 > $O/pmc_summary.txt 2>&1
# the traces are large: keep the stats, drop the per-dispatch rows except config 3's (the concurrency picture)
find $O -name "*kernel_trace.csv" ! -path "*config3*" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O; ls $O
