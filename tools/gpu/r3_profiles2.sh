# Round-3 profiles, second set: the layered sweeps on the compressed check state (config 5 at 64 frames; the headline code on the layered schedule).
# One rocprofv3 run per workload, the program directly after `--`; counters in their own passes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r03b; mkdir -p $O
B="python3 $GRAFT_REPO_ROOT/bench.py"
C5="--steps 2 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-config3 --no-cpu --no-fer-deep --config5-frames 64"
L2="--steps 5 --warmup 1 --no-early --no-fp16 --no-int8 --no-config3 --no-config5 --no-cpu --no-fer-deep"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/config5 -o config5 --output-format csv -- $B $C5 > $O/config5.json 2> $O/config5.err; echo "config5 rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/config5_fetch -o f --output-format csv -- $B $C5 > /dev/null 2> $O/config5_fetch.err; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/config5_write -o w --output-format csv -- $B $C5 > /dev/null 2> $O/config5_write.err; echo "write rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/layered2 -o layered2 --output-format csv -- $B $L2 > $O/layered2.json 2> $O/layered2.err; echo "layered2 rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/config5_fetch $O/config5_write $O/r03_config5_cst_pmc_hbm_traffic.json "python3 bench.py $C5" 64 1 1000000 3599999 > $O/pmc_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O; ls $O; cat $O/pmc_summary.txt | grep -i "layer\|ballot\|synd"
# the default bench line of the same tree
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 bench.py 2> $O/bench_default.err > $O/bench_default.json; echo "bench rc=$?"
