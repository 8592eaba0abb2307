# third session of round 3: kernel stats of the config-3 stream (C tool) and of config 5 at 64 frames on the final tree (layer records)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
O=$GRAFT_REPO_ROOT/gpurun_out/prof_s3; mkdir -p $O
$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 > $O/config3_unprofiled.json 2>/dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/config3 -o config3 --output-format csv -- $GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/config5 -o config5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-early --no-layered --no-fp16 --no-int8 --no-spa --no-config3 --no-cpu --no-fer-deep --config5-frames 64 > $O/config5_bench.json 2> $O/config5.err; echo "config5 rc=$?"
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
find $O -name "*_kernel_stats.csv" | while read f; do echo $f; head -4 $f | cut -c1-60,190-260; done
tail -1 $O/config3_unprofiled.json | cut -c1-300
