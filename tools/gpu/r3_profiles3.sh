# Round-3 profiles, third set: the config-3 stream on the sessions' default (layered) schedule, timed from C, one rocprofv3 run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export QLDPC_CODE_CACHE=/tmp/qcc; mkdir -p $QLDPC_CODE_CACHE
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r03c; mkdir -p $O
$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 -p > $O/config3_layered_unprofiled.json 2>/dev/null
$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 -p -f > $O/config3_flooding_unprofiled.json 2>/dev/null
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/config3 -o config3 --output-format csv -- $GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream -b 512 -r 5 > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete
ls $O $O/config3; tail -1 $O/config3_layered_unprofiled.json | cut -c1-300
