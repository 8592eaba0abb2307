cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/qcrypto-ldpc_amd/host/qldpc_stream
mkdir -p gpurun_out/prof_c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3 -o c3 --output-format csv -- $S -b 256 -r 5 > gpurun_out/g8.log 2>&1
echo rc=$? >> gpurun_out/g8.log
ls -R gpurun_out/prof_c3 | head -20 >> gpurun_out/g8.log
tail -5 gpurun_out/g8.log
