cd $GRAFT_REPO_ROOT
S=./qcrypto-ldpc_amd/host/qldpc_stream
run() { echo "== $*" >> gpurun_out/g4.log; "$@" >> gpurun_out/g4.log 2>&1; echo "rc=$?" >> gpurun_out/g4.log; }
run timeout -k 10 120 $S -b 256 -r 3 -v -A 1 -B 1
run timeout -k 10 120 $S -b 256 -r 3 -v -A 4 -B 4
run timeout -k 10 120 $S -b 64 -r 3 -v -A 4 -B 4
run timeout -k 10 120 $S -b 128 -r 3 -A 4 -B 4
run timeout -k 10 120 $S -b 512 -r 3 -A 4 -B 4
run timeout -k 10 120 $S -b 256 -r 3 -S 7
run timeout -k 10 120 $S -b 256 -r 3 -p
cat gpurun_out/g4.log
