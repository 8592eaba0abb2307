set -x
cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
rm -f gpurun_out/ecd2_observed.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/g1_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g1_tests.log
tail -5 gpurun_out/g1_tests.log
for lanes in 1 4; do for b in 64 256; do
QLDPC_RECON_LANES=$lanes timeout -k 10 120 ./qcrypto-ldpc_amd/host/qldpc_stream -b $b -r 5 >> gpurun_out/g1_stream.log 2>&1; echo "lanes=$lanes b=$b rc=$?" >> gpurun_out/g1_stream.log
done; done
QLDPC_RECON_LANES=4 timeout -k 10 120 ./qcrypto-ldpc_amd/host/qldpc_stream -b 256 -r 3 -p >> gpurun_out/g1_stream.log 2>&1
cat gpurun_out/g1_stream.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-fp16 --no-int8 --no-config5 --no-cpu > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err; echo "bench rc=$?"
