cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
rm -f gpurun_out/ecd2_observed.json
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_gpu.py > gpurun_out/g10_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g10_tests.log
tail -15 gpurun_out/g10_tests.log
unset ECD2_RECORD
( time timeout -k 10 900 python bench.py --steps 5 --warmup 1 > gpurun_out/g10_bench.json 2> gpurun_out/g10_bench.err ) 2>&1 | tail -4; echo "bench rc=$?"
tail -3 gpurun_out/g10_bench.err
