cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
rm -f gpurun_out/ecd2_observed.json
timeout -k 10 1100 python -m pytest tests -m gpu -q --deselect tests/test_bench_gpu.py --durations=8 > gpurun_out/g17_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g17_tests.log
tail -40 gpurun_out/g17_tests.log
