cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
timeout -k 10 600 python -m pytest tests/test_ecd2_integration.py -m gpu -q -x -k "batched_ingest" > gpurun_out/g11_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g11_tests.log
tail -5 gpurun_out/g11_tests.log
