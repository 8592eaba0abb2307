cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
rm -f gpurun_out/ecd2_observed.json
timeout -k 10 900 python -m pytest tests/test_ecd2_integration.py -m gpu -q > gpurun_out/g12_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g12_tests.log
tail -30 gpurun_out/g12_tests.log
