cd $GRAFT_REPO_ROOT
export ECD2_RECORD=1
timeout -k 10 900 python -m pytest tests/test_ecd2_integration.py -m gpu -q -k "malformed or refused or fallback or without_a_plan" > gpurun_out/g13_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g13_tests.log
tail -30 gpurun_out/g13_tests.log
