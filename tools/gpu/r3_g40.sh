cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_c_harness_gpu.py -m gpu -q -x > gpurun_out/g40_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g40_tests.log; tail -12 gpurun_out/g40_tests.log
