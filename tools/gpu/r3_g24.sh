cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_recon_gpu.py -m gpu -q -x > gpurun_out/g24_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g24_tests.log; tail -15 gpurun_out/g24_tests.log
