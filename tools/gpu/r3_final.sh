# the driver's round-end sequence on a GPU box: smoke, pytest -m gpu, default bench (tools/final_check.sh's successor for round 3)
cd $GRAFT_REPO_ROOT
( time timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/final_smoke.log
( time timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/final_tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/final_tests.log
( time timeout -k 10 900 python bench.py --steps 20 --warmup 2 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err ); echo "bench rc=$?"; tail -2 gpurun_out/final_bench.err
