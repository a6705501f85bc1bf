# GPU pass: full suite (-x), smoke, default bench
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r2}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${T}_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; cat gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
