# first GPU pass of round 2: full GPU suite (without -x so every new test reports), smoke, DP rehearsal, default bench
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r2a}
timeout -k 10 900 python -m pytest tests -q -m gpu -rA --durations=15 > gpurun_out/${T}_gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|\[delta parity\]|\[fast vs|\[parity margin\]|^FAILED|^ERROR" gpurun_out/${T}_gpu_tests.log | tail -120
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${T}_smoke.log
timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 100 --warmup 20 > gpurun_out/${T}_rehearse.json 2> gpurun_out/${T}_rehearse.err; echo "rehearse rc=$?"; cat gpurun_out/${T}_rehearse.json; tail -3 gpurun_out/${T}_rehearse.err
timeout -k 10 400 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; cat gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
