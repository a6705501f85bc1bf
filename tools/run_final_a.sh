# Round-end evidence, part A (fits one gpurun call): GPU suite, smoke, DP rehearsal, default bench, rocprofv3 kernel stats of the default bench
# command, per-step kernel summary, GEMM stamps.     usage: bash tools/run_final_a.sh <tag>  -> gpurun_out/<tag>_*
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r02}
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gpu_tests.log 2>&1 || { tail -30 gpurun_out/${T}_gpu_tests.log; exit 1; }
tail -2 gpurun_out/${T}_gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1 || { tail -20 gpurun_out/${T}_smoke.log; exit 1; }
tail -1 gpurun_out/${T}_smoke.log
timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 100 --warmup 20 > gpurun_out/${T}_rehearse.json 2> gpurun_out/${T}_rehearse.err; echo "rehearse rc=$?"
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_bench.json
python tools/micro/stamp_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_gemm_stamps.txt
python tools/micro/ws_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_gemm_schedule_ab.txt
cat gpurun_out/${T}_gemm_stamps.txt gpurun_out/${T}_gemm_schedule_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_default -o d -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_prof_default.log 2>&1
BARGS="--graph 0 --no-cpu-baseline --no-roofline --no-other-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o k -- python3 $R/bench.py $BARGS --steps 200 --warmup 20 > $R/gpurun_out/${T}_prof.log 2>&1
cd $R
cp gpurun_out/${T}_prof_default/d_kernel_stats.csv gpurun_out/${T}_rocprofv3_kernel_stats_default_bench.csv
python tools/prof_summary.py gpurun_out/${T}_prof/k_kernel_trace.csv 220 > gpurun_out/${T}_kernel_summary_bf16x3.txt
rm -f gpurun_out/${T}_prof/k_kernel_trace.csv gpurun_out/${T}_prof_default/d_kernel_trace.csv
head -30 gpurun_out/${T}_kernel_summary_bf16x3.txt
