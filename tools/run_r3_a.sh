# Round-3 evidence, part A (one gpurun call): default bench, rocprofv3 kernel stats of the default bench command, per-step kernel summary (eager),
# GEMM stamps, DP rehearsals (weak + strong), CQL roofline line + kernel summary.     usage: bash tools/run_r3_a.sh <tag>  -> gpurun_out/<tag>_*
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-r03}
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_bench.json
timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 100 --warmup 20 > gpurun_out/${T}_rehearse_weak.json 2> gpurun_out/${T}_rehearse_weak.err; echo "rehearse weak rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --rehearse --scaling strong --steps 100 --warmup 20 > gpurun_out/${T}_rehearse_strong.json 2> gpurun_out/${T}_rehearse_strong.err; echo "rehearse strong rc=$?"
python bench.py --config 5 --scaling weak --no-other-modes --no-roofline > gpurun_out/${T}_bench_config5_b512.json 2>/dev/null; cut -c1-400 gpurun_out/${T}_bench_config5_b512.json
python tools/micro/stamp_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_gemm_stamps.txt
python tools/micro/ws_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_gemm_schedule_ab.txt
cat gpurun_out/${T}_gemm_schedule_ab.txt
python tools/micro/offline_bench.py cql 78 12 1024 bf16x3 --roofline 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_cql_roofline.txt
cat gpurun_out/${T}_cql_roofline.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_default -o d -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_prof_default.log 2>&1
BARGS="--graph 0 --no-cpu-baseline --no-roofline --no-other-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o k -- python3 $R/bench.py $BARGS --steps 200 --warmup 20 > $R/gpurun_out/${T}_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof_cql -o c -- python3 $R/tools/micro/offline_bench.py cql 78 12 1024 bf16x3 --eager > $R/gpurun_out/${T}_prof_cql.log 2>&1
cd $R
cp gpurun_out/${T}_prof_default/d_kernel_stats.csv gpurun_out/${T}_rocprofv3_kernel_stats_default_bench.csv
python tools/prof_summary.py gpurun_out/${T}_prof/k_kernel_trace.csv 220 > gpurun_out/${T}_kernel_summary_bf16x3.txt
python tools/prof_summary.py gpurun_out/${T}_prof_cql/c_kernel_trace.csv 120 > gpurun_out/${T}_kernel_summary_cql_bf16x3.txt
rm -f gpurun_out/${T}_prof/k_kernel_trace.csv gpurun_out/${T}_prof_default/d_kernel_trace.csv gpurun_out/${T}_prof_cql/c_kernel_trace.csv
head -50 gpurun_out/${T}_kernel_summary_bf16x3.txt
head -30 gpurun_out/${T}_kernel_summary_cql_bf16x3.txt
