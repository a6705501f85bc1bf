# Round-end evidence run on the GPU box: full GPU suite, smoke, default bench, rocprofv3 kernel stats, reward-free and pixel rates.
set -e
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-v13}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gpu_tests.log 2>&1 || { tail -30 gpurun_out/${T}_gpu_tests.log; exit 1; }
tail -2 gpurun_out/${T}_gpu_tests.log
python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1 || { tail -20 gpurun_out/${T}_smoke.log; exit 1; }
tail -1 gpurun_out/${T}_smoke.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o k -- python3 $R/bench.py --graph 0 --no-cpu-baseline --no-roofline --no-other-modes --steps 200 --warmup 20 > $R/gpurun_out/${T}_prof.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/${T}_prof/k_kernel_trace.csv 220 > gpurun_out/${T}_kernel_summary.txt
rm -f gpurun_out/${T}_prof/k_kernel_trace.csv
python tools/micro/unsup_bench.py --precision fp32,bf16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_unsup.txt
cat gpurun_out/${T}_unsup.txt
for a in "proto fp32" "proto bf16x3" "ddpg fp32" "ddpg bf16x3"; do set -- $a; python tools/micro/pixel_bench.py 1024 $1 $2 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_pixels.txt; done
cat gpurun_out/${T}_pixels.txt
for a in "cql 78 12 1024" "td3 17 6 512" "td3 17 6 4096" "crr 24 6 1024" "bc 24 6 256 fp32,bf16x3"; do python tools/micro/offline_bench.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_offline.txt; done
cat gpurun_out/${T}_offline.txt
bash tools/prof_pixels.sh proto bf16x3 > /dev/null
bash tools/prof_unsup.sh icm_apt bf16x3 > /dev/null
bash tools/prof_unsup.sh proto bf16x3 > /dev/null
