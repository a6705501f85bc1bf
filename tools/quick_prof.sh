# usage (on the GPU box): bash tools/quick_prof.sh <precision> <tag>   -> gpurun_out/<tag>_kernel_summary.txt
set -e
R=$GRAFT_REPO_ROOT
P=${1:-bf16x3}
T=${2:-q}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o k -- python3 $R/bench.py --precision $P --graph 0 --no-cpu-baseline --no-roofline --no-other-modes --steps 200 --warmup 20 > $R/gpurun_out/${T}_prof.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/${T}_prof/k_kernel_trace.csv 220 > gpurun_out/${T}_kernel_summary.txt
rm -f gpurun_out/${T}_prof/k_kernel_trace.csv
head -22 gpurun_out/${T}_kernel_summary.txt
