# usage (GPU box): bash tools/prof_offline.sh <agent> <O> <A> <B> <precision>
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/off_prof_$1 -o k -- python3 $R/tools/micro/offline_bench.py $1 $2 $3 $4 $5 > $R/gpurun_out/off_prof_$1.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/off_prof_$1/k_kernel_trace.csv 550 > gpurun_out/off_$1_$5_kernel_summary.txt
rm -f gpurun_out/off_prof_$1/k_kernel_trace.csv
head -30 gpurun_out/off_$1_$5_kernel_summary.txt
