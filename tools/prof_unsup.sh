# usage (GPU box): bash tools/prof_unsup.sh <agent> <precision>  -> gpurun_out/unsup_<agent>_<precision>_kernel_summary.txt
set -e
R=$GRAFT_REPO_ROOT
AG=${1:-icm_apt}
P=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/unsup_prof_$AG -o k -- python3 $R/tools/micro/unsup_bench.py $AG --precision $P --steps 100 > $R/gpurun_out/unsup_prof_$AG.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/unsup_prof_$AG/k_kernel_trace.csv 130 > gpurun_out/unsup_${AG}_${P}_kernel_summary.txt
rm -f gpurun_out/unsup_prof_$AG/k_kernel_trace.csv
head -40 gpurun_out/unsup_${AG}_${P}_kernel_summary.txt
