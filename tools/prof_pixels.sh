# usage (GPU box): bash tools/prof_pixels.sh <proto|ddpg>  -> gpurun_out/pixel_<kind>_kernel_summary.txt
set -e
R=$GRAFT_REPO_ROOT
K=${1:-proto}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pixel_prof_$K -o k -- python3 $R/tools/micro/pixel_bench.py 1024 $K ${2:-fp32} > $R/gpurun_out/pixel_prof_$K.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/pixel_prof_$K/k_kernel_trace.csv 13 > gpurun_out/pixel_${K}_kernel_summary.txt
rm -f gpurun_out/pixel_prof_$K/k_kernel_trace.csv
tail -1 gpurun_out/pixel_prof_$K.log
head -24 gpurun_out/pixel_${K}_kernel_summary.txt
