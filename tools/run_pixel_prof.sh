# rocprofv3 kernel summary of a config-4 update (Proto, or KIND=icm / disagreement / ...) per precision mode.   usage: [KIND=icm] bash tools/run_pixel_prof.sh <tag> <precision> [...]
R=$GRAFT_REPO_ROOT
T=$1; shift
cd /tmp && export TMPDIR=/tmp
for P in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_pix_$P -o p -- python3 $R/tools/micro/pixel_bench.py 1024 ${KIND:-proto} $P > $R/gpurun_out/${T}_pix_$P.log 2>&1
  grep "update()/s" $R/gpurun_out/${T}_pix_$P.log
  python3 $R/tools/prof_summary.py $R/gpurun_out/${T}_pix_$P/p_kernel_trace.csv 13 > $R/gpurun_out/${T}_kernel_summary_${KIND:-proto}_pixels_$P.txt
  rm -f $R/gpurun_out/${T}_pix_$P/p_kernel_trace.csv
  head -24 $R/gpurun_out/${T}_kernel_summary_${KIND:-proto}_pixels_$P.txt | cut -c1-130
done
