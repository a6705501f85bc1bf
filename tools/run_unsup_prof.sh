# rocprofv3 kernel summary of one reward-free agent's update on states (tools/micro/unsup_bench.py).   usage: bash tools/run_unsup_prof.sh <tag> <kind> [...]
R=$GRAFT_REPO_ROOT
T=$1; shift
cd /tmp && export TMPDIR=/tmp
for K in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_un_$K -o p -- python3 $R/tools/micro/unsup_bench.py $K --precision bf16x3 > $R/gpurun_out/${T}_un_$K.log 2>&1
  grep "update()/s" $R/gpurun_out/${T}_un_$K.log
  python3 $R/tools/prof_summary.py $R/gpurun_out/${T}_un_$K/p_kernel_trace.csv 1 > $R/gpurun_out/${T}_kernel_summary_${K}_states_bf16x3.txt 2>&1
  rm -f $R/gpurun_out/${T}_un_$K/p_kernel_trace.csv
  head -16 $R/gpurun_out/${T}_kernel_summary_${K}_states_bf16x3.txt | cut -c1-120
done
