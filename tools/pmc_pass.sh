# usage (GPU box): bash tools/pmc_pass.sh <tag> <counter> [<counter> ...]   -> gpurun_out/<tag>_pmc_summary.txt (mean per dispatch)
set -e
R=$GRAFT_REPO_ROOT
T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${T}_pmc -o p -- python3 $R/bench.py --graph 0 --no-cpu-baseline --no-roofline --no-other-modes --steps 25 --warmup 5 > $R/gpurun_out/${T}_pmc.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/${T}_pmc/p_counter_collection.csv > gpurun_out/${T}_pmc_summary.txt
rm -f gpurun_out/${T}_pmc/*.csv
cat gpurun_out/${T}_pmc_summary.txt
