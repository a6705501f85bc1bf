set -e
R=$GRAFT_REPO_ROOT
cd $R
python __graft_entry__.py smoke > gpurun_out/v12_smoke.log 2>&1 || { tail -20 gpurun_out/v12_smoke.log; exit 1; }
tail -1 gpurun_out/v12_smoke.log
python bench.py > gpurun_out/v12_bench.json 2> gpurun_out/v12_bench.err
cat gpurun_out/v12_bench.json
cd /tmp && export TMPDIR=/tmp
BARGS="--graph 0 --no-cpu-baseline --no-roofline --no-other-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v12_prof -o x3 -- python3 $R/bench.py $BARGS --steps 200 --warmup 20 > $R/gpurun_out/v12_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/v12_pmc_fetch -o f -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/v12_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/v12_pmc_write -o w -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/v12_pmc_write.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/v12_prof/x3_kernel_trace.csv 220 > gpurun_out/v12_kernel_summary.txt
python tools/pmc_traffic.py gpurun_out/v12_pmc_fetch/f_counter_collection.csv gpurun_out/v12_pmc_write/w_counter_collection.csv gemm16x3 gpurun_out/v12_pmc_traffic.json > /dev/null
python tools/pmc_summary.py gpurun_out/v12_pmc_fetch/f_counter_collection.csv > gpurun_out/v12_pmc_fetch_summary.txt
python tools/pmc_summary.py gpurun_out/v12_pmc_write/w_counter_collection.csv > gpurun_out/v12_pmc_write_summary.txt
rm -f gpurun_out/v12_prof/x3_kernel_trace.csv gpurun_out/v12_pmc_fetch/*.csv gpurun_out/v12_pmc_write/*.csv
cat gpurun_out/v12_kernel_summary.txt | head -60
cat gpurun_out/v12_pmc_traffic.json
