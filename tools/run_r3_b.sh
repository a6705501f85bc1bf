# Round-3 evidence, part B: PMC passes of the bench step (FETCH / WRITE -> traffic JSON, L2 hit rate, MFMA utilisation), the other configs' update
# rates, the reward-free agents on states, pixel agents (fp32, bf16x6, bf16x3).     usage: bash tools/run_r3_b.sh <tag>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
T=${1:-r03}
BARGS="--graph 0 --no-cpu-baseline --no-roofline --no-other-modes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -o f -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/${T}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_write -o w -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/${T}_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${T}_pmc_l2 -o l -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/${T}_pmc_l2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${T}_pmc_mfma -o m -- python3 $R/bench.py $BARGS --steps 25 --warmup 5 > $R/gpurun_out/${T}_pmc_mfma.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/${T}_pmc_fetch/f_counter_collection.csv gpurun_out/${T}_pmc_write/w_counter_collection.csv gemm16p gpurun_out/${T}_pmc_traffic_bf16x3.json "eager launches of the bench step (--graph 0), 25 steps" > /dev/null
for p in fetch write l2 mfma; do python tools/pmc_summary.py gpurun_out/${T}_pmc_$p/*_counter_collection.csv > gpurun_out/${T}_pmc_${p}_summary.txt; done
rm -f gpurun_out/${T}_pmc_*/*.csv
cat gpurun_out/${T}_pmc_traffic_bf16x3.json
head -12 gpurun_out/${T}_pmc_l2_summary.txt
head -12 gpurun_out/${T}_pmc_mfma_summary.txt
for a in "cql 78 12 1024" "td3 17 6 512" "td3 17 6 4096" "crr 24 6 1024" "bc 24 6 256 fp32,bf16x3"; do python tools/micro/offline_bench.py $a 2>&1 | grep -v amdgpu.ids >> gpurun_out/${T}_offline.txt; done
cat gpurun_out/${T}_offline.txt
python tools/micro/unsup_bench.py --precision fp32,bf16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_unsup.txt
cat gpurun_out/${T}_unsup.txt
for k in ddpg proto rnd icm icm_apt disagreement diayn aps smm; do
  for p in fp32 bf16x6 bf16x3; do
    timeout -k 10 300 python tools/micro/pixel_bench.py 1024 $k $p 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/${T}_pixel_agents.txt
  done
done
cat gpurun_out/${T}_pixel_agents.txt
