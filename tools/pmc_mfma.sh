# MFMA utilisation per kernel (rocprofv3 PMC, own pass): SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_LDS_BANK_CONFLICT
# usage (GPU box): bash tools/pmc_mfma.sh  -> gpurun_out/mfma_{step,pixels}_pmc_summary.txt
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/mfma_step_pmc -o p -- python3 $R/bench.py --graph 0 --no-cpu-baseline --no-roofline --no-other-modes --steps 25 --warmup 5 > $R/gpurun_out/mfma_step_pmc.log 2>&1
rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/mfma_pix_pmc -o p -- python3 $R/tools/micro/pixel_bench.py 1024 proto bf16x3 > $R/gpurun_out/mfma_pix_pmc.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/mfma_step_pmc/p_counter_collection.csv > gpurun_out/mfma_step_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/mfma_pix_pmc/p_counter_collection.csv > gpurun_out/mfma_pixels_pmc_summary.txt
rm -f gpurun_out/mfma_step_pmc/*.csv gpurun_out/mfma_pix_pmc/*.csv
head -8 gpurun_out/mfma_step_pmc_summary.txt | cut -c1-200
head -6 gpurun_out/mfma_pixels_pmc_summary.txt | cut -c1-200
