# HBM-side traffic and matrix-pipe occupancy of the convolution kernels inside a config-4 Proto update (13 updates of tools/micro/pixel_bench.py),
# rocprofv3 --pmc in separate passes as MI355X_MICROARCH.md prescribes.     usage: bash tools/run_conv_pmc.sh <tag> <precision> [...]
R=$GRAFT_REPO_ROOT
T=$1; shift
cd /tmp && export TMPDIR=/tmp
for P in "$@"; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${T}_pmc_${P}_$tag -o c -- python3 $R/tools/micro/pixel_bench.py 1024 proto $P > $R/gpurun_out/${T}_pmc_${P}_$tag.log 2>&1
    f=$(find $R/gpurun_out/${T}_pmc_${P}_$tag -name 'c_counter_collection.csv' | head -1)
    { echo "== $P: $set (mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE to be doubled on gfx950)"; python3 $R/tools/pmc_summary.py $f | grep -E "^kernel|conv|relu_mask|aug_shift" ; } >> $R/gpurun_out/${T}_conv_pmc.txt
    rm -rf $R/gpurun_out/${T}_pmc_${P}_$tag
  done
done
cat $R/gpurun_out/${T}_conv_pmc.txt
