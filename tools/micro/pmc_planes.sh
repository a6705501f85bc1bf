# L2 hit rate and fabric traffic of one grouped split-bf16 launch, old vs new kernels (rocprofv3 --pmc, separate passes per counter set)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 262144 0; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-20)
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcp_${v}_$tag -o p -- python3 $R/tools/micro/planes_one.py $v 4 > /dev/null 2>&1
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/pmcp_${v}_$tag/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm16' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    print('variant $v', k, c, 'mean per launch %.4g' % (sum(v) / len(v)), 'n', len(v))
PY
    rm -rf $R/gpurun_out/pmcp_${v}_$tag
  done
done
