for cfg in "bf16x3 0" "bf16x3 16384" "bf16 0" "bf16 16384" "bf16 32768"; do set -- $cfg
EXORL_GEMM_TUNE=$2 timeout -k 10 120 python bench.py --precision $1 --no-cpu-baseline --no-other-modes --steps 1000 --warmup 100 > gpurun_out/b.json 2>gpurun_out/b.err || { tail -5 gpurun_out/b.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('$1', $2, d['value'],d['ms_per_step'],d['roofline']['avg_us'])"; done
EXORL_GEMM_TUNE=16384 timeout -k 10 300 python -m pytest tests/test_gpu_agent.py -x -q -m gpu -k "full_size and (td3_bc or td3)" 2>&1 | tail -2
