# A/B of the 32 -> 32 convolution kernels inside a config-4 Proto update: exorl_gemm_tune bits (default wave-specialised; 64 solo; 1073741824 strip)
for t in ${@:-0 64 1073741824}; do echo "tune $t"; EXORL_GEMM_TUNE=$t python tools/micro/pixel_bench.py 1024 proto bf16x6 2>&1 | grep "update()/s"; EXORL_GEMM_TUNE=$t python tools/micro/pixel_bench.py 1024 proto bf16x3 2>&1 | grep "update()/s"; done
