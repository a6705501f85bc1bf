R=$GRAFT_REPO_ROOT
T=r03c
cd $R
{ echo "== wave-specialised kernel (product: persistent form, per-image averages)"; python tools/micro/conv_stamp_bench.py bf16x6; python tools/micro/conv_stamp_bench.py bf16x3;
  echo "== strip kernel (exorl_gemm_tune bit 1073741824)"; python tools/micro/conv_stamp_bench.py bf16x6 --strip; python tools/micro/conv_stamp_bench.py bf16x3 --strip; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_phase_stamps.txt
{ echo "exorl_gemm_tune 0 = product; 8388608 = one image per workgroup (no persistent walk); 16 = tile kernel for the first layer's forward; 64 = tile weight-gradient kernel; 1073741824 = strip forward/dgrad kernel; 1073741904 = all three round-3a kernels";
  bash tools/micro/ws_ab.sh 0 8388608 16 64 1073741824 1073741904; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_kernels_ab.txt
cat gpurun_out/${T}_conv_kernels_ab.txt
bash tools/run_pixel_prof.sh $T bf16x6 bf16x3 | grep "update()/s"
