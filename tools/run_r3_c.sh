# Round-3 evidence, part C (after the wave-specialised convolution kernels): phase stamps of the forward convolution (both kernels), kernel
# summaries of a config-4 Proto update per precision mode, and the update rates of every pixel agent.     usage: bash tools/run_r3_c.sh <tag>
R=$GRAFT_REPO_ROOT
T=${1:-r03c}
cd $R
{ echo "== wave-specialised kernel (product)"; python tools/micro/conv_stamp_bench.py bf16x6; python tools/micro/conv_stamp_bench.py bf16x3;
  echo "== strip kernel (exorl_gemm_tune bit 1073741824)"; python tools/micro/conv_stamp_bench.py bf16x6 --strip; python tools/micro/conv_stamp_bench.py bf16x3 --strip; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_phase_stamps.txt
cat gpurun_out/${T}_conv_phase_stamps.txt
{ echo "exorl_gemm_tune 0 = product (wave-specialised forward/dgrad and weight-gradient kernels); 64 = tile weight-gradient kernel; 1073741824 = strip forward/dgrad kernel; 1073741888 = both round-3a kernels";
  bash tools/micro/ws_ab.sh 0 64 1073741824 1073741888; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_kernels_ab.txt
cat gpurun_out/${T}_conv_kernels_ab.txt
bash tools/run_pixel_prof.sh $T bf16x6 bf16x3 | grep "update()/s"
cd $R
rm -f gpurun_out/${T}_pixel_agents.txt
for k in ddpg proto rnd icm icm_apt disagreement diayn aps smm; do
  for p in fp32 bf16x6 bf16x3; do
    timeout -k 10 300 python tools/micro/pixel_bench.py 1024 $k $p 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/${T}_pixel_agents.txt
  done
done
cat gpurun_out/${T}_pixel_agents.txt
python tools/micro/act_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_act_latency.txt
cat gpurun_out/${T}_act_latency.txt
