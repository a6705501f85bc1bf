R=$GRAFT_REPO_ROOT
cd $R
rm -f gpurun_out/r03c_pixel_agents_final.txt
for k in ddpg proto rnd icm icm_apt disagreement diayn aps smm; do
  for p in bf16x6 bf16x3; do
    timeout -k 10 300 python tools/micro/pixel_bench.py 1024 $k $p 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r03c_pixel_agents_final.txt
  done
done
cat gpurun_out/r03c_pixel_agents_final.txt
