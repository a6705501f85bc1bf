# usage (GPU box): bash tools/run_pixel_agents.sh <tag>  -> update()/s of every pixel agent at BASELINE config 4 shapes (gpurun_out/<tag>_pixel_agents.txt)
R=$GRAFT_REPO_ROOT
cd $R
T=${1:-px}
for k in ddpg proto rnd icm icm_apt disagreement diayn aps smm; do
  for p in fp32 bf16x3; do
    timeout -k 10 300 python tools/micro/pixel_bench.py 1024 $k $p 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/${T}_pixel_agents.txt
  done
done
cat gpurun_out/${T}_pixel_agents.txt
