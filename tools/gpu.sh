#!/bin/bash
# gpurun wrapper: stamps the tree's commit into exorl_amd/_build_commit.txt (the GPU box gets no .git) so that evidence files
# written there (profiles/*_pmc_traffic_*.json) can say which tree they measured.   usage: tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- exorl_amd include bench.py || c="$c-dirty"
echo "$c" > exorl_amd/_build_commit.txt
exec /usr/local/graft/bin/gpurun "$@"
