#!/bin/bash
# One GPU rendering rank 0's shard (tile_world = tw) with the kernel and pool the launch policy picks:
# scene, tile_world, kernel, options, best ms of 2, Mrays/s.  Config 2 at 512 spp, the stand-ins of
# configs 3-5 at 128 spp.  Usage (GPU box): tools/policy_shards.sh <tag>
TAG=${1:-shards}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; rm -f $OUT/policy_shards.txt
run() { timeout -k 10 300 python $R/tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], 'tw', d['tile_world'], d['spp'], 'spp', d['opts'], min(d['ms']), d['mrays_per_s'])" | tee -a $OUT/policy_shards.txt; }
for tw in 1 2 3 4 8; do run auto 512 disney tile_world=$tw || exit 1; done
for sc in config3 config4 config5; do for tw in 1 2 4 8; do run auto 128 $sc tile_world=$tw || exit 1; done; done
