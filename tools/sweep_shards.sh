#!/bin/bash
# tools/sweep_shards.sh <tag>: one GPU renders rank 0's shard of the fixed config-2 frame for N = 1, 2, 4, 8
# with each scheduler: the time of a shard on one GPU = what each GPU of N does (T_N without the gather)
TAG=${1:-shards}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
run() { timeout -k 10 200 python $R/tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['tile_world'], d['kernel'], d['opts'], min(d['ms']), d['mrays_per_s'])" | tee -a $OUT/sweep.txt; }
for n in 2 4 8; do
  run auto 512 disney tile_world=$n
  run lane 512 disney tile_world=$n
  run pool4 512 disney tile_world=$n
  for sl in 48 96; do run pool4 512 disney tile_world=$n pool_slots=$sl; done
  run pool4 512 disney tile_world=$n pool_segments=1
  run pool4 512 disney tile_world=$n pool_segments=16
done
