#!/bin/bash
TAG=${1:-shards2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
run() { timeout -k 10 200 python $R/tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['tile_world'], d['kernel'], d['opts'], min(d['ms']), d['mrays_per_s'])" | tee -a $OUT/sweep.txt; }
run pool4 512 disney tile_world=2 pool_slots=128
run pool4 512 disney tile_world=2 pool_slots=80
run pool4 512 disney tile_world=2 pool_slots=96 pool_vbatch=48
for n in 4 8; do
  sl=$((n==4 ? 48 : 24))
  run pool4 512 disney tile_world=$n pool_slots=$sl pool_vbatch=32 pool_starve=8 pool_refill=8
  run pool4 512 disney tile_world=$n pool_slots=$sl pool_vbatch=16 pool_starve=8 pool_refill=8
  run pool4 512 disney tile_world=$n pool_slots=$((sl*2)) pool_vbatch=32 pool_starve=8 pool_refill=8
  run pool4 512 disney tile_world=$n pool_slots=$sl pool_vbatch=32 pool_starve=8 pool_refill=8 waves_per_simd=4
  run lane 512 disney tile_world=$n waves_per_simd=3
  run pool 512 disney tile_world=$n pool_slots=$sl pool_vbatch=32 pool_starve=8 pool_refill=8
done
