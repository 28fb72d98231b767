#!/bin/bash
# tools/sweep_configs.sh <tag> [spp]: schedulers on the stand-ins of configs 3-5
TAG=${1:-cfg}; SPP=${2:-32}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
run() { timeout -k 10 200 python $R/tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], d['opts'], min(d['ms']), d['mrays_per_s'])" | tee -a $OUT/sweep.txt; }
for sc in config3 config4 config5; do
  run pool $SPP $sc
  run pool4 $SPP $sc
  run pool4 $SPP $sc waves_per_simd=3
  run lane $SPP $sc
done
