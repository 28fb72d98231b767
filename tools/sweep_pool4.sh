#!/bin/bash
# tools/sweep_pool4.sh <tag> : parameter sweep of the pool4 scheduler on config 2 (64 spp)
TAG=${1:-sweep}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
run() { timeout -k 10 120 python $R/tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel'], d['opts'], min(d['ms']), d['mrays_per_s'])" | tee -a $OUT/sweep.txt; }
timeout -k 10 200 python -m pytest $R/tests/test_gpu_parity.py -x -q -k "all_schedulers or pooled_and_staged" 2>&1 | tail -2
run pool4 64
run pool4 64 disney waves_per_simd=3
run pool4 64 disney waves_per_simd=3 pool_refill=32
run pool4 64 disney waves_per_simd=3 pool_classes=2
run pool4 64 disney pool_classes=2
run pool4 64 disney waves_per_simd=3 pool_slots=150
run pool4 64 disney waves_per_simd=3 pool_slots=120
run pool4 512 disney waves_per_simd=3
run pool4 512 disney
