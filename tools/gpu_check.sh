#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [test] [bench] [kt] [pmc]
# Writes everything under gpurun_out/<tag>/.  Steps are joined so that a failure stops the chain.
set -o pipefail
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    test)
      timeout -k 10 600 python -m pytest "$R/tests" -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
      rc=$?; tail -4 "$OUT/pytest_gpu.log"; [ $rc -eq 0 ] || exit $rc ;;
    bench)
      timeout -k 10 400 python "$R/bench.py" --steps 3 --warmup 1 > "$OUT/bench.log" 2>&1
      rc=$?; tail -1 "$OUT/bench.log" | cut -c1-330; [ $rc -eq 0 ] || exit $rc ;;
    benchq)
      timeout -k 10 400 python "$R/bench.py" --steps 2 --warmup 1 --no-cpu > "$OUT/bench.log" 2>&1
      rc=$?; tail -1 "$OUT/bench.log" | cut -c1-330; [ $rc -eq 0 ] || exit $rc ;;
    kt)
      (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$OUT/kt.log" 2>&1)
      rc=$?; cat "$OUT"/kt/*/*_kernel_stats.csv | head -4; [ $rc -eq 0 ] || exit $rc ;;
    pmc)
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc1" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc1.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc2" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc2.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 --output-format csv -d "$OUT/pmc3" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc3.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc4" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc4.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d "$OUT/pmc5" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc5.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH --output-format csv -d "$OUT/pmc6" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 512 > "$OUT/pmc6.log" 2>&1)
      rc=$?; [ $rc -eq 0 ] || exit $rc
      python3 "$R/tools/pmc_summary.py" "$OUT" "$OUT/pmc1.log" > "$OUT/pmc_summary.json" ;;
  esac
done
