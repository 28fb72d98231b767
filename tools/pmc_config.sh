#!/bin/bash
# tools/pmc_config.sh <tag> <scene-substring> : PMC passes for one stand-in scene (32 spp)
TAG=$1; SC=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$R/tools/bench_configs.py" 32 "$SC" > "$OUT/pmc$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$OUT/pmc$i.log"; }
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"; tail -40 "$OUT/summary.json"
