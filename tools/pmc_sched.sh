#!/bin/bash
# tools/pmc_sched.sh <tag> <scheduler> [spp] [scene] [options...] : SQ / LDS / traffic counters of one scheduler
# configuration on one scene (tools/sched_bench.py: disney | config3 | config4 | config5), separate passes; the
# summary carries the run's own JSON line (scene, spp, kernel, ms per launch, rays) beside the counters.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 python3 "$R/tools/sched_bench.py" "$@" steps=1 > "$OUT/run.log" 2>&1 || { tail -3 "$OUT/run.log"; exit 1; }
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$R/tools/sched_bench.py" "$@" steps=1 > "$OUT/pmc$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$OUT/pmc$i.log"; exit 1; }
done
python3 "$R/tools/pmc_summary.py" "$OUT" "$OUT/run.log" > "$OUT/summary.json"; python3 - "$OUT/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['dispatch']); print({n: round(v,4) for n,v in d['derived'].items()}); print(d.get('run'))
PY
