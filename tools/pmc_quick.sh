#!/bin/bash
# tools/pmc_quick.sh <tag> : one SQ pass on bench.py --spp 64 with the current environment
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc1" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 64 > "$OUT/pmc1.log" 2>&1)
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc2" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu --spp 64 > "$OUT/pmc2.log" 2>&1)
python3 "$R/tools/pmc_summary.py" "$OUT" | python3 -c "
import json,sys
d=json.load(sys.stdin); c=d['counters']
print(d['dispatch']); print(d['derived'])
for k in ['SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM','SQ_INSTS_BRANCH','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES']: print('%-24s %.4g'%(k,c.get(k,0)))
"
