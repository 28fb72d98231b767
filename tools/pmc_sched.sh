#!/bin/bash
# tools/pmc_sched.sh <tag> <scheduler> [spp] [scene] [options...] : SQ / LDS / traffic counters of one scheduler configuration
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 "$R/tools/sched_bench.py" "$@" steps=1 > "$OUT/pmc$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$OUT/pmc$i.log"; }
done
python3 "$R/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"; python3 - "$OUT/summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d['counters']; print(d['dispatch'])
g=k.get
print({n: round(v,4) for n,v in d['derived'].items()})
if g('SQ_LDS_IDX_ACTIVE'): print('lds conflict share', g('SQ_LDS_BANK_CONFLICT')/g('SQ_LDS_IDX_ACTIVE'))
print('insts: valu %.3e salu %.3e lds %.3e vmem %.3e branch %.3e' % (g('SQ_INSTS_VALU',0), g('SQ_INSTS_SALU',0), g('SQ_INSTS_LDS',0), g('SQ_INSTS_VMEM',0), g('SQ_INSTS_BRANCH',0)))
print('fetch GB %.1f write GB %.1f (summed over the dispatches of the pass)' % (g('FETCH_SIZE',0)*1024*2/1e9, g('WRITE_SIZE',0)*1024/1e9))
for n in ('SQ_IFETCH','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_SCA','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES'): print(n, g(n))
PY
