#!/bin/bash
# tools/walk_diag.sh [spp [scheduler [key=value ...]]] : where the cycles of render_pool4_kernel's stages go, per scene
# (`make diag` first: the measurement build with counters inside the walk stage; never the product library).
# Prints, per scene, the share of wave cycles per stage, batches and slots per batch, and for the walk:
# cycles in refill / box loop / leaf rounds / retire, box trips and lanes per trip, leaf rounds and lanes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; SPP=${1:-32}; SCHED=${2:-pool4}; shift; shift
export VIMG_HIP_LIB=$R/v-img_amd/lib/diag/libvimg_hip.so VIMG_HIP_DIAG=1
for a in disney config3 config4 config5; do
  echo "== $a"
  timeout -k 10 300 python $R/tools/sched_bench.py $SCHED $SPP $a "$@" 2>&1 | grep "vimg stage\|vimg walk\|mrays" | grep -v " 0 cyc"
done
