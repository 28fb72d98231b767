#!/bin/bash
# tools/sweep.sh VAR v1 v2 ... : bench (2 steps, no CPU leg) for each value of an env knob
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  echo -n "$VAR=$v  "
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms')"
done
