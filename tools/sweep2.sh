#!/bin/bash
# tools/sweep2.sh "VAR1=a,b,c" "VAR2=x,y" ... : bench (2 steps, no CPU leg) over the cross product of env knobs
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  next=()
  for c in "${combos[@]}"; do
    IFS=, read -ra vs <<< "$vals"
    for v in "${vs[@]}"; do next+=("$c $var=$v"); done
  done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  echo -n "$c  "
  env $c timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms')"
done
