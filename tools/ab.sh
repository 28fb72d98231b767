#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... : bench each build of libvimg_hip (2 steps, no CPU leg)
for lib in "$@"; do
  export VIMG_HIP_LIB=$lib
  echo -n "$lib  "
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms')"
done
