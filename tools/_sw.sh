run() { timeout -k 10 200 python tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], 'tw', d['tile_world'], d['opts'], min(d['ms']), d['mrays_per_s'])"; }
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run auto 512 disney
for tw in 2 3 4 8; do run auto 512 disney tile_world=$tw; done
for sc in config3 config4 config5; do run auto 32 $sc; done
for tw in 2 4 8; do run auto 64 config4 tile_world=$tw;  run auto 64 config5 tile_world=$tw; done
