run() { timeout -k 10 200 python tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], 'tw', d['tile_world'], d['opts'], min(d['ms']), d['mrays_per_s'])"; }
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run auto 512 disney; run auto 512 disney tile_world=2; run auto 512 disney tile_world=4; run auto 512 disney tile_world=8
run auto 128 disney res=3600x1600
run auto 128 disney res=900x400
for sc in config3 config4 config5; do run auto 32 $sc; done
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | cut -c1-250
