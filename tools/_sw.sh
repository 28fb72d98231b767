run() { timeout -k 10 200 python tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], 'tw', d['tile_world'], d['opts'], min(d['ms']), d['mrays_per_s'])"; }
run auto 512 disney
for SG in 8 24 32 64; do run auto 512 disney pool_segments=$SG; done
run auto 512 disney pool_gbreak=48
run auto 512 disney pool_vbatch=56
run auto 512 disney pool_refill=24
