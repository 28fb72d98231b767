run() { timeout -k 10 200 python tools/sched_bench.py "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['scene'], d['kernel'], 'tw', d['tile_world'], d['opts'], min(d['ms']), d['mrays_per_s'])"; }
for sc in config4 config5; do
  run auto 128 $sc
  run auto 128 $sc tile_world=2
  for P in 48 64 96; do run pool4g 128 $sc tile_world=2 pool_slots=$P; done
  run auto 128 $sc tile_world=4
  for P in 24 32 48 64; do run pool4g 128 $sc tile_world=4 pool_slots=$P; done
  run auto 128 $sc tile_world=8
  for P in 16 24 32; do run pool4g 128 $sc tile_world=8 pool_slots=$P; done
done
