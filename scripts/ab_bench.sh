#!/bin/bash
# A/B the field kernel under different compile-time knobs on the GPU box: scripts/ab_bench.sh "<flags A>" "<flags B>" ...
# Each variant: forced rebuild, bench.py short run (extra bench args in $BENCH_ARGS), print ms/launch and TFLOP/s.
for flags in "$@"; do
  CNERF_EXTRA_FLAGS="$flags" python conditioned-nerf-gan_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('flags=[$flags] rep$rep', 'rays/s %.3fM' % (r['value']/1e6), 'kernel ms %.2f' % r['roofline']['avg_launch_ms'], 'TF %.1f' % r['roofline']['achieved'], 'frac %.3f' % r['roofline']['frac'])"
  done
done
python conditioned-nerf-gan_amd/build.py --force > /dev/null 2>&1
