#!/bin/bash
# bench.py's train_step (render forward + backward at batch 8) under the three settings of CNERF_SCATTER
mkdir -p gpurun_out/r3
for n in default coarse chain; do
  if [ "$n" = default ]; then unset CNERF_SCATTER; else export CNERF_SCATTER=$n; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-fast-path --no-gan-step --steps 3 --warmup 1 > gpurun_out/r3/ab_$n.json 2>/dev/null
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r3/ab_$n.json').read().strip().splitlines()[-1])
print('$n', d['train_step']['fp16x3_forward_fp16_backward']['fwd_bwd_ms'])"
done
