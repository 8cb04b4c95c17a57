#!/bin/bash
# A/B of the single-pass fp16 forward compiled for one / two waves per SIMD (register cap 512 / 256); prints the bench's fp16 path.
# Usage (on the GPU box): bash scripts/ab_h1_occ.sh
set -e
cd "$(dirname "$0")/.."
for flags in "" "-DCNERF_H3_OCC=2" "-DCNERF_H3_OCC=2 -DCNERF_H3_NOPREFETCH"; do
    rm -f conditioned-nerf-gan_amd/csrc/field_h1.o
    CNERF_H1_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
    echo "== h1 flags: '$flags'"
    python bench.py --no-train-step 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['fp16_single_pass_path']
print('   fp16 single pass: %.2f ms per launch, %.2f M rays/s' % (p['avg_launch_ms'], p['value'] / 1e6))"
done
rm -f conditioned-nerf-gan_amd/csrc/field_h1.o
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
