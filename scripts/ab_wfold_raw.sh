#!/bin/bash
# Which reduction does the weight-folded fp32 epilogue need in front of v_sin_f32?  Accuracy micro-benchmark, then the bench with
# -DCNERF_F32_WFOLD_REDUCE=0/1/2 (none / u - rint(u) / v_fract; kernel time and the oracle check of the timed image).  Usage (on the GPU box): bash scripts/ab_wfold_raw.sh
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/vsin_raw_range scripts/ubench/vsin_raw_range.hip && /tmp/vsin_raw_range
for flags in "-DCNERF_F32_WFOLD_REDUCE=0" "-DCNERF_F32_WFOLD_REDUCE=1" ""; do
    rm -f conditioned-nerf-gan_amd/csrc/field_kernel.o
    CNERF_EXTRA_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
    echo "== field_kernel.hip flags: '$flags'"
    python bench.py --no-train-step 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, c = d['roofline'], d['check']
print('   %.2f ms per launch (%.3f of peak); rgb_sigma err %.3e / %.3e, pixels %.3e, pass %s' % (r['avg_launch_ms'], r['frac'], c['rgb_sigma_err'], c['fine_rgb_sigma_err'], c['pixels_err_forced'], c['pass']))"
done
