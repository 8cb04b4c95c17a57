# A/B of the fp16x3 forward kernel's epilogue placement on the GPU box (diagnostic): rebuilds field_h3.hip with macros,
# prints the field kernel time of the fp16x3 path of bench.py.
for F in "-DCNERF_H3_BALANCE=0 -DCNERF_H3_VPM=7" "-DCNERF_H3_BALANCE=1 -DCNERF_H3_VPM=5" "-DCNERF_H3_BALANCE=1 -DCNERF_H3_VPM=7" "-DCNERF_H3_BALANCE=1 -DCNERF_H3_VPM=4" "-DCNERF_H3_BALANCE=1 -DCNERF_H3_VPM=6"; do
  touch conditioned-nerf-gan_amd/csrc/field_h3.hip
  CNERF_EXTRA_FLAGS="$F" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || echo "build failed for [$F]"
  echo "== flags: [$F]"
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp16x3 launch ms', round(d['fp16x3_split_path']['avg_launch_ms'],3), 'rays/s', round(d['fp16x3_split_path']['value']), '| fp16 launch ms', round(d['fp16_single_pass_path']['avg_launch_ms'],3))"
done
touch conditioned-nerf-gan_amd/csrc/field_h3.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
