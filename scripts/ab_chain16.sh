# A/B of chain16 variants on the GPU box: rebuilds bwd16.hip with a macro, times forward + backward (diagnostic only).
set -e
for F in "" "-DC16_NOSCATTER" "-DC16_NOSTORE" "-DC16_NOSCATTER -DC16_NOSTORE"; do
  touch conditioned-nerf-gan_amd/csrc/bwd16.hip
  CNERF_EXTRA_FLAGS="$F" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "== flags: [$F]"
  python scripts/time_backward.py 8 fp16x3 fp16 2>&1 | grep "fwd+bwd"
done
touch conditioned-nerf-gan_amd/csrc/bwd16.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
