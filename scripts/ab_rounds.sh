# Same-box A/B: round 2's tree (unpacked next to this one as _r02_tree by the caller: git archive 611610c) against the current tree --
# forward kernels per launch (bench.py fast-path fields) and render forward + backward per step (scripts/time_backward.py).
for T in _r02_tree .; do
  ( cd $T
    python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
    echo "== tree $T"
    timeout -k 10 300 python bench.py --no-train-step --no-cpu-baseline $( [ $T = . ] && echo --no-gan-step ) 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('  fp32 %.2f ms/launch (%.3f of peak) | fp16x3 %.2f (%.3f) | fp16 %.2f (%.3f) | step %.2f ms' % (d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['fp16x3_split_path']['avg_launch_ms'], d['fp16x3_split_path']['frac_of_fp16_mfma_peak'], d['fp16_single_pass_path']['avg_launch_ms'], d['fp16_single_pass_path']['frac_of_fp16_mfma_peak'], d['ms_per_step']))"
    for V in SHORTSIREN_FG TALLSIREN_dRes TALLSIREN_FG; do
      echo "  $V: $(CNERF_VARIANT=$V python scripts/time_backward.py 8 fp16x3 fp16 2>&1 | grep 'fwd' | tr '\n' '|')"
    done
    echo "  SHORTSIREN_FG fp32/fp32: $(python scripts/time_backward.py 8 fp32 fp32 2>&1 | grep 'fwd+bwd')"
  )
done
