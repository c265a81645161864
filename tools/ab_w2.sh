#!/bin/bash
# A/B of the two-blocks-per-CU GEMM build (tf_repos_amd.build --variant w2 -DDR_WAVES2) against the product library, c2, on one box.
# Writes gpurun_out/ab_w2.txt
out=gpurun_out/ab_w2.txt
: > $out
run() {   # label, env...
  label=$1; shift
  for rep in 1 2; do
    r=$(env "$@" python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-classic-reference --no-end-to-end 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['stage_ms']
print('%.4f ms/step | in-step L0 fwd %.2f us frac %.3f | alone fwd %.1f dgrad %.1f wgrad %.1f | stages fwd %.1f bwd %.1f' % (d['ms_per_step'], 1e3 * d['roofline']['ms'], d['roofline']['frac'], 1e3 * s['mlp0_fwd'], 1e3 * s['mlp0_dgrad'], 1e3 * s['mlp0_wgrad'], 1e3 * s['forward'], 1e3 * s['backward_dense']))")
    echo "$label: $r" | tee -a $out
  done
}
run "base                       " DCTR_X=0
run "w2                         " DCTR_LIB_VARIANT=w2
run "w2 dgrad tile 2x13         " DCTR_LIB_VARIANT=w2 DCTR_DR_TILE=d0
run "w2 fwd 4x7                 " DCTR_LIB_VARIANT=w2 DCTR_DR_TILE=f1
run "w2 dgrad 2x13 + wgrad late " DCTR_LIB_VARIANT=w2 DCTR_DR_TILE=d0 DCTR_WGRAD_LATE=1
