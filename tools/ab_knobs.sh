#!/bin/bash
# A/B of run-time knobs on one box: each line of the here-doc is "label | ENV=VAL ENV=VAL ..."; bench.py c2, 400 steps, 2 repetitions.
# usage: bash tools/ab_knobs.sh out_name < spec      (writes gpurun_out/<out_name>.txt)
out=gpurun_out/${1:-ab_knobs}.txt
: > $out
while IFS='|' read -r label envs; do
  [ -z "$label" ] && continue
  for rep in 1 2; do
    r=$(env $envs python bench.py --steps ${STEPS:-400} --warmup 40 --no-cpu-baseline --no-classic-reference --no-end-to-end ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = d['stage_ms']
print('%.4f ms/step | host %.3f | L0 fwd in-step %.2f us | alone fwd %.1f dgrad %.1f wgrad %.1f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], 1e3 * (d['roofline']['layers'][0]['ms'] if 'layers' in d['roofline'] else d['roofline']['ms']), 1e3 * s['mlp0_fwd'], 1e3 * s['mlp0_dgrad'], 1e3 * s['mlp0_wgrad']))")
    echo "$label: $r" | tee -a $out
  done
done
