#!/bin/bash
# Same-box A/B of the hot 3x3 kernel's launch form at the model level: GVFI_P3X3_FORM=1 (round-2 kernel) / 2 (tile per workgroup,
# wave-private epilogue) / default (persistent stream kernel), alternating, headline workload + R 2K.  usage: tools/ab_p3x3_form.sh [rounds]
R=${1:-2}
for i in $(seq $R); do
  for f in 1 2 0; do
    for cfg in "--batch 8" "--batch 1 --height 1088 --width 2048 --ds 0.5 --n-interp 8"; do
      GVFI_P3X3_FORM=$f python bench.py --steps 20 --warmup 5 --configs none --no-cpu-baseline $cfg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('form $f', d['config']['workload'][:40], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], '| hot kernel', r.get('kernel', '')[:40], 'frac', r['frac'])
"
    done
  done
done
