#!/bin/bash
# Same-box A/B of environment settings on the R and F 448x256 headlines (graph replay, 20 steps), alternating, REPS rounds.
# usage: tools/ab_env.sh REPS "A=1 B=2" "C=3" ...      (each argument: one setting = a list of VAR=value; "" = defaults)
R=$1; shift
for i in $(seq $R); do
  for s in "$@"; do
    for mdl in r f; do
      line=$(env $s timeout 300 python bench.py --configs none --no-cpu-baseline --steps 20 --warmup 5 --model $mdl 2>/dev/null | tail -1)
      echo "[$s] model=$mdl $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "frames/s", d["ms_per_step"], "ms")')"
    done
  done
done
