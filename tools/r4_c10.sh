# round-4 call 10: kernels after the dependent-load clean-up: tests + kernel times + bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c10; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_f.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
for c in f448 r448 r4k; do
  a=""; [ $c = f448 ] && a="--model f"; [ $c = r4k ] && a="--batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o run -- python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 $a > $O/prof_$c.log 2>&1
  python tools/rocpd_stats.py $O/prof_$c $O/kernel_stats_$c.md > /dev/null; rm -rf $O/prof_$c
  grep -E "cost_embed1|cost_lookup|softsplat_gather|warp_nhwc_kernel|total kernel" $O/kernel_stats_$c.md | cut -c1-170
  python -c "import sys,json; d=json.loads(open('$O/prof_$c.log').read().strip().splitlines()[-1]); print('$c (profiled run)', d['value'], d['ms_per_step'])"
done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_all.json 2> $O/bench_all.err; echo "rc $?" >> $O/bench_all.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4c10/bench_all.json') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for c in d.get('configs',[]): print(c.get('baseline_config'), c.get('value'), c.get('ms_per_step'), c.get('error'))
PY
