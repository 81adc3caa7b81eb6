# round-4 call 4: list-based gather form of the softmax splat -- kernel + end-to-end parity, same-box A/B against the atomic scatter
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gimm_model.py -m gpu -q -p no:cacheprovider -x -k "splat or golden or e2e or gimm or taps" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "hires_matches and not f_" > $O/hires.log 2>&1; grep -E "^(2k_|4k_|demo)|passed|failed" $O/hires.log | cut -c1-200
for v in 1 0 1 0; do
  GVFI_SPLAT_GATHER=$v timeout 300 python bench.py --configs none --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('448 gather=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
for v in 1 0; do
  GVFI_SPLAT_GATHER=$v timeout 300 python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4k gather=$v', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_r_448.md > /dev/null; rm -rf $O/prof; grep -i "splat" $O/kernel_stats_r_448.md | cut -c1-200
