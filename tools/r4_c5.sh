# round-4 call 5: folded finalisation (col7), LDS-staged decoder taps (combine_warps_up), gather-splat kernel tests; 4K A/B; full bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "splat or col7 or combine" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "hires_matches and not f_" > $O/hires.log 2>&1; grep -E "^(2k_|4k_|demo)|passed|failed" $O/hires.log | cut -c1-160
b4k() { timeout 300 python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4k $1', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
b4k new
GVFI_FOLD_FINALIZE=0 b4k nofold
b4k new
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_all.json 2> $O/bench_all.err; echo "rc $?" >> $O/bench_all.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_r_4k.md > /dev/null; rm -rf $O/prof; head -16 $O/kernel_stats_r_4k.md | cut -c1-150
