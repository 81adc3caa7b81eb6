cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2j; mkdir -p $O
timeout 300 python bench.py > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-1200
timeout 200 python bench.py --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 > $O/bench_f_448.json; cut -c1-160 $O/bench_f_448.json
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_2k.json; cut -c1-160 $O/bench_f_2k.json
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_4k.json; cut -c1-160 $O/bench_f_4k.json
timeout 300 python -m pytest tests/test_gimmvfi_f.py tests/test_kernels_f.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_f -o run -- python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_f.log 2>&1
python tools/rocpd_stats.py $O/prof_f $O/kernel_stats_f_448.md > /dev/null; rm -rf $O/prof_f; head -10 $O/kernel_stats_f_448.md | cut -c1-150
