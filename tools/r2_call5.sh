cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_hires.py -m gpu -q -x -p no:cacheprovider -rP 2>&1 | grep -E "^(448x256|demo|2k_|4k_|demo2k)|passed|failed|Error" > $O/tests.log; cat $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --shapes $O/conv_shapes_r_448.md 2>/dev/null | tail -1 > $O/bench_r_448.json; cut -c1-200 $O/bench_r_448.json
timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_2k.md 2>/dev/null | tail -1 > $O/bench_r_2k.json; cut -c1-200 $O/bench_r_2k.json
timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_4k.md 2>/dev/null | tail -1 > $O/bench_r_4k.json; cut -c1-200 $O/bench_r_4k.json
grep -E "patch|conv_igemm_kernel" $O/conv_shapes_r_448.md $O/conv_shapes_r_4k.md
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof4k -o run -- python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof4k.log 2>&1
python tools/rocpd_stats.py $O/prof4k $O/kernel_stats_r_4k.md > /dev/null; rm -rf $O/prof4k; head -24 $O/kernel_stats_r_4k.md | cut -c1-150
