# Round-2 closing call (after the MFMA attention / GELU changes of GIMM-VFI-F): whole GPU suite, smoke(), F bench lines, F kernel-trace summary.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2t; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=10 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|demo|2k_|4k_|demo2k|F |SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 200 python bench.py --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 > $O/bench_f_448.json; cut -c1-160 $O/bench_f_448.json
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_2k.json; cut -c1-160 $O/bench_f_2k.json
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_4k.json; cut -c1-160 $O/bench_f_4k.json
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_f -o run -- python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_f.log 2>&1
python tools/rocpd_stats.py $O/prof_f $O/kernel_stats_f_448.md > /dev/null; rm -rf $O/prof_f; head -10 $O/kernel_stats_f_448.md | cut -c1-140
