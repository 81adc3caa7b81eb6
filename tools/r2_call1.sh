# Round-2 measurement call 1 (one gpurun call): the whole GPU test-suite incl. the new 448x256 live-oracle / 2K / 4K /
# demo-frame parity tests, the driver's bench line, the lane-count A/B of the RAFT recurrence, 2K / 4K bench lines (R and F),
# the F switches, the hot-kernel phase timeline and a rocprofv3 kernel-trace summary of the default bench.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=15 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3
grep -E "^(448x256|demo|2k_|4k_|demo2k)" $O/gpu_tests.log
timeout 300 python bench.py --shapes $O/conv_shapes_r_448.md > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-400
timeout 400 python tools/lanes_probe.py 2>&1 | grep -v amdgpu.ids > $O/lanes_probe.txt; cat $O/lanes_probe.txt
timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_2k.md > $O/bench_r_2k.json 2> $O/bench_r_2k.err; tail -1 $O/bench_r_2k.json | cut -c1-300
timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_4k.md > $O/bench_r_4k.json 2> $O/bench_r_4k.err; tail -1 $O/bench_r_4k.json | cut -c1-300
timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_f_448.md > $O/bench_f_448.json 2> $O/bench_f_448.err; tail -1 $O/bench_f_448.json | cut -c1-200
for sw in GVFI_F_S2D GVFI_ATTN_LDS; do env $sw=1 timeout 100 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120 > $O/bench_f_448_$sw.json; echo "$sw: $(cat $O/bench_f_448_$sw.json)"; done
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_f_2k.json 2> $O/bench_f_2k.err; tail -1 $O/bench_f_2k.json | cut -c1-200; tail -2 $O/bench_f_2k.err
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_f_4k.json 2> $O/bench_f_4k.err; tail -1 $O/bench_f_4k.json | cut -c1-200; tail -2 $O/bench_f_4k.err
timeout 100 python tools/conv_timeline.py "final.resblock 256->256" 256 > $O/timeline_hot.txt 2>&1; cat $O/timeline_hot.txt
timeout 100 python tools/conv_bench.py bf16 "final.resblock 256->256" > $O/conv_bench_hot.txt 2>&1; cat $O/conv_bench_hot.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_r -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_r.log 2>&1
python tools/rocpd_stats.py $O/prof_r $O/kernel_stats_r_448.md > /dev/null; rm -rf $O/prof_r; head -30 $O/kernel_stats_r_448.md | cut -c1-170
