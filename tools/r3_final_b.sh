# round-3 evidence call B: bench lines of every configuration, per-shape tables, kernel-trace summaries, CLI
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3fb; mkdir -p $O
timeout 400 python bench.py --shapes $O/conv_shapes_r_448.md > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-200
timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_2k.md 2>/dev/null | tail -1 > $O/bench_r_2k.json; cut -c1-150 $O/bench_r_2k.json
timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_4k.md 2>/dev/null | tail -1 > $O/bench_r_4k.json; cut -c1-150 $O/bench_r_4k.json
timeout 300 python bench.py --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 > $O/bench_f_448.json; cut -c1-150 $O/bench_f_448.json
timeout 200 python bench.py --model f --flow-precision bf16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_448_allbf16.json; cut -c1-150 $O/bench_f_448_allbf16.json
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_2k.json; cut -c1-150 $O/bench_f_2k.json
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_4k.json; cut -c1-150 $O/bench_f_4k.json
for m in r f; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o run -- python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_$m.log 2>&1
  python tools/rocpd_stats.py $O/prof_$m $O/kernel_stats_${m}_448.md > /dev/null; rm -rf $O/prof_$m
done
head -14 $O/kernel_stats_r_448.md | cut -c1-150
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt
timeout 300 python tools/cli_bench.py 17 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
