# round-3 evidence call B: bench lines of every configuration, per-shape tables, kernel-trace summaries, CLI, comb-block A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3fb; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_hires.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x -k "not fp32" 2>&1 | tail -2 | tee $O/e2e_bf16.txt
for i in 1 2; do for v in 1 0; do
  echo "R 4K col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done; done | tee $O/ab_col7.txt
for v in 1 0; do
  echo "R 2K col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
  echo "R 448 col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done | tee -a $O/ab_col7.txt
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
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_4k -o run -- python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_4k.log 2>&1
python tools/rocpd_stats.py $O/prof_4k $O/kernel_stats_r_4k.md > /dev/null; rm -rf $O/prof_4k
head -14 $O/kernel_stats_r_448.md | cut -c1-150
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt
timeout 300 python tools/cli_bench.py 17 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
