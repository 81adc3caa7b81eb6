# Round-2 evidence call after the halo-staged 3x3 kernel (conv_p3x3.hip): whole GPU suite, bench lines, kernel-trace
# summaries, PMC passes of the new dominant kernel.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=10 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|demo|2k_|4k_|demo2k|F |SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log
timeout 300 python bench.py --shapes $O/conv_shapes_r_448.md > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-250
timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_2k.md 2>/dev/null | tail -1 > $O/bench_r_2k.json; cut -c1-160 $O/bench_r_2k.json
timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_r_4k.md 2>/dev/null | tail -1 > $O/bench_r_4k.json; cut -c1-160 $O/bench_r_4k.json
timeout 200 python bench.py --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 > $O/bench_f_448.json; cut -c1-160 $O/bench_f_448.json
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_2k.json; cut -c1-160 $O/bench_f_2k.json
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_4k.json; cut -c1-160 $O/bench_f_4k.json
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench.txt
for m in r f; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o run -- python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_$m.log 2>&1
  python tools/rocpd_stats.py $O/prof_$m $O/kernel_stats_${m}_448.md > /dev/null; rm -rf $O/prof_$m
done
head -12 $O/kernel_stats_r_448.md | cut -c1-150
pmc() { n=$1; shift; rm -rf $O/pmc_$n; ONLYP3=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$n -o run -- python tools/conv_bench.py bf16 "final.resblock 256->256 3x3 @256" > $O/pmc_$n.log 2>&1; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python tools/pmc_report.py p3x3 $O/pmc_mfma $O/pmc_fetch $O/pmc_write > $O/pmc_hotconv.txt 2>&1; cat $O/pmc_hotconv.txt
rm -rf $O/pmc_mfma $O/pmc_fetch $O/pmc_write
