# Round-2 closing call: whole GPU suite, smoke(), the default bench line, the 4K kernel-trace summary, HBM bytes per kernel.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2q; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=10 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|demo|2k_|4k_|demo2k|F |SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py --shapes $O/conv_shapes_r_448.md > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-200
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_r -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_r.log 2>&1
python tools/rocpd_stats.py $O/prof_r $O/kernel_stats_r_448.md > /dev/null; rm -rf $O/prof_r
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof4k -o run -- python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof4k.log 2>&1
python tools/rocpd_stats.py $O/prof4k $O/kernel_stats_r_4k.md > /dev/null; rm -rf $O/prof4k; head -9 $O/kernel_stats_r_4k.md | cut -c1-140
for c in FETCH_SIZE WRITE_SIZE; do GIMMVFI_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_all_$c -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_all_$c.log 2>&1; done
python tools/pmc_table.py $O/kernel_stats_r_448.md $O/hbm_table_r_448.md $O/pmc_all_FETCH_SIZE $O/pmc_all_WRITE_SIZE | head -16 | cut -c1-170
rm -rf $O/pmc_all_FETCH_SIZE $O/pmc_all_WRITE_SIZE
