cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $O
timeout 400 python tools/ring_bench.py --stamps > $O/ring_bench.txt 2>&1; tail -1 $O/ring_bench.txt | cut -c1-200
timeout 300 python bench.py --shapes $O/conv_shapes_r_448.md --no-cpu-baseline > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-200
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_r -o run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_r.log 2>&1
python tools/rocpd_stats.py $O/prof_r $O/kernel_stats_r_448.md > /dev/null; rm -rf $O/prof_r; head -30 $O/kernel_stats_r_448.md | cut -c1-170
timeout 900 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "448x256" > $O/gpu_448.log 2>&1; grep -E "^(R |F |448)|passed|failed|Error|assert" $O/gpu_448.log | cut -c1-200
