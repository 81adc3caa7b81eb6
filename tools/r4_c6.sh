# round-4 call 6: unconditional tap loads (samplers), device-side composition of the CLI, flow-scale families of GIMM-VFI-F
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -rP > $O/tests.log 2>&1; tail -3 $O/tests.log; grep "^CLI" $O/tests.log
timeout 900 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "family or (hires_matches and not f_)" > $O/hires.log 2>&1; grep -E "^(2k_|4k_|demo|FAMILY)|passed|failed|Error" $O/hires.log | cut -c1-230
b4k() { timeout 300 python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4k $1', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
b4k new; b4k new
GVFI_CLI_TIMING=1 timeout 300 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; cat $O/cli_bench_2k.txt | cut -c1-300
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt | cut -c1-300
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_r_4k.md > /dev/null; rm -rf $O/prof; head -14 $O/kernel_stats_r_4k.md | cut -c1-150; grep -E "warp_nhwc|warp_blend" $O/kernel_stats_r_4k.md | cut -c1-150
