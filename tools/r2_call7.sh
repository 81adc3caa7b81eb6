cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hires.py tests/test_gpu_e2e.py tests/test_zz_cli_f.py tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -rP > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|demo|2k_|4k_|demo2k|F |CLI)|passed|failed|rc " $O/gpu_tests.log
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench.txt
timeout 300 python tools/cli_bench.py 9 2048 1080 8 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
