cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2p; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "flow_to_image or cli" -rP 2>&1 | grep -E "^CLI|passed|failed|Error" | head
timeout 300 python tools/cli_bench.py 9 2048 1080 8 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt; tail -3 $O/cli_bench_2k.txt | cut -c1-300
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench.txt
