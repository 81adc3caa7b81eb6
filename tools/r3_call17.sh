# CLI end to end after the parallel PNG writer (VideoSink) and the wider result drain
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3r; mkdir -p $O
export GVFI_CLI_TIMING=1
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt
timeout 300 python tools/cli_bench.py 17 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
timeout 300 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k_33.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k_33.txt
python -c "import os; print('cpus', os.cpu_count())"
