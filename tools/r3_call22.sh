# CLI end to end with the hand-written PNG encoder (png_bytes_rgb) in VideoSink
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3w; mkdir -p $O
export GVFI_CLI_TIMING=1
timeout 70 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
timeout 40 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt
