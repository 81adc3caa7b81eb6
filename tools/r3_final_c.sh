# round-3 final evidence call: the whole GPU suite at HEAD, smoke, driver line, GIMM-VFI-F lines (space-to-depth form), CLI
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3fc; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -rP > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|R |F |demo|2k_|4k_|demo2k|SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log | cut -c1-220 > $O/gpu_parity.log; tail -2 $O/gpu_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_r_448.json 2> $O/bench_r_448.err; tail -1 $O/bench_r_448.json | cut -c1-120
timeout 300 python bench.py --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 > $O/bench_f_448.json; cut -c1-120 $O/bench_f_448.json
timeout 300 python bench.py --model f --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_2k.json; cut -c1-120 $O/bench_f_2k.json
timeout 300 python bench.py --model f --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_f_4k.json; cut -c1-120 $O/bench_f_4k.json
export GVFI_CLI_TIMING=1
timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt
timeout 300 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_2k.txt
