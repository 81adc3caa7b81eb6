# round-4 call 13: row-linear kernel v2 (LDS-staged rows) -- GPU kernel cases, same-box A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c13; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "test_conv" > $O/tests.log 2>&1; tail -2 $O/tests.log
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline --model f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in 1 0 1 0; do GVFI_LIN=$v b "F448 lin=$v" --steps 10 --warmup 3; done
GVFI_LIN=1 timeout 400 python bench.py --configs none --no-cpu-baseline --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f448.md > $O/bench_f448.json 2> $O/bench_f448.err; grep -E "conv_lin" $O/conv_shapes_f448.md | cut -c1-160 | head -12
GVFI_LIN=1 timeout 600 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "f_448_b8_flow or (hires_f_matches and 2k_ds050 and not fast)" > $O/hires.log 2>&1; grep -E "passed|failed" $O/hires.log
