# round-4 call 12: row-linear kernel (conv_lin.hip) -- GPU kernel cases, F parity, same-box A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -k "conv or flowformer or gpu_f" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "(hires_f_matches and not fast) or f_448_b8_flow" > $O/hires.log 2>&1; grep -E "^F |passed|failed" $O/hires.log | cut -c1-200
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline --model f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in 1 0 1 0; do GVFI_LIN=$v b "F448 lin=$v" --steps 10 --warmup 3; done
for v in 1 0; do GVFI_LIN=$v b "F4k lin=$v" --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
timeout 400 python bench.py --configs none --no-cpu-baseline --model f --steps 5 --warmup 2 --shapes $O/conv_shapes_f448.md > $O/bench_f448.json 2> $O/bench_f448.err; grep -E "conv_lin" $O/conv_shapes_f448.md | cut -c1-160
