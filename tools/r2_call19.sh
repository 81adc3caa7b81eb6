cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2u; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x -rP 2>&1 | grep -E "^\[gimmvfi_f bf16|passed|failed|Error" | head -8
for v in 0 65536 1000000000 0 65536 1000000000; do GVFI_TOK_LINEAR_ROWS=$v timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130; done | tee $O/bench_f_toklin_ab.txt
