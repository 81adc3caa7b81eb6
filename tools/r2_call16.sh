cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2r; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x -rP 2>&1 | grep -E "^F |PSNR|passed|failed|Error|assert" | head -20
for v in 0 1 0 1; do GVFI_ATTN_MFMA=$v timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130; done | tee $O/bench_f_ab.txt
