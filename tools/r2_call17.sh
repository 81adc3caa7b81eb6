cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "test_conv" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x -rP 2>&1 | grep -E "^\[gimmvfi_f bf16|passed|failed|Error" | head -8
for v in 1 1; do timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130; done | tee $O/bench_f.txt
timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline --shapes $O/conv_shapes_f_448.md 2>/dev/null | tail -1 | cut -c1-130; grep -E "128->512|512->128|128->128 1x1|GELU" $O/conv_shapes_f_448.md | head -8 | cut -c1-140
