# fused flow-token chains of GIMM-VFI-F (csrc/token_chain.hip): kernel parity on the GPU, same-box A/B, hi-res fixtures in the default policy
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3u; mkdir -p $O
timeout 100 python -m pytest tests/test_kernels_f.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | tee $O/kernels_f.txt
for v in 1 0; do
  echo "F tokchain=$v: $(GVFI_F_TOKCHAIN=$v timeout 100 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done | tee $O/tokchain_ab.txt
timeout 200 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "hires_f_matches and bf16 and not fast" 2>&1 | grep -E "^F |passed|failed" | cut -c1-200 | tee $O/hires_f_bf16.txt
