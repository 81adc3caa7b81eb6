# column kernel of the 7x7 combination block (conv_col7.hip): parity, micro-benchmark against the patch kernel, phase cycles, end to end
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
PAD16=1 timeout 300 python tools/conv_bench.py bf16 comb 2>&1 | tee $O/conv_bench_comb.txt
for a in 3 7; do for s in "comb0 9->18 7x7 @4K" "comb2 18->3 7x7 @4K"; do ALGO=$a PAD16=1 timeout 120 python tools/patch_timeline.py "$s" 2>&1 | tail -1; done; done | tee $O/timeline_comb.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -x -k "not fp32" 2>&1 | tail -3
for i in 1 2; do for v in 1 0; do
  echo "R 4K col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --height 2176 --width 4096 --ds 0.25 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done; done | tee $O/ab_4k.txt
for v in 1 0; do
  echo "R 2K col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
  echo "R 448 col7=$v: $(GVFI_COL7=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done | tee $O/ab_2k_448.txt
