# column kernel v2 (LDS-DMA patch staging, one-phase fragment ring): parity, micro-benchmark, phase cycles
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "test_conv" 2>&1 | tail -3
PAD16=1 timeout 300 python tools/conv_bench.py bf16 comb 2>&1 | grep comb | tee $O/conv_bench_comb.txt
for a in 7; do for s in "comb0 9->18 7x7 @4K" "comb2 18->3 7x7 @4K"; do ALGO=$a PAD16=1 timeout 120 python tools/patch_timeline.py "$s" 2>&1 | tail -1; done; done | tee $O/timeline_comb.txt
