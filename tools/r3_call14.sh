cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do for v in 1 0; do
  echo "R fuse=$v: $(GVFI_FUSE_SEAM=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done; done
for v in 1 0; do
  echo "F fuse=$v: $(GVFI_FUSE_SEAM=$v timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
