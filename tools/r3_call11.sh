cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
timeout 300 python tools/ring_bench.py --stamps > $O/ring_bench_kb256.txt 2>&1
GVFI_WDIR256=0 timeout 300 python tools/ring_bench.py > $O/ring_bench_kb128.txt 2>&1
for i in 1 2; do
  echo kb256; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
  echo kb128; GVFI_WDIR256=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
done
echo "F kb256"; timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
echo "F kb128"; GVFI_WDIR256=0 timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
