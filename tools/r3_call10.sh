cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3j; mkdir -p $O
OCC2=$GRAFT_REPO_ROOT/gimm-vfi_amd/lib/libgimmvfi_hip_occ2.so
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do
  echo occ3; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
  echo occ2; GVFI_LIB_PATH=$OCC2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
done
echo "occ3 lanes 3"; GVFI_RAFT_LANES=3 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
echo "F occ3"; timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
echo "F occ2"; GVFI_LIB_PATH=$OCC2 timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
timeout 300 python tools/ring_bench.py > $O/ring_bench_occ3.txt 2>&1; head -3 $O/ring_bench_occ3.txt | cut -c1-200
