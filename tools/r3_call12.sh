cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3l; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
