cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "p3x3 or stats" > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
timeout 600 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -x -k "448" -rP 2>&1 | grep -E "^448|passed|failed" 
for v in 1 1; do timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
