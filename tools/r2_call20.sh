cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2v; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py > $O/bench_r_448.json 2>/dev/null; tail -1 $O/bench_r_448.json | cut -c1-160
timeout 100 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-130
