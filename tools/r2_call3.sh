cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
timeout 300 python tools/conv_bench.py bf16 2>&1 | grep -v amdgpu.ids > $O/conv_bench.txt; cat $O/conv_bench.txt
ABLATE=1 timeout 200 python tools/conv_bench.py bf16 "final.resblock 256->256" 2>&1 | grep -v amdgpu.ids > $O/ablate_hot.txt; cat $O/ablate_hot.txt
timeout 100 python tools/conv_timeline.py "final.resblock 256->256" 256 2>&1 | grep -v amdgpu.ids > $O/timeline_hot.txt; cat $O/timeline_hot.txt
timeout 200 python bench.py --no-cpu-baseline --shapes $O/conv_shapes_r_448.md 2>/dev/null | tail -1 > $O/bench_r_448.json; cut -c1-200 $O/bench_r_448.json
