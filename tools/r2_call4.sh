cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "test_conv" 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
for i in 1 2; do timeout 300 python tools/conv_bench.py bf16 "final.resblock" 2>&1 | grep -v amdgpu.ids >> $O/conv_bench.txt; done; cat $O/conv_bench.txt
timeout 100 python tools/conv_timeline.py "final.resblock 256->256" 256 2>&1 | grep -v amdgpu.ids > $O/timeline_hot.txt; cat $O/timeline_hot.txt
GVFI_DEEP_RING=1 timeout 100 python tools/conv_timeline.py "final.resblock 256->256" 256 2>&1 | grep -v amdgpu.ids > $O/timeline_hot_deep.txt; cat $O/timeline_hot_deep.txt
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
GVFI_DEEP_RING=1 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
