cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "p3x3 or test_conv" > $O/kernel_tests.log 2>&1; tail -2 $O/kernel_tests.log
timeout 200 python tools/p3x3_timeline.py "final.resblock 256->256 3x3 @256" 2>&1 | grep -v amdgpu.ids | tee $O/p3x3_timeline.txt
P3=1 timeout 300 python tools/conv_bench.py bf16 "final.resblock" 2>&1 | grep -v amdgpu.ids | tee $O/conv_p3.txt
P3=1 RES=1 timeout 300 python tools/conv_bench.py bf16 "final.resblock" 2>&1 | grep -v amdgpu.ids | tee $O/conv_p3_res.txt
for v in 0 1 0 1; do GVFI_P3X3=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done | tee $O/bench_ab_448.txt
