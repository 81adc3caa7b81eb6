cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "p3x3" > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
for f in "side 64" "cnn_enc" "final.up"; do P3S=1 timeout 100 python tools/conv_bench.py bf16 "$f" 2>&1 | grep -v amdgpu.ids; done | tee $O/conv_p3s.txt
for f in "side 64" "cnn_enc"; do P3S=1 RES=1 timeout 100 python tools/conv_bench.py bf16 "$f" 2>&1 | grep -v amdgpu.ids; done | tee $O/conv_p3s_res.txt
for v in 0 1 0 1; do GVFI_P3X3=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done | tee $O/bench_ab_448.txt
