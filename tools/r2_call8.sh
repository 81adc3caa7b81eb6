cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r2h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "p3x3" > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
P3=1 timeout 300 python tools/conv_bench.py bf16 "final.resblock" > $O/conv_p3.txt 2>&1; cat $O/conv_p3.txt
P3=1 RES=1 timeout 300 python tools/conv_bench.py bf16 "final.resblock" > $O/conv_p3_res.txt 2>&1; cat $O/conv_p3_res.txt
for v in 0 1; do GVFI_P3X3=$v timeout 300 python bench.py --height 1088 --width 2048 --ds 0.5 --batch 1 --n-interp 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done | tee $O/bench_ab_2k.txt
for v in 0 1; do GVFI_P3X3=$v timeout 300 python bench.py --height 2176 --width 4096 --ds 0.25 --batch 1 --n-interp 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done | tee $O/bench_ab_4k.txt
