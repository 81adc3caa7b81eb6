cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
tools/microbench/mfma_peak > $O/mfma_peak.txt 2>&1; tail -4 $O/mfma_peak.txt
timeout 400 python tools/ring_bench.py --stamps > $O/ring_bench.txt 2>&1; tail -3 $O/ring_bench.txt | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
timeout 900 python tools/f_policy_diag.py demo2k_ds050 4k_ds025 > $O/f_policy.txt 2>&1; tail -4 $O/f_policy.txt | cut -c1-250
timeout 200 python bench.py --model f --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_f_fp32.json 2>$O/bench_f_fp32.err; tail -1 $O/bench_f_fp32.json | cut -c1-200
timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_f_bf16.json 2>$O/bench_f_bf16.err; tail -1 $O/bench_f_bf16.json | cut -c1-200
