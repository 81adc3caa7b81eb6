cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_kernels_f.py -m gpu -q -p no:cacheprovider -x -k "fp16" 2>&1 | tail -3
timeout 1200 python tools/f_policy_diag.py demo_864x736 2k_ds050 demo2k_ds050 4k_ds025 "--policies=dec:f16;upd:f16" > $O/f_policy_f16.txt 2>&1; grep flow_precision $O/f_policy_f16.txt | cut -c1-300
for fp in bf16 dec:f16 dec; do timeout 200 python bench.py --model f --flow-precision $fp --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
for l in 1 2 3 4; do echo "lanes $l"; GVFI_RAFT_LANES=$l timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
