cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
tools/microbench/mfma_peak > $O/mfma_peak.txt 2>&1; tail -7 $O/mfma_peak.txt
timeout 1200 python tools/f_policy_diag.py demo_864x736 2k_ds050 demo2k_ds050 4k_ds025 "--policies=bf16;tok;upd;dec" > $O/f_policy.txt 2>&1; tail -4 $O/f_policy.txt | cut -c1-260
timeout 600 python tools/f_policy_diag.py demo2k_ds050_fh015 4k_ds025_fh015 "--policies=bf16;dec;FULL-FP32" > $O/f_policy_fh015.txt 2>&1; tail -6 $O/f_policy_fh015.txt | cut -c1-260
for fp in bf16 tok dec; do GIMMVFI_F_FLOW_PRECISION=$fp timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
HIP_FORCE_DEV_KERNARG=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
HIP_FORCE_DEV_KERNARG=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
