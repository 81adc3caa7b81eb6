cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 400 python tools/ring_bench.py --stamps > $O/ring_bench.txt 2>&1; tail -2 $O/ring_bench.txt | cut -c1-200
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg; echo "WDIR=$1 STATE_F32=$2"; GVFI_WDIR=$1 GVFI_GRU_STATE_F32=$2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
for cfg in "1 1" "0 0"; do set -- $cfg; echo "F WDIR=$1 STATE_F32=$2"; GVFI_WDIR=$1 GVFI_GRU_STATE_F32=$2 timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
timeout 900 python tools/f_policy_diag.py demo_864x736 2k_ds050 demo2k_ds050 4k_ds025 "--policies=bf16" > $O/f_policy_state32.txt 2>&1; tail -4 $O/f_policy_state32.txt | cut -c1-260
GVFI_GRU_STATE_F32=0 timeout 400 python tools/f_policy_diag.py demo2k_ds050 "--policies=bf16" > $O/f_policy_state16.txt 2>&1; tail -1 $O/f_policy_state16.txt | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
