# GIMM-VFI-F: space-to-depth form of the filter == stride convolutions (GVFI_S2D) and the Twins / cost-encoder linears on the
# weights-direct variant (GVFI_F_LIN_WDIR), same-box A/B; kernel parity of the new form
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3t; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "space_to_depth" 2>&1 | tail -2
for i in 1 2; do
for v in "1 0" "0 0" "1 1"; do set -- $v
  echo "F s2d=$1 lin_wdir=$2: $(GVFI_S2D=$1 GVFI_F_LIN_WDIR=$2 timeout 200 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done; done | tee $O/f_s2d_ab.txt
timeout 600 python -m pytest tests/test_gimmvfi_f.py tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -x -k "F or gimmvfi_f or f_" 2>&1 | tail -2
