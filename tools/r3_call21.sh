# token chains: second A/B pair + the GIMM-VFI-F end-to-end GPU tests against the live oracle with the chains on
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3v; mkdir -p $O
for v in 0 1; do
  echo "F tokchain=$v: $(GVFI_F_TOKCHAIN=$v timeout 60 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c40-75)"
done | tee $O/tokchain_ab2.txt
timeout 100 python -m pytest tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tee $O/gimmvfi_f_gpu.txt
