# parity of the GVFI_MISC_LANES=1 launch order (and of the default order on the same tree): the end-to-end + hi-res GPU tests
# against the committed fixtures, without the live-CPU-oracle cases
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5p; mkdir -p $O
timeout 150 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x -k "not live_oracle" > $O/default_e2e.log 2>&1; echo "rc $?" >> $O/default_e2e.log
GVFI_MISC_LANES=1 timeout 330 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -k "not live_oracle" > $O/misc1_e2e_hires.log 2>&1; echo "rc $?" >> $O/misc1_e2e_hires.log
tail -2 $O/default_e2e.log $O/misc1_e2e_hires.log
