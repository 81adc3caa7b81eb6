cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "corr_lookup or fp16" 2>&1 | tail -2
# A/B of the LDS-staged correlation look-up (same process would be better; whole-bench A/B x2 to see the noise)
for v in 0 1 0 1; do echo "GVFI_LOOKUP_LDS=$v"; GVFI_LOOKUP_LDS=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140; done
timeout 200 python tools/lookup_bench.py > $O/lookup_ab.txt 2>&1; cat $O/lookup_ab.txt
timeout 1500 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "f_ or _f" > $O/gpu_hires_f.log 2>&1; grep -E "^F |passed|failed|Error|assert " $O/gpu_hires_f.log | cut -c1-260
