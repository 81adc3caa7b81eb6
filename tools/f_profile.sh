# GIMM-VFI-F (BASELINE.json configs[3]) measurement recipe, one gpurun call (~40 s of box time):
#   /usr/local/graft/bin/gpurun --timeout 150 -- 'bash tools/f_profile.sh'
# writes the bench line, the per-conv-shape table, the rocprofv3 kernel-trace summary and the per-stage times under
# gpurun_out/ ; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# first: the A/B switch paths that have only run in the emulator so far
GVFI_TEST_UNMEASURED=1 timeout 120 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py tests/test_zz_cli_f.py -m gpu -q -p no:cacheprovider -k "switch or cli" 2>&1 | tail -5 > gpurun_out/f_switch_tests.log
for sw in GVFI_F_S2D GVFI_ATTN_LDS; do env $sw=1 timeout 60 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160 > gpurun_out/f_bench_$sw.json; done
timeout 60 python bench.py --model f --steps 5 --warmup 2 --no-cpu-baseline --shapes gpurun_out/f_conv_shapes.md > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
timeout 60 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f -o runf -- python bench.py --model f --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/f_bench_prof.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_f gpurun_out/f_kernel_stats.md > /dev/null; rm -rf gpurun_out/prof_f
timeout 60 python tools/f_stage_times.py > gpurun_out/f_stage_times.md 2>&1
cat gpurun_out/f_switch_tests.log gpurun_out/f_bench_GVFI_*.json; tail -1 gpurun_out/f_bench.json | cut -c1-300; head -12 gpurun_out/f_kernel_stats.md | cut -c1-160; cat gpurun_out/f_stage_times.md
