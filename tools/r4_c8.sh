# round-4 call 8: fused flow-token path of GIMM-VFI-F -- kernel bit-identity on the GPU, end-to-end parity, same-box A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_hires.py -m gpu -q -p no:cacheprovider -rP -k "hires_f_matches and bf16 and not fast" > $O/hires.log 2>&1; grep -E "^F |passed|failed" $O/hires.log | cut -c1-200
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline --model f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in 1 0 1 0; do GVFI_F_TOKPATH=$v b "F448 tokpath=$v" --steps 10 --warmup 3; done
for v in 1 0; do GVFI_F_TOKPATH=$v b "F4k tokpath=$v" --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --model f --steps 5 --warmup 2 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_f_448.md > /dev/null; rm -rf $O/prof; head -16 $O/kernel_stats_f_448.md | cut -c1-150; grep -E "token_path" $O/kernel_stats_f_448.md | cut -c1-150
