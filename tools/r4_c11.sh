# round-4 call 11: LDS-staged cost_embed1 -- kernel test, A/B at F 448 / F 4K
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c11; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_f.py tests/test_gimmvfi_f.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline --model f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in new old new old; do if [ $v = old ]; then export GVFI_COST_EMBED_LDS0=1; else unset GVFI_COST_EMBED_LDS0; fi; b "F448 cost_embed=$v" --steps 10 --warmup 3; done
for v in new old; do if [ $v = old ]; then export GVFI_COST_EMBED_LDS0=1; else unset GVFI_COST_EMBED_LDS0; fi; b "F4k cost_embed=$v" --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
unset GVFI_COST_EMBED_LDS0
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --model f --steps 5 --warmup 2 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_f_448.md > /dev/null; rm -rf $O/prof; grep -E "cost_embed1|total kernel" $O/kernel_stats_f_448.md | cut -c1-170
