# round-4 call 9: fused flow-token path v2 (key / value rows and look-up taps in flight together) -- bit-identity + A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c9; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_f.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; tail -2 $O/tests.log
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline --model f "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in 1 0 1 0; do GVFI_F_TOKPATH=$v b "F448 tokpath=$v" --steps 10 --warmup 3; done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python bench.py --configs none --no-cpu-baseline --model f --steps 5 --warmup 2 > $O/prof.log 2>&1
python tools/rocpd_stats.py $O/prof $O/kernel_stats_f_448.md > /dev/null; rm -rf $O/prof; grep -E "token_path" $O/kernel_stats_f_448.md | cut -c1-150
