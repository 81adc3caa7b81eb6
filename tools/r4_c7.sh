# round-4 call 7: side lane (flow-independent front ends beside the recurrence) A/B, CLI with 8 decode threads, --dry rehearsal
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; tail -2 $O/tests.log
b() { tag=$1; shift; timeout 300 python bench.py --configs none --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for v in 1 0 1 0; do GVFI_SIDE_BRANCH=$v b "448 side=$v" --steps 20 --warmup 5; done
for v in 1 0; do GVFI_SIDE_BRANCH=$v b "2k side=$v" --steps 5 --warmup 2 --batch 1 --height 1088 --width 2048 --ds 0.5 --n-interp 8; done
for v in 1 0; do GVFI_SIDE_BRANCH=$v b "4k side=$v" --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
GVFI_CLI_TIMING=1 timeout 300 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; cat $O/cli_bench_2k.txt | cut -c1-300
timeout 400 python bench.py --gpus 2 --dry --steps 3 --warmup 1 --batch 1 --height 1088 --width 2048 --ds 0.5 --n-interp 8 > $O/dry_r2k.json 2> $O/dry_r2k.err; tail -1 $O/dry_r2k.json | cut -c1-900; tail -2 $O/dry_r2k.err | cut -c1-300
