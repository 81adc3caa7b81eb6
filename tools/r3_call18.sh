# start-offset A/B of the hot 3x3 kernel (GVFI_P3_SKEW = n x 8 k cycles on every other CU's first workgroup)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3s; mkdir -p $O
for i in 1 2; do for v in 0 1 2 4; do
  echo "R 448 skew=$v: $(GVFI_P3_SKEW=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])")"
done; done | tee $O/skew_448.txt
for v in 0 2 4; do
  echo "R 2K skew=$v: $(GVFI_P3_SKEW=$v timeout 200 python bench.py --height 1088 --width 2048 --ds 0.5 --n-interp 8 --batch 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])")"
done | tee $O/skew_2k.txt
