# same-box A/B of the round-5 launch-sequence switches: GVFI_ENC_LANES (the two encoders of the flow estimator in parallel) and
# GVFI_POST_LANES (the flow-independent work behind the recurrence beside the motion path).  usage (GPU box): bash tools/ab_lanes.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r5l}; mkdir -p $O; : > $O/ab_lanes.txt
run() {  # enc post label args...
  e=$1; p=$2; lab=$3; shift 3
  line=$(GVFI_ENC_LANES=$e GVFI_POST_LANES=$p timeout 400 python bench.py --configs none --no-cpu-baseline "$@" --details $O/tmp.json 2>$O/err.txt | tail -1)
  echo "enc=$e post=$p $lab $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null || tail -2 $O/err.txt)" >> $O/ab_lanes.txt
}
for rep in 1 2; do
  for c in "0 0" "1 0" "1 1" "0 1"; do run $c r448 --steps 20 --warmup 5; done
  for c in "0 0" "1 0" "1 1"; do run $c f448 --steps 20 --warmup 5 --model f; done
done
for c in "0 0" "1 1"; do run $c r4k --steps 10 --warmup 3 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
for c in "0 0" "1 1"; do run $c f4k --steps 10 --warmup 3 --model f --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8; done
cat $O/ab_lanes.txt
