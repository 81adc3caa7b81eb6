cd $GRAFT_REPO_ROOT; O=gpurun_out/r5o; mkdir -p $O; : > $O/ab_misc_lanes.txt
run() { v=$1; lab=$2; shift 2
  line=$(GVFI_MISC_LANES=$v timeout 400 python bench.py --configs none --no-cpu-baseline "$@" --details $O/tmp.json 2>$O/err.txt | tail -1)
  echo "GVFI_MISC_LANES=$v $lab $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' 2>/dev/null || tail -2 $O/err.txt)" >> $O/ab_misc_lanes.txt; }
for rep in 1 2 3; do for v in 0 1; do run $v r448 --steps 20 --warmup 5; done; done
cat $O/ab_misc_lanes.txt
