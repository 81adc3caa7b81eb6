# round-4 call 1: the new default bench line (headline + every BASELINE configuration) on this round's box = baseline before kernel work
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c1; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_all.json 2> $O/bench_all.err; echo "rc $?" >> $O/bench_all.err
tail -c 6000 $O/bench_all.json; tail -5 $O/bench_all.err
