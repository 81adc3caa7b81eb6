cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3g; mkdir -p $O
timeout 400 python tools/ring_bench.py --stamps > $O/ring_bench.txt 2>&1; tail -2 $O/ring_bench.txt | cut -c1-300
