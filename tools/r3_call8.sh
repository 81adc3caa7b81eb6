cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3h; mkdir -p $O
tools/microbench/icache > $O/icache.txt 2>&1; cat $O/icache.txt
