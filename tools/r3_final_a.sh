# round-3 evidence call A: the whole GPU suite + smoke
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r3fa; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=10 > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|R |F |demo|2k_|4k_|demo2k|SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log | cut -c1-220 > $O/gpu_parity.log; tail -3 $O/gpu_parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
