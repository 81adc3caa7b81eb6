# HBM traffic of the dominant kernel (hot 256->256 3x3 convolution): FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Every pass is bounded by `timeout`.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { n=$1; shift; rm -rf gpurun_out/pm$n; ONLY256=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pm$n -- python tools/conv_bench.py bf16 "final.resblock 256" > gpurun_out/pm$n.log 2>&1; }
run f FETCH_SIZE
run w WRITE_SIZE
run h TCC_HIT_sum TCC_MISS_sum
python tools/pmc_report.py glds_kernel gpurun_out/pmf gpurun_out/pmw gpurun_out/pmh
