# round-4 call 2: timestep-batched frame synthesis -- GPU suite without the live-CPU-oracle cases + the full bench line (A/B vs r4c1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=8 -k "not live_oracle" > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
grep -E "^(448x256|R |F |demo|2k_|4k_|demo2k|SNU|XTEST|CLI)|passed|failed|rc " $O/gpu_tests.log | cut -c1-220 > $O/gpu_parity.log; tail -4 $O/gpu_parity.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_all.json 2> $O/bench_all.err; echo "rc $?" >> $O/bench_all.err
GVFI_T_BATCH_PIX=0 timeout 600 python bench.py --steps 5 --warmup 2 --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8 --no-cpu-baseline > $O/bench_4k_nobatch.json 2>> $O/bench_all.err
tail -3 $O/bench_all.err
