#!/bin/bash
# One parametrised evidence script for the GPU box (replaces the per-call scripts of rounds 2-4).  Run through gpurun:
#   gpurun --timeout 2400 -- 'bash tools/evidence.sh <out-tag> <step> [<step> ...]'
# Outputs go to gpurun_out/<out-tag>/; copy what is to be judged into profiles/ (rNN_ prefix).  Steps:
#   tests            the whole GPU suite (-m gpu) + parity lines          tests-fast   the suite without the live-CPU-oracle cases
#   smoke            __graft_entry__.smoke()
#   bench            the default bench line (headline + every BASELINE configuration; two steps in flight, bench.py's default)
#                    (the shapes- / prof- / timeline- / hbm- steps run ONE step at a time: per-kernel figures, not throughput)
#   shapes-<cfg>     per-conv-shape table of one configuration (cfg: r448 r2k r4k f448 f4k)
#   prof-<cfg>       rocprofv3 --kernel-trace summary (tools/rocpd_stats.py) of one configuration
#   timeline-<cfg>   wall-clock structure of one step (first start / last end per kernel, GPU-busy union; tools/phase_timeline.py)
#   pmc-p3x3         PMC passes of the hot 3x3 kernel (SQ + GRBM | FETCH_SIZE | WRITE_SIZE, separate passes)
#   pmc-wdir         PMC passes of the weights-direct recurrence kernel (SQ x2 | TCC | TCP | FETCH_SIZE | WRITE_SIZE)
#   hbm-<cfg>        HBM bytes per kernel (FETCH_SIZE / WRITE_SIZE passes over a bench run, joined with prof-<cfg>'s durations)
#   cli-2k, cli-448  end-to-end CLI throughput incl. PNG decode and video writing (tools/cli_bench.py)
#   cli-2k-dry8      the 8-rank CLI result path rehearsed on one GPU (rank 0 real, 7 CPU stand-in ranks; per-rank host seconds)
#   ab-<ENVVAR>      same-box A/B of a 0/1 environment switch on the R and F 448x256 headlines (two repetitions each)
#   fpolicy          GIMM-VFI-F precision policies of the flow estimator against the hardest reference fixture (tools/f_policy_diag.py)
#   dry-<cfg>        bench.py --gpus 2 --dry: the sharded step's bookkeeping with real frame shapes on one GPU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
cfg_args() {
  case $1 in
    r448) echo "";;
    r2k) echo "--batch 1 --height 1088 --width 2048 --ds 0.5 --n-interp 8";;
    r4k) echo "--batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8";;
    f448) echo "--model f";;
    f4k) echo "--model f --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8";;
    r448fp32) echo "--precision fp32";;
  esac
}
for step in "$@"; do
  case $step in
    tests|tests-fast)
      K=""; [ $step = tests-fast ] && K='-k not(live_oracle)'
      timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -rP --durations=10 ${K:+-k "not live_oracle"} > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
      grep -E "^(448x256|R |F |demo|2k_|4k_|demo2k|SNU|XTEST|CLI|FAMILY)|passed|failed|rc " $O/gpu_tests.log | cut -c1-230 > $O/gpu_parity.log; tail -3 $O/gpu_parity.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log;;
    bench) ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/bench_full.json ) > $O/bench_line.json 2> $O/bench_all.err; echo "rc $?" >> $O/bench_all.err; tail -c 1700 $O/bench_line.json;;
    shapes-*) c=${step#shapes-}; timeout 400 python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 5 --warmup 2 $(cfg_args $c) --shapes $O/conv_shapes_$c.md > $O/bench_$c.json 2> $O/bench_$c.err; head -12 $O/conv_shapes_$c.md | cut -c1-160;;
    prof-*) c=${step#prof-}; timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o run -- python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 5 --warmup 2 $(cfg_args $c) > $O/prof_$c.log 2>&1
      python tools/rocpd_stats.py $O/prof_$c $O/kernel_stats_$c.md > /dev/null; rm -rf $O/prof_$c; head -14 $O/kernel_stats_$c.md | cut -c1-160;;
    timeline-*) c=${step#timeline-}; timeout 400 rocprofv3 --kernel-trace -d $O/tl_$c -o run -- python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 5 --warmup 2 $(cfg_args $c) > $O/tl_$c.log 2>&1
      python tools/phase_timeline.py $O/tl_$c prep_images $O/phase_timeline_$c.md > /dev/null; rm -rf $O/tl_$c; head -3 $O/phase_timeline_$c.md | cut -c1-220;;
    pmc-p3x3)
      p() { n=$1; shift; rm -rf $O/pmc_$n; ONLYP3=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$n -o run -- python tools/conv_bench.py bf16 "final.resblock 256->256 3x3 @256" > $O/pmc_$n.log 2>&1; }
      p mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS; p fetch FETCH_SIZE; p write WRITE_SIZE
      python tools/pmc_report.py p3x3 $O/pmc_mfma $O/pmc_fetch $O/pmc_write > $O/pmc_p3x3.txt 2>&1; cut -c60-200 $O/pmc_p3x3.txt; rm -rf $O/pmc_*/;;
    pmc-wdir)
      p() { n=$1; shift; rm -rf $O/pw_$n; RING_ONLY="gru zr 128+128" RING_WDIR_ONLY=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/pw_$n -o run -- python tools/ring_bench.py > $O/pw_$n.log 2>&1; }
      p sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
      p sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
      p tcc TCC_HIT_sum TCC_MISS_sum; p tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum; p fetch FETCH_SIZE; p write WRITE_SIZE
      python tools/pmc_report.py ConvArgs2 $O/pw_sq $O/pw_sq2 $O/pw_tcc $O/pw_tcp $O/pw_fetch $O/pw_write > $O/pmc_wdir.txt 2>&1; cut -c85-200 $O/pmc_wdir.txt; rm -rf $O/pw_*/;;
    hbm-*) c=${step#hbm-}       # needs kernel_stats_<cfg>.md of a prof-<cfg> step of the same tag (un-profiled durations)
      for ctr in FETCH_SIZE WRITE_SIZE; do rm -rf $O/hbm_$ctr; timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/hbm_$ctr -o run -- python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 3 --warmup 1 $(cfg_args $c) > $O/hbm_$ctr.log 2>&1; done
      python tools/pmc_table.py $O/kernel_stats_$c.md $O/hbm_table_$c.md $O/hbm_FETCH_SIZE $O/hbm_WRITE_SIZE > /dev/null; rm -rf $O/hbm_FETCH_SIZE $O/hbm_WRITE_SIZE; head -16 $O/hbm_table_$c.md | cut -c1-170;;
    cli-2k) GVFI_CLI_TIMING=1 timeout 300 python tools/cli_bench.py 33 2048 1088 8 0.5 > $O/cli_bench_2k.txt 2>&1; cut -c1-300 $O/cli_bench_2k.txt;;
    cli-2k-dry8) timeout 400 python tools/cli_bench.py 65 2048 1088 8 0.5 8 dry > $O/cli_bench_2k_dry8.txt 2>&1; cut -c1-400 $O/cli_bench_2k_dry8.txt;;
    cli-448) timeout 200 python tools/cli_bench.py 65 448 256 2 > $O/cli_bench_448.txt 2>&1; grep -E "video_Nx|CLI:" $O/cli_bench_448.txt | cut -c1-300;;
    ab-host)   # same-box A/B of the round-5 host-side changes (zero-once buffers + graph-static outputs) at 448x256 and 4K
      : > $O/ab_host.txt
      for rep in 1 2; do for val in 0 1; do for cfg in r448 r4k; do
        line=$(GVFI_ZERO_ONCE=$val GIMMVFI_STATIC_OUTPUTS=$val timeout 400 python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 10 --warmup 3 $(cfg_args $cfg) --details $O/ab_tmp.json 2>/dev/null | tail -1)
        echo "zero_once=static_outputs=$val $cfg $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/ab_host.txt
      done; done; done; cat $O/ab_host.txt;;
    ab-*)   # ab-<ENVVAR>: same-box A/B of a switch (0 / 1 / 0 / 1) on the R and F 448x256 headlines, graph replay only
      v=${step#ab-}; : > $O/ab_$v.txt
      for rep in 1 2; do for val in 0 1; do for mdl in r f; do
        line=$(env $v=$val timeout 300 python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 20 --warmup 5 --model $mdl --details $O/ab_tmp.json 2>/dev/null | tail -1)
        echo "$v=$val model=$mdl $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/ab_$v.txt
      done; done; done; cat $O/ab_$v.txt;;
    wdir-floor)   # what bounds the recurrence: the same graph with the weights-direct launches' K loop / epilogue skipped (garbage results)
      : > $O/wdir_floor.txt
      for dbg in 0 16 8 24 0; do for lanes in 2 1; do
        line=$(GVFI_WDIR_DBG=$dbg GVFI_RAFT_LANES=$lanes timeout 300 python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 20 --warmup 5 --details $O/ab_tmp.json 2>/dev/null | tail -1)
        echo "GVFI_WDIR_DBG=$dbg lanes=$lanes $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/wdir_floor.txt
      done; done; cat $O/wdir_floor.txt;;
    fpolicy-all) timeout 900 python tools/f_policy_diag.py "--policies=dec:f16;enc:f16,dec:f16;enc:f16,cost:f16,dec:f16" > $O/f_policy_all.txt 2>&1; cut -c1-250 $O/f_policy_all.txt | tail -14;;
    fpol-bench)   # speed of the candidate GIMM-VFI-F policies, graph replay, same box: 448x256 B=8 and 4K DS 0.25 8x
      : > $O/fpol_bench.txt
      for pol in "dec:f16" "f16" "enc:f16,cost:f16,dec:f16" "dec:f16" "f16"; do
        for cfg in f448 f4k; do
          line=$(timeout 400 python bench.py --in-flight 1 --configs none --no-cpu-baseline --steps 10 --warmup 3 $(cfg_args $cfg) --flow-precision "$pol" --details $O/ab_tmp.json 2>/dev/null | tail -1)
          echo "$cfg $pol $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O/fpol_bench.txt
        done; done; cat $O/fpol_bench.txt;;
    fpolicy) timeout 600 python tools/f_policy_diag.py demo2k_ds050 "--policies=dec:f16;enc,dec:f16;cost,dec:f16;enc,cost,dec:f16;enc:f16,cost:f16,dec:f16" > $O/f_policy.txt 2>&1; cut -c1-260 $O/f_policy.txt | tail -8;;
    dry-*) c=${step#dry-}; timeout 400 python bench.py --gpus 2 --dry --steps 3 --warmup 1 $(cfg_args $c) > $O/dry_$c.json 2> $O/dry_$c.err; tail -1 $O/dry_$c.json | cut -c1-700;;
    *) echo "unknown step $step";;
  esac
done
