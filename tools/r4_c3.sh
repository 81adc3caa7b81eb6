# round-4 call 3: where the time goes after timestep batching (kernel traces) + first PMC passes of the weights-direct
# recurrence kernel + refreshed PMC of the hot 3x3 kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c3; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
prof() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o run -- python bench.py --configs none --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/prof_$tag.log 2>&1
  python tools/rocpd_stats.py $O/prof_$tag $O/kernel_stats_$tag.md > /dev/null; rm -rf $O/prof_$tag; }
prof r_448
prof r_4k --batch 1 --height 2176 --width 4096 --ds 0.25 --n-interp 8
prof f_448 --model f
head -14 $O/kernel_stats_r_4k.md | cut -c1-160
pmcw() { n=$1; shift; rm -rf $O/pw_$n; RING_ONLY="gru zr 128+128" RING_WDIR_ONLY=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/pw_$n -o run -- python tools/ring_bench.py > $O/pw_$n.log 2>&1; }
pmcw sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pmcw sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
pmcw tcc TCC_HIT_sum TCC_MISS_sum
pmcw tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pmcw fetch FETCH_SIZE
pmcw write WRITE_SIZE
python tools/pmc_report.py ConvArgs2 $O/pw_sq $O/pw_sq2 $O/pw_tcc $O/pw_tcp $O/pw_fetch $O/pw_write > $O/pmc_wdir.txt 2>&1; cat $O/pmc_wdir.txt | cut -c60-200
tail -3 $O/pw_sq.log
rm -rf $O/pw_*/
pmc() { n=$1; shift; rm -rf $O/pmc_$n; ONLYP3=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$n -o run -- python tools/conv_bench.py bf16 "final.resblock 256->256 3x3 @256" > $O/pmc_$n.log 2>&1; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python tools/pmc_report.py p3x3 $O/pmc_mfma $O/pmc_fetch $O/pmc_write > $O/pmc_p3x3.txt 2>&1; cat $O/pmc_p3x3.txt | cut -c60-200
rm -rf $O/pmc_*/
