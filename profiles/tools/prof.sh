# rocprofv3 summaries for profiles/r02 (kernel-trace stats; separate PMC passes)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02prof; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s1 -- $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o pipe -- python $R/bench.py --pipeline-only > $O/pipe.log 2>&1
B2="python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $B2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $B2 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $B2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o q2 -- $B2 > $O/pmc_sq2.log 2>&1
find $O -name "*.csv" | head -30; du -sh $O
