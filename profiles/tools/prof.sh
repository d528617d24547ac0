# rocprofv3 summaries for profiles/rNN (kernel-trace stats; separate PMC passes — never combined with sys / hip traces). Run on a GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/tools/prof.sh r03'
# then summarise the passes under gpurun_out/<tag>prof (find ... counter_collection.csv) with profiles/summarize_pmc.py / summarize_sq.py.
set -x
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}prof; mkdir -p $O
# one context, eager launches (every kernel a separate dispatch), the bench's own input pool: the 8 distinct pairs in rotation
B="python $R/bench.py --steps 24 --warmup 8 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s1 -- $B > $O/stats.log 2>&1
# the same on the nominal pair alone (--pool 1: what rounds 1 and 2 profiled)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats0 -o s0 -- $B --pool 1 > $O/stats0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o pipe -- python $R/bench.py --pipeline-only > $O/pipe.log 2>&1
B2="python $R/bench.py --steps 8 --warmup 1 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round"   # PMC passes: the bench's own pool, every pair once in the timed steps (round 5: was --pool 1)
export ROLO_PROF_COMMAND="$B2"   # (ROLO_PROF_COMMIT: the box has no .git — export it from the authoring side: gpurun -- 'ROLO_PROF_COMMIT=<hash> bash profiles/tools/prof.sh r05')
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $B2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $B2 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $B2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o q2 -- $B2 > $O/pmc_sq2.log 2>&1
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F $W $O/pmc_traffic.json
python profiles/summarize_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq2 -name "*counter_collection.csv" | head -1) > $O/sq_counters.csv
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1_nograph.csv
cp $(find $O/stats0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1_nograph_pair0.csv
cp $(find $O/pipe -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pipeline.csv
# keep only the summaries in the merge-back (the raw traces are tens of MB)
rm -rf $O/stats $O/stats0 $O/pipe $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2
ls -la $O; head -12 $O/kernel_stats_streams1_nograph.csv; head -8 $O/sq_counters.csv
