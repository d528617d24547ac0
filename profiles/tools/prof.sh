# rocprofv3 summaries for profiles/rNN (kernel-trace stats; separate PMC passes — never combined with sys / hip traces). Run on a GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/tools/prof.sh r03'
# then summarise the passes under gpurun_out/<tag>prof (find ... counter_collection.csv) with profiles/summarize_pmc.py / summarize_sq.py.
set -x
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}prof; mkdir -p $O
# one context, eager launches (every kernel a separate dispatch), the bench's own input pool: the 8 distinct pairs in rotation
# (round 5: --load-hint busy — the profiled context runs alone, the headline runs four: pin the kernels the busy device picks, i.e. the 64-query packet walk)
B="python $R/bench.py --steps 24 --warmup 8 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round --load-hint busy"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s1 -- $B > $O/stats.log 2>&1
# the same on the nominal pair alone (--pool 1: what rounds 1 and 2 profiled)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats0 -o s0 -- $B --pool 1 > $O/stats0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe -o pipe -- python $R/bench.py --pipeline-only > $O/pipe.log 2>&1
B2="python $R/bench.py --steps 8 --warmup 1 --no-cpu --no-config5 --no-pipeline --streams 1 --no-graph --single-round --load-hint busy"   # PMC passes: the bench's own pool, every pair once in the timed steps (round 5: was --pool 1)
export ROLO_PROF_COMMAND="$B2"   # (ROLO_PROF_COMMIT: the box has no .git — export it from the authoring side: gpurun -- 'ROLO_PROF_COMMIT=<hash> bash profiles/tools/prof.sh r05')
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $B2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $B2 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o q -- $B2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -o q2 -- $B2 > $O/pmc_sq2.log 2>&1
# the same SQ pass on the nominal pair alone: the launch profiles/tools/wavestats.py instruments (s_memrealtime per wavefront) — the two wave lifetimes must agree
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq0 -o q0 -- $B2 --pool 1 > $O/pmc_sq0.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc_grbm -o g -- $B2 --pool 1 > $O/pmc_grbm.log 2>&1
# round 6: an L2 picture of the neighbour search — TCC hits / misses / requests of the same single-context command (rocprofv3 serialises the dispatches while it collects
# counters, so "the walk beside another context's LM chain" cannot be collected this way: that regime is measured by event-timed walks, profiles/tools/concurrency.py)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o t -- $B2 > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc2 -o t2 -- $B2 > $O/pmc_tcc2.log 2>&1
cd $R
python profiles/tools/tcc_summary.py $(find $O/pmc_tcc -name "*counter_collection.csv" | head -1) $(find $O/pmc_tcc2 -name "*counter_collection.csv" | head -1) > $O/l2_counters.csv 2>> $O/pmc_tcc.log
python profiles/tools/codeobj_notes.py rolo_amd/librolo_hip.so > $O/codeobj_notes.csv
python profiles/summarize_sq.py $(find $O/pmc_sq0 -name "*counter_collection.csv" | head -1) $(find $O/pmc_grbm -name "*counter_collection.csv" | head -1) --notes $O/codeobj_notes.csv > $O/sq_counters_pair0.csv
# issue rate of the instructions the hot kernels are made of (cycles per wavefront instruction per SIMD): the constant of bench.py's valu_issue block
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate profiles/tools/valu_rate.hip > $O/valu_rate.log 2>&1 && /tmp/valu_rate > $O/valu_rate.txt 2>> $O/valu_rate.log
# per-wavefront s_memrealtime records of the packet walk on the nominal pair (instrumented build, if it was shipped with the snapshot)
if [ -f rolo_amd/librolo_hip_knnstats.so ]; then ROLO_KNN_SUB=0 ROLO_HIP_LIB=$R/rolo_amd/librolo_hip_knnstats.so python profiles/tools/wavestats.py > $O/knn_wavestats.txt 2>&1; fi
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F $W $O/pmc_traffic.json
python profiles/summarize_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq2 -name "*counter_collection.csv" | head -1) --notes $O/codeobj_notes.csv > $O/sq_counters.csv
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1_nograph.csv
cp $(find $O/stats0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_streams1_nograph_pair0.csv
cp $(find $O/pipe -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pipeline.csv
# keep only the summaries in the merge-back (the raw traces are tens of MB)
rm -rf $O/stats $O/stats0 $O/pipe $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 $O/pmc_sq0 $O/pmc_grbm $O/pmc_tcc $O/pmc_tcc2 $R/gpurun_out/wave_rec_*.npy
ls -la $O; head -12 $O/kernel_stats_streams1_nograph.csv; head -8 $O/sq_counters.csv
