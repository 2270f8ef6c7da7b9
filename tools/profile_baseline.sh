#!/bin/bash
# tools/profile_baseline.sh [tag] — run ON THE GPU BOX: the rocprofv3 evidence of every BASELINE.json kernel from the shipped
# library (C2 k_egg 1920x1080, C3 k_raytracer 4K, C4 k_clouds 4K, C5 k_atmosphere / k_planet 8K):
#   1) --kernel-trace --stats of tools/time_baseline.py                 -> average duration per kernel
#   2) one --pmc pass per counter group (kernel-trace only beside it)    -> per-launch counters, largest grid only
#   3) --kernel-trace of `bench.py --streams 3` and `--streams 1`: begin / end timestamps of the headline's launches
#      (what makes ms_per_step < kernel_ms with frames in flight)
# Output: gpurun_out/<tag>/apps_pmc.txt, gpurun_out/<tag>/streams3_trace.txt; copy into profiles/.
set -u
TAG=${1:-r06_baseline}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
S=$OUT/apps_pmc.txt
if [ "${SBX_PROFILE_ONLY_TRACE:-0}" != "1" ]; then
echo "# python tools/time_baseline.py (unprofiled, HIP events, median of 8 serial launches)" > $S
python tools/time_baseline.py >> $S 2>$OUT/time.log
echo "# rocprofv3 --kernel-trace --stats -- python tools/time_baseline.py" >> $S
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python tools/time_baseline.py > $OUT/trace.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -r cat >> $S
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
            "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT" \
            "VALUBusy VALUUtilization" "WRITE_SIZE" "FETCH_SIZE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-28)
  rocprofv3 --kernel-trace -f csv --pmc $pass -d $OUT/$tag -o pmc -- python tools/time_baseline.py 12 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
  echo "# rocprofv3 --kernel-trace --pmc $pass -- python tools/time_baseline.py 12   (full-frame dispatches only)" >> $S
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid >> $S; else echo "(no counters; see $tag.log)" >> $S; tail -3 $OUT/$tag.log >> $S; fi
done
fi
# 3) the headline with three and with one frame in flight: launch intervals of k_clouds
T=$OUT/streams3_trace.txt
: > $T
for ns in 3 1; do
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/bench_s$ns -o t -- python bench.py --steps 20 --warmup 5 --streams $ns --no-cpu-baseline --pmc off --no-other-configs --sustained-seconds 0 > $OUT/bench_s$ns.log 2>&1
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --streams $ns --no-cpu-baseline --pmc off --no-other-configs --sustained-seconds 0" >> $T
  tail -1 $OUT/bench_s$ns.log | cut -c1-600 >> $T
  find $OUT/bench_s$ns -name '*kernel_stats.csv' | head -1 | xargs -r head -4 >> $T
  f=$(find $OUT/bench_s$ns -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py "$f" k_clouds >> $T
done
find $OUT -name '*.csv' -size +1M -delete
find $OUT -name '*.db' -delete
[ -f $S ] && cat $S | head -150
cat $T
