#!/bin/bash
# tools/profile_gpu.sh <tag> [bench args...] — run ON THE GPU BOX (via gpurun).
# 1) rocprofv3 --kernel-trace --stats of bench.py  -> per-kernel average duration
# 2) separate --pmc passes (never combined with trace domains other than kernel-trace):
#      VALU issue / utilisation, wave occupancy & stalls, HBM write/fetch bytes
# Summaries land in gpurun_out/<tag>/summary.txt ; copy what should be judged into profiles/.
set -u
TAG=${1:-prof}; shift || true
ARGS=${@:---steps 5 --warmup 1 --no-cpu-baseline}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "# bench (unprofiled)" > $OUT/summary.txt
python bench.py $ARGS 2>/dev/null | tail -1 >> $OUT/summary.txt
echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" >> $OUT/summary.txt
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/trace.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -r cat >> $OUT/summary.txt
pmc() {
  name=$1; shift
  echo "# rocprofv3 --kernel-trace --pmc $* -- python bench.py $ARGS" >> $OUT/summary.txt
  rocprofv3 --kernel-trace -f csv --pmc $* -d $OUT/pmc_$name -o pmc -- python bench.py $ARGS > $OUT/pmc_$name.log 2>&1
  f=$(find $OUT/pmc_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid >> $OUT/summary.txt; else echo "(no counter file; see pmc_$name.log)" >> $OUT/summary.txt; tail -5 $OUT/pmc_$name.log >> $OUT/summary.txt; fi
}
pmc valu SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
pmc busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
pmc derived VALUBusy VALUUtilization
pmc wr WRITE_SIZE
pmc rd FETCH_SIZE
# keep only small files for the merge back
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
cat $OUT/summary.txt
