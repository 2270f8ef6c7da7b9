#!/bin/bash
# tools/profile_apps.sh — run ON THE GPU BOX: VALUBusy / VALUUtilization / instruction counts of every app kernel
# at its BASELINE size (separate --pmc passes with kernel-trace only).  Output: gpurun_out/apps_pmc.txt
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/apps_pmc
mkdir -p $OUT
: > gpurun_out/apps_pmc.txt
for pass in "VALUBusy VALUUtilization" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-24)
  rocprofv3 --kernel-trace -f csv --pmc $pass -d $OUT/$tag -o pmc -- python tools/time_apps.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
  echo "# --pmc $pass" >> gpurun_out/apps_pmc.txt
  [ -n "$f" ] && python tools/pmc_summary.py "$f" --largest-grid >> gpurun_out/apps_pmc.txt
done
find $OUT -name '*.csv' -size +1M -delete
cat gpurun_out/apps_pmc.txt
