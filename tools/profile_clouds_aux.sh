#!/bin/bash
# tools/profile_clouds_aux.sh — run ON THE GPU BOX: the instantiations of k_clouds away from the default aux block (general and y-z
# suns, coverage .7, SKY_SPHERE, long marches) and k_clouds<false,...> of SBX_APP_CLOUDS_SKY, which never had counters (VERDICT r4
# Weak #9): times (tools/time_clouds_aux.py), rocprofv3 --stats, and the VALU / SALU / LDS / binary64 counts per kernel.
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_clouds_aux
mkdir -p $OUT
S=$OUT/clouds_aux_pmc.txt
echo "# python tools/time_clouds_aux.py" > $S
python tools/time_clouds_aux.py >> $S 2>$OUT/time.log
echo "# SBX_APP_CLOUDS_SKY 3840x2160 (tools/time_apps.py)" >> $S
python tools/time_apps.py 2>/dev/null | grep -i "clouds_sky\|clouds_best\|clouds_ue4\|clouds_tex\|vinyl\|sdf_ao" >> $S
echo "# rocprofv3 --kernel-trace --stats -- python tools/time_clouds_aux.py" >> $S
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python tools/time_clouds_aux.py > $OUT/trace.log 2>&1
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -r cat >> $S
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-28)
  rocprofv3 --kernel-trace -f csv --pmc $pass -d $OUT/$tag -o pmc -- python tools/time_clouds_aux.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
  echo "# rocprofv3 --kernel-trace --pmc $pass -- python tools/time_clouds_aux.py   (largest grid per kernel)" >> $S
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid >> $S; else echo "(no counters)" >> $S; fi
done
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace -f csv --pmc $pass -d $OUT/sky -o pmc -- python -c "
import sys; sys.path.insert(0, '.')
import torch, shaderbox_amd
R = shaderbox_amd.Renderer(0)
for _ in range(6): R.render('clouds_sky', 3840, 2160, .37)
torch.cuda.synchronize()" > $OUT/sky.log 2>&1
  f=$(find $OUT/sky -name '*counter_collection.csv' | head -1)
  echo "# rocprofv3 --kernel-trace --pmc $pass -- 6 x SBX_APP_CLOUDS_SKY 3840x2160" >> $S
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid >> $S; else echo "(no counters)" >> $S; fi
done
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
cat $S | head -120
