#!/bin/bash
# tools/pmc_ab.sh name1 name2 ... — run ON THE GPU BOX: PMC counters of the render kernel for A/B libraries built by
# tools/ab_build.py ('base' = the shipped library).  One counter group per rocprofv3 pass (kernel-trace + pmc only).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab
mkdir -p $OUT
APP=${SBX_AB_APP:-clouds}; W=${SBX_AB_W:-3840}; H=${SBX_AB_H:-2160}
# SBX_AB_GROUPS="ctr ctr;ctr ctr": counter groups of this run instead of the three below (one rocprofv3 pass each)
for name in "$@"; do
  echo "=== $name"
  i=0
  DEFAULT_GROUPS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA;SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS"
  IFS=';' read -ra GROUPS_ARR <<< "${SBX_AB_GROUPS:-$DEFAULT_GROUPS}"
  for grp in "${GROUPS_ARR[@]}"; do
    i=$((i+1))
    rm -rf $OUT/$name.$i
    rocprofv3 --kernel-trace -f csv --pmc $grp -d $OUT/$name.$i -o pmc -- python tools/ab_time.py --app $APP --width $W --height $H --reps 4 $name > $OUT/$name.$i.log 2>&1
    f=$(find $OUT/$name.$i -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid; else echo "(no counters; see log)"; tail -3 $OUT/$name.$i.log; fi
  done
done
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
