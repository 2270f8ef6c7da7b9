#!/bin/bash
# tools/pmc_ab_mem.sh name1 name2 ... — run ON THE GPU BOX: HBM traffic (WRITE_SIZE, FETCH_SIZE: separate passes), instruction
# counts and per-kernel durations of A/B libraries built by tools/ab_build.py ('base' = the shipped library).
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_mem
mkdir -p $OUT
APP=${SBX_AB_APP:-clouds}; W=${SBX_AB_W:-3840}; H=${SBX_AB_H:-2160}
for name in "$@"; do
  echo "=== $name"
  i=0
  for grp in "WRITE_SIZE" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $OUT/$name.$i
    rocprofv3 --kernel-trace -f csv --pmc $grp -d $OUT/$name.$i -o pmc -- python tools/ab_time.py --app $APP --width $W --height $H --reps 4 $name > $OUT/$name.$i.log 2>&1
    f=$(find $OUT/$name.$i -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid; else echo "(no counters; see log)"; tail -3 $OUT/$name.$i.log; fi
  done
  rm -rf $OUT/$name.t
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$name.t -o kt -- python tools/ab_time.py --app $APP --width $W --height $H --reps 4 $name > $OUT/$name.t.log 2>&1
  f=$(find $OUT/$name.t -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -6 "$f" | cut -c1-160
done
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
