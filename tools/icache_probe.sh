#!/bin/bash
# tools/icache_probe.sh — run ON THE GPU BOX: instruction-cache counters of the kernels whose code is larger than the 64 KB instruction
# cache (k_planet<true, .> 72-80 KB, k_vinyl<true, .> 66-123 KB) beside one that fits (k_clouds, 17 KB) — VERDICT r5 #8.
# Output: gpurun_out/icache/icache.txt; copy into profiles/r06_icache.txt.
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache
mkdir -p $OUT
S=$OUT/icache.txt
echo "# counters rocprofv3 offers on this device with ICACHE / IFETCH in their names" > $S
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQC_INST[A-Z_0-9]*" | sort -u | tr '\n' ' ' >> $S
echo >> $S
cat > /tmp/icache_cases.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
for _ in range(20):
    R.render("clouds", 3840, 2160, 0.37)
torch.cuda.synchronize()
for app, w, h in [("clouds", 3840, 2160), ("planet", 7680, 4320), ("planet_atmosphere", 7680, 4320), ("vinyl", 3840, 2160), ("vinyl_gpu", 3840, 2160),
                  ("egg", 1920, 1080), ("atmosphere", 7680, 4320)]:
    buf = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    for _ in range(4):
        R.render(app, w, h, 0.37, out=buf)
    torch.cuda.synchronize()
    del buf
PY
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-28)
  rm -rf $OUT/$tag
  rocprofv3 --kernel-trace -f csv --pmc $pass -d $OUT/$tag -o pmc -- python /tmp/icache_cases.py > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
  echo "# rocprofv3 --kernel-trace --pmc $pass   (full-frame dispatches only)" >> $S
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" --largest-grid >> $S; else echo "(no counters; see $tag.log)" >> $S; tail -3 $OUT/$tag.log >> $S; fi
done
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
cat $S
