# APP_EGG 1920x1080 one launch at a time: hot-first (shipped) against the dispatch table "tiles measured long first, then hot-first"
for cfg in "SBX_EGG_ORDER=0" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=6000" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=10000" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=4000" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=2500" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=1" "SBX_EGG_ORDER=0"; do
echo "## $cfg"
env $cfg timeout 120 python tools/ab_time.py --app egg --width 1920 --height 1080 --reps 60 base 2>&1 | grep -v amdgpu.ids
done
