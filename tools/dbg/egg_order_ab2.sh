for cfg in "SBX_EGG_ORDER=0" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=0" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=4000" "SBX_EGG_ORDER=0" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=0" "SBX_EGG_ORDER=1 SBX_EGG_LONG_TICKS=4000"; do
echo "## $cfg"
env $cfg timeout 120 python tools/ab_time.py --app egg --width 1920 --height 1080 --reps 60 base 2>&1 | grep -v amdgpu.ids
env $cfg timeout 120 python tools/ab_time.py --app egg --width 3840 --height 2160 --reps 40 base 2>&1 | grep -v amdgpu.ids
done
