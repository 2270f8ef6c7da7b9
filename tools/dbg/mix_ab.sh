for m in 0 6 2 3 12 24; do
echo "## SBX_TILE_ORDER_MIX=$m"
SBX_TILE_ORDER_MIX=$m timeout 100 python tools/dbg/order_ab.py | tail -1
SBX_TILE_ORDER_MIX=$m timeout 100 python tools/dbg/order_strip_ab.py | tail -1
done
