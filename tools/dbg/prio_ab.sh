for l in "" prio300 prio200 prio100 prio30; do
echo "## lib=$l"
SBX_AB_LIB=$l timeout 100 python tools/dbg/order_ab.py | tail -1
SBX_AB_LIB=$l timeout 100 python tools/dbg/order_strip_ab.py | tail -1
done
