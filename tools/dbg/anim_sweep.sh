run() { env "$@" timeout 160 python tools/dbg/egg_anim_ab.py 2>&1 | tail -1 | sed "s/^/$* : /"; }
run SBX_TILE_ORDER=0
run SBX_TILE_ORDER=1
run SBX_TILE_ORDER=0 SYNC_EVERY=8
run SBX_TILE_ORDER=1 SYNC_EVERY=8
run SBX_TILE_ORDER=0
run SBX_TILE_ORDER=1
