for o in 0 1 2 0 1 2; do SBX_TILE_ORDER=$o timeout 300 python bench.py --emulate-ranks 8 --app clouds 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['emulated'][0]
print('ORDER=$o', d['value'], e['exchange'], e['relief'], e['per_rank_ms'], {k:(v['root_ms'],v['slowest_peer_ms']) for k,v in e['exchanges_tried'].items()})"; done
