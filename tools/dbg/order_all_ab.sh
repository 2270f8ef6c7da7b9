export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_round5.py -q -m gpu -k dispatch_order 2>&1 | tail -1
for o in 0 1 0 1; do
SBX_TILE_ORDER=$o timeout 200 python tools/dbg/order_ab.py 2>&1 | tail -1
SBX_TILE_ORDER=$o timeout 200 python tools/dbg/order_strip_ab.py 2>&1 | tail -1
SBX_TILE_ORDER=$o timeout 300 python bench.py --emulate-ranks 8 --app clouds 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['emulated'][0]
print('emulate ORDER=$o', d['value'], e['exchange'], e['relief'], e['per_rank_ms'])"
SBX_TILE_ORDER=$o timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2963$o bench.py --gpus 8 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gloo8 ORDER=$o', d['value'], d['ms_per_step'], d['value_pipelined'], d['exchange']['chosen'][:330])"
done
