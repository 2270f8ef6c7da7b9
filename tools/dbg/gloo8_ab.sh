export HSA_ENABLE_IPC_MODE_LEGACY=0
for o in 0 1 0 1; do
SBX_TILE_ORDER=$o timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2963$o bench.py --gpus 8 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ORDER=$o', d['value'], d['ms_per_step'], d['value_pipelined'], d['exchange']['chosen'][:330])"
done
