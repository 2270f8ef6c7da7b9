export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo n1 rc=$?
for n in 2 8; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --backend gloo --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_gloo_$n.json 2> gpurun_out/bench_gloo_$n.err; echo gloo$n rc=$?
done
timeout 600 python bench.py --emulate-ranks 8 > gpurun_out/bench_emulated8.json 2> gpurun_out/bench_emulated8.err; echo emu rc=$?
