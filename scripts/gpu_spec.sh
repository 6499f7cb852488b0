mkdir -p gpurun_out/quick
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "specialised or golden or plain_kernels or two_queues or several_queues" -p no:cacheprovider 2>&1 | tail -8
{
for SP in 1 0; do
  for Q in 1 2; do SPEC=$SP QUEUES=$Q python scripts/bench_world.py balance 32768 3000; done
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
python bench.py --no-cpu-baseline
} 2>&1 | grep "^{" | cut -c1-400 | tee gpurun_out/quick/spec_rates.jsonl
