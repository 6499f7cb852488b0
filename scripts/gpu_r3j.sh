mkdir -p gpurun_out/r3j
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --durations=8 > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3j/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3j/pytest.log | cut -c1-300 | head -30
grep -E "^E  +" gpurun_out/r3j/pytest.log | cut -c1-300 | head -30
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tee gpurun_out/r3j/bench_driver_args.json | cut -c1-1500
{
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager ACTIONS=fixed python scripts/bench_env.py football 131072
for F in random fixed; do for CP in 0 1; do FORCES=$F COMPACT=$CP QUEUES=1 python scripts/bench_world.py football 131072 100; done; done
} 2>&1 | grep "^{" | cut -c1-500 | tee gpurun_out/r3j/rates.jsonl
