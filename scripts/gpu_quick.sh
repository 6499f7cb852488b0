# quick A/B after a kernel change: GPU suite + a few rates
mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/quick/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/quick/pytest.log | cut -c1-300 | head -30
grep -E "^E  +" gpurun_out/quick/pytest.log | cut -c1-300 | head -30
{
for W in "balance 32768" "transport 16384" "navigation 65536" "football 131072" "football 16384" "balance 1048576"; do
  for Q in 1 2; do QUEUES=$Q python scripts/bench_world.py $W 1000; done
done
ONLY=fused-eager python scripts/bench_env.py balance 32768; ONLY=fused-graph python scripts/bench_env.py balance 32768
python scripts/bench_rollout_env.py balance 32768 100
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
} 2>&1 | grep "^{" | cut -c1-420 > gpurun_out/quick/rates.jsonl
cat gpurun_out/quick/rates.jsonl
