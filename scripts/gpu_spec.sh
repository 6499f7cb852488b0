mkdir -p gpurun_out/quick
{
for W in "football 131072" "football 16384" "navigation 8192"; do
  for SP in 1 0; do SPEC=$SP python scripts/bench_world.py $W 1000; done
done
python scripts/bench_rollout_env.py balance 32768 100
} 2>&1 | grep "^{" | cut -c1-300 | tee gpurun_out/quick/spec_rates2.jsonl
