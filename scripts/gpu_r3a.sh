# round 3, call A: full GPU suite (with the 115-case reference suite and the new boundary tests), the cost of the exact
# broad phase inside the launch, and this box's baseline rates for the kernels to be worked on
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider --durations=15 > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3a/pytest.log | cut -c1-300 | head -40
grep -E "^E  +" gpurun_out/r3a/pytest.log | cut -c1-300 | head -40
timeout 600 python scripts/bench_exact.py 300 2>&1 | grep "^{" > gpurun_out/r3a/exact_cost.jsonl
cat gpurun_out/r3a/exact_cost.jsonl
{
for W in "football 131072" "football 16384" "balance 32768" "navigation 65536" "navigation 8192"; do
  QUEUES=1 python scripts/bench_world.py $W 500
done
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager python scripts/bench_env.py balance 32768
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
} 2>&1 | grep "^{" | cut -c1-600 > gpurun_out/r3a/rates.jsonl
cat gpurun_out/r3a/rates.jsonl
