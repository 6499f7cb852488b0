# round 3, call B: the lane-compacted kernel - parity first, then its rate against the interpreter on the same box
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_compact_gpu.py -q --timeout=300 -p no:cacheprovider -x > gpurun_out/r3b/pytest_compact.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest_compact.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3b/pytest_compact.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3b/pytest_compact.log | cut -c1-300 | head -30
timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_hip_parity.py tests/test_env_gpu.py -q --timeout=300 -p no:cacheprovider -k "football or Football" > gpurun_out/r3b/pytest_football.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest_football.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3b/pytest_football.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3b/pytest_football.log | cut -c1-300 | head -30
{
for W in "football 131072" "football 16384" "football 1024"; do
  for CP in 0 1; do for Q in 1 2; do COMPACT=$CP QUEUES=$Q python scripts/bench_world.py $W 500; done; done
done
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager python scripts/bench_env.py football 16384
} 2>&1 | grep "^{" | cut -c1-600 > gpurun_out/r3b/rates.jsonl
cat gpurun_out/r3b/rates.jsonl
