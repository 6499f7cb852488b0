mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_compact_gpu.py -q --timeout=300 -p no:cacheprovider > gpurun_out/r3d/pytest_compact.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest_compact.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3d/pytest_compact.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3d/pytest_compact.log | cut -c1-300 | head -20
timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_hip_parity.py tests/test_env_gpu.py tests/test_round3_fixes_gpu.py -q --timeout=300 -p no:cacheprovider -k "football or Football or round3" > gpurun_out/r3d/pytest_football.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest_football.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3d/pytest_football.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3d/pytest_football.log | cut -c1-300 | head -20
for W in "football 1024" "football 131072"; do python scripts/trace_compact.py $W 2>&1 | tail -11; done | tee gpurun_out/r3d/trace_compact.txt
{
for F in random fixed; do for W in "football 131072" "football 16384"; do
  for CP in 0 1; do FORCES=$F COMPACT=$CP QUEUES=1 python scripts/bench_world.py $W 100; done
  FORCES=$F COMPACT=1 QUEUES=2 python scripts/bench_world.py $W 100
done; done
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager python scripts/bench_env.py football 16384
} 2>&1 | grep "^{" | cut -c1-600 > gpurun_out/r3d/rates.jsonl
cat gpurun_out/r3d/rates.jsonl
