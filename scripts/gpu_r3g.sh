mkdir -p gpurun_out/r3h
timeout 1200 python -m pytest tests/test_compact_gpu.py tests/test_env_fused_gpu.py tests/test_env_gpu.py tests/test_round3_fixes_gpu.py -q --timeout=300 -p no:cacheprovider > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3h/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3h/pytest.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3h/pytest.log | cut -c1-300 | head -20
{
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager python scripts/bench_env.py football 16384
REPS=3 python scripts/bench_rollout_env.py football 131072 20
REPS=5 python scripts/bench_rollout_env.py football 16384 50
} 2>&1 | grep "^{\|Error\|error" | cut -c1-600 > gpurun_out/r3h/rates.jsonl
cat gpurun_out/r3h/rates.jsonl
