mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_env_fused_gpu.py -k "compact or football" -q --timeout=300 -p no:cacheprovider > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3k/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3k/pytest.log | cut -c1-300 | head
grep -E "^E  +" gpurun_out/r3k/pytest.log | cut -c1-300 | head
{
for F in random fixed; do for CP in 0 1; do FORCES=$F COMPACT=$CP QUEUES=1 python scripts/bench_world.py football 131072 100; done; done
FORCES=random COMPACT=1 QUEUES=2 python scripts/bench_world.py football 131072 100
FORCES=random COMPACT=1 QUEUES=1 python scripts/bench_world.py football 16384 100
} 2>&1 | grep "^{" | cut -c1-500 | tee gpurun_out/r3k/rates.jsonl
