# specialised fused-environment kernels (transport epilogue, ingest-only prologue): bitwise tests + A/B rates
mkdir -p gpurun_out/specenv
timeout 600 python -m pytest tests/test_env_fused_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider -k "specialised or rollout or transport or navigation" > gpurun_out/specenv/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/specenv/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/specenv/pytest.log | cut -c1-300 | head -30
grep -E "^E  +" gpurun_out/specenv/pytest.log | cut -c1-300 | head -30
{
for S in 1 0; do
  SPEC=$S timeout 120 python scripts/bench_rollout_env.py transport 16384 100
  SPEC=$S ONLY=fused-graph timeout 120 python scripts/bench_env.py transport 16384
  SPEC=$S ONLY=fused-graph timeout 120 python scripts/bench_env.py navigation 8192
  SPEC=$S ONLY=fused-graph timeout 120 python scripts/bench_env.py navigation 65536
done
} 2>&1 | grep "^{" | cut -c1-420 > gpurun_out/specenv/rates.jsonl
cat gpurun_out/specenv/rates.jsonl
