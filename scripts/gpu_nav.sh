# navigation one-launch Environment.step: bitwise tests + rates
mkdir -p gpurun_out/nav
timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_hip_parity.py tests/test_scenarios_vs_reference.py -m gpu -q --timeout=300 -p no:cacheprovider -x -k "${K:-navigation or lidar or rays or graph or checkpoint}" > gpurun_out/nav/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/nav/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/nav/pytest.log | cut -c1-300 | head -30
grep -E "^E  +" gpurun_out/nav/pytest.log | cut -c1-300 | head -40
{
  ONLY=fused-eager timeout 120 python scripts/bench_env.py navigation 8192
  ONLY=fused-graph timeout 120 python scripts/bench_env.py navigation 8192
  ONLY=fused-eager timeout 120 python scripts/bench_env.py navigation 65536
} 2>&1 | grep "^{" | cut -c1-420 > gpurun_out/nav/rates.jsonl
cat gpurun_out/nav/rates.jsonl
