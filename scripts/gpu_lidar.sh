# LIDAR kernels: parity + bitwise tests, rates
mkdir -p gpurun_out/lidar
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_fused_gpu.py tests/test_scenarios_vs_reference.py -m gpu -q --timeout=300 -p no:cacheprovider -x -k "lidar or navigation or rays" > gpurun_out/lidar/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/lidar/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/lidar/pytest.log | cut -c1-300 | head -30
grep -E "^E  +" gpurun_out/lidar/pytest.log | cut -c1-300 | head -40
{
  timeout 120 python scripts/bench_lidar.py 65536
  timeout 120 python scripts/bench_lidar.py 8192
  ONLY=fused-eager timeout 120 python scripts/bench_env.py navigation 65536
  ONLY=fused-graph timeout 120 python scripts/bench_env.py navigation 65536
} 2>&1 | grep "^{" | cut -c1-420 > gpurun_out/lidar/rates.jsonl
cat gpurun_out/lidar/rates.jsonl
