mkdir -p gpurun_out/r3p
timeout 1500 python -m pytest tests/test_env_fused_gpu.py tests/test_env_gpu.py tests/test_scenarios_vs_reference.py -q --timeout=600 -p no:cacheprovider -m gpu -k "navigation" > gpurun_out/r3p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3p/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3p/pytest.log | cut -c1-300 | head
grep -E "^E  +" gpurun_out/r3p/pytest.log | cut -c1-300 | head -20
for B in 8192 16384 65536; do
  for T in 1 64; do echo "tiles=$T"; NAV_TILES=$T python scripts/bench_bound.py navigation $B | tail -1; done
done
