mkdir -p gpurun_out/r3l
timeout 1200 python -m pytest tests/test_specialize_gpu.py tests/test_hip_parity.py -q --timeout=600 -p no:cacheprovider -k "special or Special" > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3l/pytest.log | cut -c1-300 | head
grep -E "^E  +" gpurun_out/r3l/pytest.log | cut -c1-300 | head -20
python scripts/bench_specialize.py 2>&1 | grep "^{\|Error" | tee gpurun_out/r3l/specialize_rates.jsonl
