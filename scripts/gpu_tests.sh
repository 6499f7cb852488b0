mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/pytest_gpu.log | cut -c1-300
grep -E "^E  +(Assertion|.*Error)" gpurun_out/pytest_gpu.log | cut -c1-330 | head -40
