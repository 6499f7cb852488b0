mkdir -p gpurun_out
echo "== full collection, -k graph"; python -m pytest tests -m gpu -x -q -k graph 2>&1 | tail -3
echo "== env+parity files, -k graph"; python -m pytest tests/test_env_gpu.py tests/test_hip_parity.py -x -q -k graph 2>&1 | tail -3
echo "== full collection, -k 'graph and balance' with log"; AMD_LOG_LEVEL=3 python -m pytest tests -m gpu -x -q -k "graph and balance" > gpurun_out/amdlog.txt 2>&1; grep -n -i "capture" gpurun_out/amdlog.txt | head -20
grep -n -B30 -m1 "hipErrorStreamCaptureInvalidated\|hipErrorStreamCaptureUnsupported\|hipErrorStreamCaptureImplicit" gpurun_out/amdlog.txt | grep -i "hip[A-Z][A-Za-z]* *(" | tail -30
