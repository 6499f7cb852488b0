timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_fused_gpu.py tests/test_compact_gpu.py tests/test_round3_fixes_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x -k "exact or navigation or gave or barrier" 2>&1 | tail -3
export ACTIONS=zero
b() { python scripts/bench_bound.py navigation $1 | tail -1 | cut -c130-260; }
echo -n "first 16384: "; b 16384
echo -n "second 16384: "; b 16384
echo -n "8192: "; b 8192
echo -n "65536: "; b 65536
echo -n "65536 again: "; b 65536
unset ACTIONS
python scripts/bench_exact.py 300 2>&1 | grep "^{" > gpurun_out/r03_exact_in_launch_cost.jsonl; cat gpurun_out/r03_exact_in_launch_cost.jsonl | cut -c1-230
