mkdir -p gpurun_out/quick
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_env_fused_gpu.py tests/test_env_gpu.py -m gpu -q -k "specialised or rollout or one_launch or golden or several_queues or checkpoint" -p no:cacheprovider 2>&1 | tail -12
{
python scripts/bench_rollout_env.py balance 32768 100
ONLY=fused-eager python scripts/bench_env.py balance 32768; ONLY=fused-graph python scripts/bench_env.py balance 32768
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
} 2>&1 | grep "^{" | cut -c1-1500 | tee gpurun_out/quick/spec_rates.jsonl
