mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_env_fused_gpu.py tests/test_env_gpu.py -q --timeout=600 -p no:cacheprovider -k "balance" > gpurun_out/r3m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3m/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3m/pytest.log | cut -c1-300 | head
grep -E "^E  +" gpurun_out/r3m/pytest.log | cut -c1-300 | head -20
python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r3m/bench.json
python - <<'P'
import json
d = json.load(open("gpurun_out/r3m/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], d["repeats"])
e = d["env_step"]; print("env_step us", e["us_per_step"], "gpu", e["gpu_us_per_step"], "bound", e["bound"]["us_per_step"], e["bound"]["gpu_us_per_step"], "frac", e["roofline"]["frac"])
print("persistent", d["persistent_rollout"]["us_per_step"])
P
python scripts/bench_rollout_env.py balance 32768 100 | tail -1
