timeout 1200 python -m pytest tests/test_env_fused_gpu.py tests/test_compact_gpu.py tests/test_env_gpu.py tests/test_specialize_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x 2>&1 | tail -4
python scripts/bench_rollout_env.py football 131072 50 | tail -1
python scripts/bench_rollout_env.py football 16384 50 | tail -1
ONLY=fused-eager python scripts/bench_env.py football 16384 | tail -1
python - <<'P'
import torch, time, sys
sys.path.insert(0, ".")
from vectorizedmultiagentsimulator_amd.environment import make_env
for spec in (False, True):
    env = make_env("navigation", num_envs=32768, device="cuda:0", seed=0, n_agents=8, validate_actions=False, specialize=spec)
    for _ in range(100): env.step([env.get_random_action(a) for a in env.agents])
    acts = [torch.zeros_like(env.get_random_action(a)) for a in env.agents]
    env.bind(acts)
    for _ in range(200): env.step_bound()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): env.step_bound()
    e1.record(); torch.cuda.synchronize()
    print("navigation 32768 specialize", spec, "lanes", env.world._get_backend().lanes_per_env, "specialized", env.world._get_backend().specialized, "step_bound_us", round(e0.elapsed_time(e1), 2))
P
