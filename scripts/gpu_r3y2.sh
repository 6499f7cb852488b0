timeout 1200 python -m pytest tests/test_specialize_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x -k navigation 2>&1 | tail -4
python - <<'P'
import torch, time, sys
sys.path.insert(0, ".")
from vectorizedmultiagentsimulator_amd.environment import make_env
for B in (32768, 24576, 49152):
  for spec in (False, True):
    env = make_env("navigation", num_envs=B, device="cuda:0", seed=0, n_agents=8, validate_actions=False, specialize=spec)
    for _ in range(100): env.step([env.get_random_action(a) for a in env.agents])
    acts = [torch.zeros_like(env.get_random_action(a)) for a in env.agents]
    env.bind(acts)
    for _ in range(200): env.step_bound()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000): env.step_bound()
    e1.record(); torch.cuda.synchronize()
    print("navigation", B, "specialize", spec, "lanes", env.world._get_backend().lanes_per_env, "specialized", env.world._get_backend().specialized, "step_bound_us", round(e0.elapsed_time(e1), 2))
P
