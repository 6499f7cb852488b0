"""Per-block timing of Environment.step to spot warm-up / clock effects."""
import sys, time
sys.path.insert(0, ".")
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
print("one_launch", env._one_launch)
acts = [env.get_random_action(a) for a in env.agents]
for blk in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(250):
        env.step(acts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"block {blk}: issue {1e6*(t1-t0)/250:.1f} us/step, total {1e6*(t2-t0)/250:.1f} us/step", flush=True)
