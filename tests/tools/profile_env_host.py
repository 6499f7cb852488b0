"""Host-side cost of Environment.step (fused, eager): cProfile over 500 steps."""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
env = make_env(name, num_envs=32768, device="cuda:0", seed=0, validate_actions=False, **kw)
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(20):
    env.step(acts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    env.step(acts)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e6 * (t1 - t0) / 500:.1f} us/step, with final sync {1e6 * (t2 - t0) / 500:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    env.step(acts)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
