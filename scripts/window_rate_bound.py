"""Per-window rate of the bound navigation step over a long run (is the rate a property of the first second?)."""
import sys, time, torch
sys.path.insert(0, ".")
from vectorizedmultiagentsimulator_amd.environment import make_env
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
W = int(sys.argv[2]) if len(sys.argv) > 2 else 20
env = make_env("navigation", num_envs=B, device="cuda:0", seed=0, n_agents=8, validate_actions=False)
for _ in range(100): env.step([env.get_random_action(a) for a in env.agents])
acts = [torch.zeros_like(env.get_random_action(a)) for a in env.agents]
env.bind(acts)
for _ in range(300): env.step_bound()
torch.cuda.synchronize()
out = []
for w in range(W):
    t0 = time.perf_counter()
    for _ in range(2000): env.step_bound()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 2000 * 1e6, 1))
print(B, "us/step per 2000-step window:", out)
