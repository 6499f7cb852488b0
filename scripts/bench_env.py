"""End-to-end Environment.step rate (physics + LIDAR + observation/reward/done), eager vs HIP graph."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
for graph in (False, True):
    env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, graph=graph, **kw)
    acts = [env.get_random_action(a) for a in env.agents]
    for _ in range(5):
        env.step(acts)
    torch.cuda.synchronize()
    n = 200 if graph else 30
    t0 = time.perf_counter()
    for _ in range(n):
        env.step(acts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"scenario": name, "num_envs": B, "graph": graph, "env_step_us": dt * 1e6, "env_steps_per_s": B / dt}))
