"""World.step of a box-box world (transport with two packages: the level-2 kernel), physics only."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = make_env("transport", num_envs=B, device="cuda:0", seed=0, validate_actions=False, n_packages=2)
for _ in range(30):
    env.step([env.get_random_action(a) for a in env.agents])
be = env.world._get_backend()
be.step_n(50); torch.cuda.synchronize()
t0 = time.perf_counter(); be.step_n(300); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 300
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(200): env.step(acts)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): env.step(acts)
torch.cuda.synchronize(); de = (time.perf_counter() - t0) / 500
print(json.dumps({"scenario": "transport n_packages=2", "num_envs": B, "lanes": be.lanes_per_env, "world_step_us": round(dt * 1e6, 2),
                  "env_step_us": round(de * 1e6, 2), "one_launch": bool(env._one_launch)}))
