"""One-launch Environment.step at a given waves-per-tile geometry, specialised at run time for it:
python scripts/ab_env_lanes.py navigation 8192 8   (lanes 0: the library's choice with its built-in specialisation)"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "navigation"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
if lanes:
    kw["lanes_per_env"] = lanes
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, specialize=True if lanes else None, **kw)
be = env.world._get_backend()
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(100):
    env.step([env.get_random_action(a) for a in env.agents])
acts = [torch.zeros_like(a) for a in acts]
env.bind(acts)
for _ in range(300):
    env.step_bound()
torch.cuda.synchronize()
n, wins = 2000, []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        env.step_bound()
    e1.record(); torch.cuda.synchronize()
    wins.append(e0.elapsed_time(e1) / n * 1e3)
print(json.dumps({"scenario": name, "num_envs": B, "lanes": be.lanes_per_env, "specialized": be.specialized, "one_launch": env._one_launch, "step_bound_us": sorted(round(w, 2) for w in wins)}))
