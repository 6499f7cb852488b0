"""Cost of one vmas_world_step_n call of K steps on 1 / 2 HIP queues (fork + join included), HIP-event timed per call:
python scripts/bench_step_n.py balance 32768"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
for _ in range(20):
    env.step([env.get_random_action(a) for a in env.agents])
be = env.world._get_backend()
for K in (5, 10, 20, 50, 100, 400):
    row = {"scenario": name, "num_envs": B, "K": K}
    for q in (1, 2):
        be.set_queues(q)
        torch.cuda.synchronize()
        first = None
        ts = []
        for rep in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            be.step_n(K)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / K
            if first is None:
                first = t
            ts.append(t)
        ts.sort()
        row[f"q{q}_us_per_step_first_call"] = round(first, 2)
        row[f"q{q}_us_per_step_median"] = round(ts[len(ts) // 2], 2)
    print(json.dumps(row), flush=True)
