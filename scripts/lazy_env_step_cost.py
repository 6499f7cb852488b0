"""The one-launch Environment.step with fresh random actions (no validation), the reference's broad-phase rule on / off:
python scripts/lazy_env_step_cost.py [balance 32768]   (the lazy form's cost where no tile asks)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
for rep in range(2):
    for exact in (True, False):
        env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, exact_broad_phase=exact, **kw)
        cyc = [[env.get_random_action(a) for a in env.agents] for _ in range(25)]
        for k in range(300):
            env.step(cyc[k % 25])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 3000
        for k in range(n):
            env.step(cyc[k % 25])
        e1.record(); torch.cuda.synchronize()
        be = env.world._get_backend()
        print(json.dumps({"scenario": name, "num_envs": B, "exact": exact, "env_step_gpu_us": round(e0.elapsed_time(e1) / n * 1e3, 2),
                          "lazy_stats": be.lazy_stats() if exact else None}))
