"""Experiment: the batch split over S streams (S independent HipWorlds of batch/S environments, one stream each).
Environments are independent, so World.step of the whole batch = S concurrent launches; the launch gap of one stream
overlaps the compute of the others.  python scripts/exp_streams.py [scenario] [envs] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
for S in (1, 2, 4, 8):
    for lanes in (0, 8, 16):
        envs = [make_env(name, num_envs=B // S, device="cuda:0", seed=i, validate_actions=False, **kw) for i in range(S)]
        for e in envs:
            for _ in range(10):
                e.step([e.get_random_action(a) for a in e.agents])
        bes = [e.world._get_backend() for e in envs]
        if lanes:
            for be in bes:
                be.set_lanes_per_env(lanes)
        streams = [torch.cuda.Stream() for _ in range(S)]
        torch.cuda.synchronize()
        def run(n):
            # interleave the enqueues chunk-wise so that no stream runs far ahead of the host
            for c in range(0, n, 50):
                for be, st in zip(bes, streams):
                    be.step_n(min(50, n - c), stream=st)
        run(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps({"scenario": name, "envs": B, "streams": S, "lanes": bes[0].lanes_per_env,
                          "us_per_world_step": round(dt * 1e6, 2), "env_steps_per_s": round(B / dt)}), flush=True)
        del envs, bes
