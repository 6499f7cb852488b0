"""End-to-end Environment.step rate (action ingest + physics + LIDAR + observation/reward/done):
tensor-op path vs fused kernels, eager vs HIP graph."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
if os.environ.get("NAV_TILES"):  # A/B: navigation's one-launch step up to this many tiles per CU
    from vectorizedmultiagentsimulator_amd import fused as _F
    _F.NavigationPost.ONE_LAUNCH_MAX_TILES_PER_CU = int(os.environ["NAV_TILES"])
only = os.environ.get("ONLY")  # e.g. ONLY=fused-eager
for fused in (False, True):
    for graph in (False, True):
        if only and only != f"{'fused' if fused else 'plain'}-{'graph' if graph else 'eager'}":
            continue
        env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, graph=graph, fused=fused, **kw)
        if os.environ.get("SPEC") == "0":  # A/B: the interpreter instead of the world-specialised kernel
            env.world._get_backend().set_specialized(False)
        # ACTIONS=random (default; SURVEY.md 8d: u ~ U(-u_range, u_range) per step and agent, pre-generated): a pool of 64
        # action sets cycled; ACTIONS=fixed: one set held for the whole run (rounds 1-2: bodies pile up against the walls)
        pool = 1 if os.environ.get("ACTIONS", "random") == "fixed" else 64
        acts = [[env.get_random_action(a) for a in env.agents] for _ in range(pool)]
        t_warm, k = time.perf_counter(), 0
        while k < (300 if (graph or fused) else 5) or time.perf_counter() - t_warm < 0.25:  # (one-time costs, and the clocks
            env.step(acts[k % pool])                                                      #  of a just-started process)
            k += 1
        torch.cuda.synchronize()
        n = 1000 if (graph or fused) else 30
        t0 = time.perf_counter()
        for k in range(n):
            env.step(acts[k % pool])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({"scenario": name, "num_envs": B, "specialized": env.world._get_backend().specialized, "fused": fused, "graph": graph, "actions": os.environ.get("ACTIONS", "random"), "env_step_us": round(dt * 1e6, 2),
                          "env_steps_per_s": round(B / dt)}), flush=True)
