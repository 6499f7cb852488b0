"""Cost of the reference's exact batch-global broad phase INSIDE the step launch (VmasStepArgs.exact_broad_phase: pair
bits ORed by every tile + a grid-wide barrier per substep) against the default per-environment evaluation, World.step
physics only, at the batch sizes where the in-kernel form applies (at most one 64-environment tile per CU).
python scripts/bench_exact.py  ->  one JSON line per (scenario, batch)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env

KW = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name in ("balance", "transport", "navigation", "football"):
    for B in (256, 1024, 4096, 8192, 16384):
        env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **KW[name])
        for _ in range(20):
            env.step([env.get_random_action(a) for a in env.agents])
        be = env.world._get_backend()
        out = {"scenario": name, "num_envs": B, "lanes": be.lanes_per_env, "specialized": be.specialized}
        for label, fn in (("per_env_us", lambda: be.step()), ("exact_in_launch_us", lambda: be.step_exact()),
                          ("per_env_interpreter_us", None)):
            if fn is None:  # exact steps run the interpreter (PLAIN == 0): the fair twin of the exact form
                be.set_specialized(False)
                fn = lambda: be.step()  # noqa: E731
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps)
            out[label] = round(best * 1e6, 2)
        out["exact_status"] = be.exact_status()
        print(json.dumps(out), flush=True)
        del env, be
