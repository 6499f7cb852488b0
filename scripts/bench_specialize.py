"""World.step rate of a world WITHOUT a built-in specialisation, on the interpreter and on the kernel compiled for it at run
time (specialize.py): python scripts/bench_specialize.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env

for name, B, kw in (("balance", 32768, dict(n_agents=3)), ("balance", 4096, dict(n_agents=3)), ("transport", 16384, dict(n_packages=2))):
    out = {"scenario": name, "kw": kw, "num_envs": B}
    for spec in (False, True):
        env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, specialize=spec, **kw)
        pool = [[env.get_random_action(a) for a in env.agents] for _ in range(32)]
        for k in range(30):
            env.step(pool[k % 32])
        be = env.world._get_backend()
        assert be.specialized == spec
        be.set_queues(1)
        be.step_n(200)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            be.step_n(1000)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 1000)
        out["world_step_us_" + ("runtime_spec" if spec else "interpreter")] = round(best * 1e6, 2)
        env.bind(pool[0])
        for _ in range(100):
            env.step_bound()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            env.step_bound()
        torch.cuda.synchronize()
        out["env_step_bound_us_" + ("runtime_spec" if spec else "interpreter")] = round((time.perf_counter() - t0) / 1000 * 1e6, 2)
    print(json.dumps(out), flush=True)
