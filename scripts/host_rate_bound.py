import sys, time, torch
sys.path.insert(0, ".")
from vectorizedmultiagentsimulator_amd.environment import make_env
for B in (16384, 65536):
    env = make_env("navigation", num_envs=B, device="cuda:0", seed=0, n_agents=8, validate_actions=False)
    for _ in range(100): env.step([env.get_random_action(a) for a in env.agents])
    acts = [torch.zeros_like(env.get_random_action(a)) for a in env.agents]
    env.bind(acts)
    for _ in range(300): env.step_bound()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(2000): env.step_bound()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(B, "enqueue us/step", round((t1 - t0) / 2000 * 1e6, 2), "total us/step", round((t2 - t0) / 2000 * 1e6, 2))
