import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from vectorizedmultiagentsimulator_amd.environment import make_env
for mode in ("held", "random"):
    env = make_env("balance", num_envs=32768, device="cuda:0", seed=0, validate_actions=False, n_agents=4)
    be = env.world._get_backend()
    acts = [env.get_random_action(a) for a in env.agents]
    if mode == "held":
        env.bind(acts)
        for _ in range(2300): env.step_bound()
    else:
        for _ in range(2300): env.step([env.get_random_action(a) for a in env.agents])
    torch.cuda.synchronize()
    print(mode, be.lazy_stats())
