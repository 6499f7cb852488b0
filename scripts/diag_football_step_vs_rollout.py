"""Football 5 v 5: what a single-step launch costs against a step of a K-step launch (physics only, and the whole
Environment.step).  The Environment.rollout lines time the Python call between two events: for small K the host's own
preparation (output allocation, argument blocks: ~0.2 ms) is inside the window - compare K = 5 with K = 20 for the per-step cost."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
env = make_env("football", num_envs=B, device="cuda:0", seed=0, validate_actions=False, n_blue_agents=5, n_red_agents=5, ai_red_agents=False)
for _ in range(30):
    env.step([env.get_random_action(a) for a in env.agents])
be = env.world._get_backend()
be.set_compact(1); be.set_queues(1)
g = torch.Generator(device="cuda:0").manual_seed(1)
def ev(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
snap = env.get_state()
for K in (1, 2, 5, 20):
    forces = torch.zeros(K, *be.agent_ft.shape, device="cuda:0")
    forces[:, :10, 0:2, :B] = (torch.rand(K, 10, 2, B, device="cuda:0", generator=g) * 2 - 1) * 0.3
    env.set_state(snap)
    print("physics: one launch of K=%d steps: %.1f us per step" % (K, ev(lambda: be.rollout(K, forces), K)))
env.set_state(snap)
forces = torch.zeros(20, *be.agent_ft.shape, device="cuda:0")
forces[:, :10, 0:2, :B] = (torch.rand(20, 10, 2, B, device="cuda:0", generator=g) * 2 - 1) * 0.3
print("physics: 20 launches of one step: %.1f us per step" % ev(lambda: be.step_n(20, forces), 20))
for K in (1, 2, 5, 20):
    acts = [(torch.rand(K, B, 2, device="cuda:0", generator=g) * 2 - 1) for _ in env.agents]
    env.set_state(snap)
    print("Environment.rollout K=%d: %.1f us per step" % (K, ev(lambda: env.rollout(acts), K)))
