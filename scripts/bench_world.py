"""Physics-only rate (World.step launches enqueued from C, forces fixed) of any native scenario's world:
python scripts/bench_world.py football 131072 [steps].  Honours VMAS_ABLATE / --lanes like bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "football"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lanes = int(os.environ.get("LANES", "0"))
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
for _ in range(30):  # a few real steps so that the state is a typical mid-episode one
    env.step([env.get_random_action(a) for a in env.agents])
be = env.world._get_backend()
if lanes:
    be.set_lanes_per_env(lanes)
if os.environ.get("SPEC"):
    be.set_specialized(os.environ["SPEC"] != "0")
if os.environ.get("QUEUES"):
    be.set_queues(int(os.environ["QUEUES"]))
be.step_n(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
be.step_n(steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"scenario": name, "num_envs": B, "lanes": be.lanes_per_env, "queues": be.queues(steps), "specialized": be.specialized,
                  "lib": os.environ.get("VMAS_HIP_LIB", "libvmas_hip.so"), "ablate": os.environ.get("VMAS_ABLATE", "0"),
                  "world_step_us": round(dt * 1e6, 2), "env_steps_per_s": round(B / dt)}))
