"""LIDAR kernel timing on `navigation` (BASELINE config 4): all 8 agents x 12 rays x 7 targets."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = make_env("navigation", num_envs=B, device="cuda:0", n_agents=8, seed=0, validate_actions=False)
w = env.world
be = w._get_backend()
if os.environ.get("LIDAR_COMPACT"):  # A/B: 0 the plain kernel, 1 the lane-compacted cast
    be.set_lidar_compact(int(os.environ["LIDAR_COMPACT"]))
for _ in range(50 if os.environ.get("MIDGAME", "1") == "1" else 0):  # a typical mid-episode state, not the spawn
    env.step([env.get_random_action(a) for a in env.agents])
import time
t_warm = time.perf_counter()
while time.perf_counter() - t_warm < 0.25:  # (the clocks of a just-started process are ramping)
    be.cast_rays()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    be.cast_rays()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
rays = B * 8 * 12
print(json.dumps({"B": B, "compact": be.lidar_compact, "lidar_us": us, "rays_per_s": rays / (us * 1e-6), "ray_tests_per_s": rays * 7 / (us * 1e-6),
                  "bytes_per_env": 8 * 12 + 8 * 12 * 4, "GBps": B * (96 + 384) / (us * 1e-6) / 1e9}))
# physics step of the same world, and a full Environment.step for the Amdahl picture
forces = torch.zeros(1, *be.agent_ft.shape, device="cuda:0")
e0.record()
be.step_n(n)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"world_step_us": e0.elapsed_time(e1) / n * 1e3, "lanes": be.lanes_per_env}))
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(3): env.step(acts)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): env.step(acts)
torch.cuda.synchronize()
print(json.dumps({"env_step_ms": (time.perf_counter() - t0) / 20 * 1e3, "env_steps_per_s": B * 20 / (time.perf_counter() - t0)}))
