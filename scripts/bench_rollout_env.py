"""Environment.step() one launch per step vs Environment.rollout() (K steps in one launch, vmas_world_rollout_env):
python scripts/bench_rollout_env.py balance 32768 [K]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8),
      "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
if os.environ.get("FOOTBALL_FORM"):  # A/B: include/vmas_debug_hip.h, vmas_debug_football_form
    import ctypes
    _be = env.world._get_backend()
    _be.lib.vmas_debug_football_form.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    assert _be.lib.vmas_debug_football_form(_be._h, int(os.environ["FOOTBALL_FORM"])) == 0
if os.environ.get("SPEC") == "0":  # A/B: the interpreter instead of the world-specialised kernel
    env.world._get_backend().set_specialized(False)
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = [(torch.rand(K, B, 2, device="cuda:0", generator=g) * 2 - 1) for _ in env.agents]
# the rollout's outputs in buffers of the caller (rollout(out=)): config 5's K = 50 steps are 23 GB of observations, and a
# fresh allocation of that size inside the timed region is a 100 ms hipMalloc whenever the caching allocator has no free
# block of it - 2 100 us per step instead of 185 in two of the round's runs (profiles/r04i_*), nothing the kernel did
buf = {n_: torch.empty(shape, dtype=dt, device="cuda:0") for n_, shape, dt in env.rollout_fields(K)}
t_warm = time.perf_counter()  # (at least a quarter of a second of the measured work first: a process that has just been
while time.perf_counter() - t_warm < 0.25:  #  set up finds the GPU's clocks ramping - the first ~0.1 s reads up to 5x slow)
    env.rollout(acts, out=buf)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get('REPS', 20))
t0 = time.perf_counter(); e0.record()
for _ in range(reps):
    out = env.rollout(acts, out=buf)
e1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / (reps * K)
gpu = e0.elapsed_time(e1) * 1e-3 / (reps * K)
for k in range(50):
    env.step([u[k % K] for u in acts])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(reps * K // 4):
    env.step([u[k % K] for u in acts])
torch.cuda.synchronize()
step = (time.perf_counter() - t0) / (reps * K // 4)
print(json.dumps({"scenario": name, "num_envs": B, "K": K, "football_form": os.environ.get("FOOTBALL_FORM"), "specialized": env.world._get_backend().specialized, "rollout_us_per_step_gpu": round(gpu * 1e6, 2),
                  "rollout_us_per_step_wall": round(wall * 1e6, 2), "rollout_env_steps_per_s": round(B / wall),
                  "step_us_per_step_wall": round(step * 1e6, 2), "step_env_steps_per_s": round(B / step)}))
