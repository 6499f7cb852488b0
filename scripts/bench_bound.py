"""Environment.bind + step_bound rate (GPU-bound: the one-launch step kernel's own time): python scripts/bench_bound.py balance 32768
With VMAS_HIP_LIB=libvmas_hip_profile.so: VMAS_ENV_ABLATE (1 queries off, 2 observations off, 4 epilogue off, 8 prologue
off; interpreter only - set SPEC=0) attributes the kernel's time to its stages."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
if os.environ.get("NAV_TILES"):  # A/B: navigation's one-launch step up to this many tiles per CU
    from vectorizedmultiagentsimulator_amd import fused as _F
    _F.NavigationPost.ONE_LAUNCH_MAX_TILES_PER_CU = int(os.environ["NAV_TILES"])
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
if os.environ.get("EXACT") is not None:  # A/B: the reference's broad-phase rule (default, the lazy form) against the per-environment form
    kw = dict(kw, exact_broad_phase=bool(int(os.environ["EXACT"])))
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
if os.environ.get("SPEC") == "0":
    env.world._get_backend().set_specialized(False)
if os.environ.get("LANES"):
    env.world._get_backend().set_lanes_per_env(int(os.environ["LANES"]))
acts = [env.get_random_action(a) for a in env.agents]
if os.environ.get("ACTIONS") == "zero":  # a typical mid-episode configuration held still (one fixed random action drives
    for _ in range(100):                 # the agents apart for good: no sensor has anything in reach, no contact is live)
        env.step([env.get_random_action(a) for a in env.agents])
    acts = [torch.zeros_like(a) for a in acts]
env.bind(acts)
for _ in range(300):
    env.step_bound()
torch.cuda.synchronize()
import time
n = 2000
windows = []
for _ in range(int(os.environ.get("WINDOWS", "5"))):  # (each window: n steps between synchronisations, HIP events inside)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        env.step_bound()
    t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    windows.append((e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
windows.sort()
med = windows[len(windows) // 2]
if os.environ.get("DIAG"):
    st = env.world._state
    print("DIAG state nan", int(torch.isnan(st).sum()), "absmax", float(st[:, :2].abs().max()), "pos std", float(st[:8, :2, :B].std()),
          "vel absmax", float(st[:8, 2:4, :B].abs().max()), file=sys.stderr)
    o = env.step_bound()[0]
    print("DIAG obs lidar mean", float(torch.stack(o)[..., 6:].mean()), "nonzero frac", float((torch.stack(o)[..., 6:] > 0).float().mean()), file=sys.stderr)
print(json.dumps({"scenario": name, "num_envs": B, "specialized": env.world._get_backend().specialized,
                  "ablate": os.environ.get("VMAS_ENV_ABLATE"), "actions": os.environ.get("ACTIONS", "fixed"), "exact": os.environ.get("EXACT"), "lanes": env.world._get_backend().lanes_per_env,
                  "step_bound_us": round(med[0], 2), "enqueue_us": round(med[1], 2), "wall_us": round(med[2], 2),
                  "windows_us": [round(w[0], 2) for w in windows]}))
