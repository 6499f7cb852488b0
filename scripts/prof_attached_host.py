"""Host-side profile of the REFERENCE's env.step after attach() (cProfile, top functions by own time):
python scripts/prof_attached_host.py balance 32768 [deferred|off|on]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch
from oracle import ref
from vectorizedmultiagentsimulator_amd.adapter import attach
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
mode = {"deferred": "deferred", "off": False, "on": True}[sys.argv[3] if len(sys.argv) > 3 else "off"]
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
env = ref.make_env(name, num_envs=B, device="cuda:0", seed=0, **kw)
h = attach(env, validate_actions=mode)
assert h.fused is not None, h.fused_reason
cycle = [[env.get_random_action(a) for a in env.agents] for _ in range(25)]
for k in range(300):
    env.step(cycle[k % 25])
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for k in range(n):
    env.step(cycle[k % 25])
torch.cuda.synchronize()
print("env.step wall us:", (time.perf_counter() - t0) / n * 1e6)
env.reset()
torch.cuda.synchronize()
n2 = 200  # (short: the queue must not fill, or the host time is the GPU's)
t0 = time.perf_counter()
for k in range(n2):
    env.step(cycle[k % 25])
print("env.step host-only us (no sync, 200 steps):", (time.perf_counter() - t0) / n2 * 1e6)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(n):
    env.step(cycle[k % 25])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:18]
for (f, line, fn), (cc, nc, tt, ct, _) in rows:
    print("%7.2f us own %7.2f us cum  x%-5.1f %s:%d %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, os.path.basename(f), line, fn))
