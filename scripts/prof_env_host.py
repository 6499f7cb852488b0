"""Host-side profile of Environment.step (cProfile, top functions by own time): python scripts/prof_env_host.py balance 32768"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}, "navigation": dict(n_agents=8)}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(300):
    env.step(acts)
torch.cuda.synchronize()
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    env.step(acts)
torch.cuda.synchronize()
print("env.step wall us:", (time.perf_counter() - t0) / n * 1e6)
t0 = time.perf_counter()
for _ in range(n):
    env.step(acts)
print("env.step host-only us (no sync):", (time.perf_counter() - t0) / n * 1e6)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    env.step(acts)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:22]
for (f, line, fn), (cc, nc, tt, ct, _) in rows:
    print("%7.2f us own %7.2f us cum  x%-5.1f %s:%d %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, os.path.basename(f), line, fn))
