export ACTIONS=zero
b() { python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c130-260; }
python scripts/window_rate_bound.py 16384 5
b
python scripts/window_rate_bound.py 16384 5
b
python - <<'P'
import sys, time, torch
sys.path.insert(0, ".")
from vectorizedmultiagentsimulator_amd.environment import make_env
env = make_env("navigation", num_envs=16384, device="cuda:0", seed=0, n_agents=8, validate_actions=False)
for _ in range(100): env.step([env.get_random_action(a) for a in env.agents])
acts = [torch.zeros_like(env.get_random_action(a)) for a in env.agents]
env.bind(acts)
for _ in range(300): env.step_bound()
torch.cuda.synchronize()
for w in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(2000): env.step_bound()
    e1.record(); torch.cuda.synchronize()
    print("with events: wall", round((time.perf_counter() - t0) / 2000 * 1e6, 1), "events", round(e0.elapsed_time(e1) / 2000 * 1e3, 1))
for w in range(3):
    t0 = time.perf_counter()
    for _ in range(2000): env.step_bound()
    torch.cuda.synchronize()
    print("no events: wall", round((time.perf_counter() - t0) / 2000 * 1e6, 1))
P
