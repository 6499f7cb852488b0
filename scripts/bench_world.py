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
exact = os.environ.get("EXACT", "1") != "0"  # the reference's batch-global broad phase (the default of every host path since round 6)
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, exact_broad_phase=exact, **kw)
for _ in range(30):  # a few real steps so that the state is a typical mid-episode one
    env.step([env.get_random_action(a) for a in env.agents])
forces = None
if os.environ.get("FORCES", "fixed") == "random":
    # SURVEY.md 8(d)'s protocol: per step and policy agent u ~ U(-u_range, u_range), pre-generated (here: the agent-force
    # rows a real rollout of `steps` random-action steps produced, scripted agents included); FORCES=fixed (default, what
    # rounds 1-2 measured with) holds the last action for the whole run - bodies pile up against the walls: a contact-dense
    # stress state
    snap = env.get_state()
    rows = []
    for _ in range(steps):
        env.step([env.get_random_action(a) for a in env.agents])
        rows.append(env.world._agent_ft.clone())
    forces = torch.stack(rows).contiguous()
    env.set_state(snap)
be = env.world._get_backend()
if lanes:
    be.set_lanes_per_env(lanes)
if os.environ.get("SPEC"):
    be.set_specialized(os.environ["SPEC"] != "0")
if os.environ.get("COMPACT"):
    be.set_compact(int(os.environ["COMPACT"]))
if os.environ.get("QUEUES"):
    be.set_queues(int(os.environ["QUEUES"]))
t_warm = time.perf_counter()  # (a quarter of a second of launches first: the clocks of a just-started process are ramping)
while time.perf_counter() - t_warm < 0.25:
    be.step_n(min(50, steps), None if forces is None else forces[: min(50, steps)], exact=exact)
    torch.cuda.synchronize()
    if os.environ.get("FORCES", "fixed") != "random":
        break  # (a held action changes the state it is timed on: keep the protocol of the earlier rounds, one warm-up call)
if forces is not None:
    env.set_state(snap)
torch.cuda.synchronize()
t0 = time.perf_counter()
be.step_n(steps, forces, exact=exact)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
if os.environ.get("LAZY_STATS"):
    print("lazy stats", be.lazy_stats(), file=sys.stderr)
if os.environ.get("COMPACT_STATS"):
    print("compact stats", be.compact_stats(), file=sys.stderr)
print(json.dumps({"scenario": name, "num_envs": B, "lanes": be.lanes_per_env, "queues": be.queues(steps), "specialized": be.specialized, "compact": be.compact,
                  "lib": os.environ.get("VMAS_HIP_LIB", "libvmas_hip.so"), "ablate": os.environ.get("VMAS_ABLATE", "0"), "forces": os.environ.get("FORCES", "fixed"), "exact": exact, "exact_form": be.exact_form() if hasattr(be.lib, "vmas_world_exact_form") else None,
                  "world_step_us": round(dt * 1e6, 2), "host_enqueue_us": round(t_enq / steps * 1e6, 2), "env_steps_per_s": round(B / dt)}))
