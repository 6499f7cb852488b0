"""Attribute the one-launch balance step kernel's time to its stages (specialised kernel): the full launch, the launch
without the ingest prologue (forces read from agent_ft), plain physics.  python scripts/bench_env_stages.py [num_envs]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from vectorizedmultiagentsimulator_amd import _abi as A
from vectorizedmultiagentsimulator_amd.environment import make_env
from vectorizedmultiagentsimulator_amd.fused import _stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = make_env("balance", num_envs=B, device="cuda:0", seed=0, validate_actions=False, n_agents=4)
acts = [env.get_random_action(a) for a in env.agents]
env.bind(acts)
for _ in range(100):
    env.step_bound()
L = env._launch
desc, buf, _ = env._bound
lib = A.load_library()
be = env.world._get_backend()
def timed(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 2)
s = _stream(env.device)
full = timed(env.step_bound)
no_ing = timed(lambda: lib.vmas_world_step_env(L._h, L._st, L._ft, L._ld, None, None, None, 1, C.byref(desc), C.byref(buf), s))
ing_only = timed(lambda: lib.vmas_world_step_env(L._h, L._st, L._ft, L._ld, None, L._ing, None, 0, None, None, s))
plain = timed(lambda: be.step())
print(json.dumps({"num_envs": B, "specialized": be.specialized, "full_us": full, "epilogue_only_us": no_ing,
                  "prologue_only_us": ing_only, "plain_physics_us": plain}))
