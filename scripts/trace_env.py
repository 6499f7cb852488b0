"""Per-wave timeline of a one-launch Environment.step on the world-specialised kernel (balance, transport): s_memtime stamps
of the -DVMAS_PROFILE -DVMAS_TRACE build with VMAS_TRACE=2.  python scripts/trace_env.py balance 32768"""
import ctypes, os, sys
os.environ["VMAS_TRACE"] = "2"
os.environ.setdefault("VMAS_HIP_LIB", "libvmas_hip_trace.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "balance"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
kw = {"balance": dict(n_agents=4), "transport": {}}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
be = env.world._get_backend()
for _ in range(50):
    env.step([env.get_random_action(a) for a in env.agents])
torch.cuda.synchronize()
tiles = (B + 63) // 64
buf = np.zeros(tiles * 16 * 16, np.uint64)
lib = be.lib
lib.vmas_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
assert lib.vmas_debug_trace(be._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
lanes = be.lanes_per_env
t = buf.reshape(tiles, 16, 16).astype(np.int64)[:, :lanes]
print(name, B, "specialized", be.specialized, "waves per tile", lanes, "tiles", tiles, "(s_memtime ticks)")
names = ["start", "loads + action ingest", "load barrier", "gather", "barrier", "integrate + stores", "barrier", "epilogue"]
last = 7 if t[:, :, 7].any() else 5  # (only balance's epilogue is stamped)
for k in range(1, last + 1):
    d = t[:, :, k] - t[:, :, k - 1]
    print("  %-22s mean %7.0f  max-wave-of-tile mean %7.0f  max %7d" % (names[k], d.mean(), d.max(axis=1).mean(), d.max()))
print("per-tile span mean", (t[:, :, last].max(axis=1) - t[:, :, 0].min(axis=1)).mean())
