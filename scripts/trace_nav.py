"""Per-wave timeline of the one-launch navigation Environment.step (physics + LIDAR / observation / reward epilogue):
s_memtime stamps of the -DVMAS_PROFILE -DVMAS_TRACE build.  python scripts/trace_nav.py [num_envs]"""
import ctypes, os, sys
os.environ.setdefault("VMAS_TRACE", "1")  # 2: the world-specialised kernel's stamps
os.environ.setdefault("VMAS_HIP_LIB", "libvmas_hip_trace.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from vectorizedmultiagentsimulator_amd.environment import make_env
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
if os.environ.get("NAV_TILES"):
    from vectorizedmultiagentsimulator_amd import fused as _F
    _F.NavigationPost.ONE_LAUNCH_MAX_TILES_PER_CU = int(os.environ["NAV_TILES"])
env = make_env("navigation", num_envs=B, device="cuda:0", seed=0, validate_actions=False, n_agents=8)
be = env.world._get_backend()
lanes = be.lanes_per_env
acts = [env.get_random_action(a) for a in env.agents]
for _ in range(50):
    env.step([env.get_random_action(a) for a in env.agents])
torch.cuda.synchronize()
tiles = (B + 63) // 64
buf = np.zeros(tiles * 16 * 16, np.uint64)
lib = be.lib
lib.vmas_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
assert lib.vmas_debug_trace(be._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(tiles, 16, 16).astype(np.int64)[:, :lanes]
t0 = t[:, :, 0].min()
print("specialized", be.specialized, "VMAS_TRACE", os.environ["VMAS_TRACE"])
names = ["start", "loads", "load barrier", "gather end", "barrier", "integrate end", "epilogue in", "pair bits", "lidar units",
         "lidar barrier", "obs+reward"]
print("lanes", lanes, "tiles", tiles, "(s_memtime ticks)")
for k in range(1, 11):
    d = t[:, :, k] - t[:, :, k - 1]
    print("  %-14s mean %7.0f  max-wave-of-tile mean %7.0f  max %7d" % (names[k], d.mean(), d.max(axis=1).mean(), d.max()))
if t[:, :, 11].any():
    sub = ["start", "near pairs queued", "barrier", "items done (-> lidar units stamp)"]
    seq = [7, 11, 12, 14, 8]
    for i in range(1, len(seq)):
        d = t[:, :, seq[i]] - t[:, :, seq[i - 1]]
        print("    lidar: %-48s mean %7.0f  max-wave-of-tile mean %7.0f" % (sub[i - 1], d.mean(), d.max(axis=1).mean()))
print("kernel span", t[:, :, 10].max() - t0, " per-tile span mean", (t[:, :, 10].max(axis=1) - t[:, :, 0].min(axis=1)).mean())
start = np.sort(t[:, 0, 0] - t0)
print("tile start quantiles (ticks after the first tile's): 25%% %d 50%% %d 75%% %d 100%% %d" % tuple(start[[len(start) // 4, len(start) // 2, 3 * len(start) // 4, -1]]))
