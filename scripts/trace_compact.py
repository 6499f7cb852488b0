"""Per-wave phase sums of one launch of the lane-compacted kernel (VMAS_TRACE build: libvmas_hip_trace.so).
python scripts/trace_compact.py football 1024"""
import os, sys, ctypes
os.environ["VMAS_TRACE"] = "1"
os.environ.setdefault("VMAS_HIP_LIB", "libvmas_hip_trace.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from vectorizedmultiagentsimulator_amd.environment import make_env
name = sys.argv[1] if len(sys.argv) > 1 else "football"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kw = {"navigation": dict(n_agents=8), "football": dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)}[name]
env = make_env(name, num_envs=B, device="cuda:0", seed=0, validate_actions=False, **kw)
for _ in range(30):
    env.step([env.get_random_action(a) for a in env.agents])
be = env.world._get_backend()
hold = int(os.environ.get("HOLD", "0"))  # the last action held for this many steps first: bodies drift into the walls
if hold:
    be.set_compact(0)
    be.step_n(hold); torch.cuda.synchronize()
be.set_compact(1)
be.set_queues(1)
assert be.compact
exact = os.environ.get("EXACT", "1") != "0"  # the lazy exact broad phase (the default of every host path since round 6)
be.step_n(20, exact=exact); torch.cuda.synchronize()
be.step_n(1, exact=exact); torch.cuda.synchronize()
tiles = (B + 63) // 64
buf = np.zeros(tiles * 16 * 16, np.uint64)
lib = be.lib
lib.vmas_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
assert lib.vmas_debug_trace(be._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(tiles, 16, 16).astype(np.int64)
nw = int((t[:, :, 0] > 0).sum(axis=1).max())
t = t[:, :nw]
t0 = t[:, :, 0].min()
print(f"{name} {B} (held {hold} steps, exact broad phase {exact}): tiles {tiles}, waves/tile {nw}; kernel span {t[:, :, 3].max() - t0} ticks (s_memtime)")
start = t[:, :, 0] - t0
print("wave start after the grid's first wave (ticks): mean %.0f | median %.0f | max %.0f; last wave of a tile after its first: mean %.0f" % (
    start.mean(), np.median(start), start.max(), (t[:, :, 0].max(axis=1) - t[:, :, 0].min(axis=1)).mean()))
print("per wave means (ticks): load %.0f | load barrier %.0f | total %.0f" % (
    (t[:, :, 1] - t[:, :, 0]).mean(), (t[:, :, 2] - t[:, :, 1]).mean(), (t[:, :, 3] - t[:, :, 0]).mean()))
if (t[:, :, 12] > 0).all():
    print("  inside load: start -> every load of the first batch requested %.0f | -> entity rows in LDS (waits for the loads, sincos of the lines) %.0f | -> agent rows, blob, zeroing %.0f" % (
        (t[:, :, 12] - t[:, :, 0]).mean(), (t[:, :, 13] - t[:, :, 12]).mean(), (t[:, :, 1] - t[:, :, 13]).mean()))
names = ["prologue+integrate", "A broad", "A barrier", "B narrow", "B barrier", "C add contacts", "contacts N", "rounds"]
for k, nm in enumerate(names):
    v = t[:, :, 4 + k]
    print("  %-20s mean %9.1f  max %9d" % (nm, v.mean(), v.max()))
print("  lazy exact broad phase: overlap phase + publish mean %.1f max %d | look at the batch's words mean %.1f max %d" % (
    t[:, :, 14].mean(), t[:, :, 14].max(), t[:, :, 15].mean(), t[:, :, 15].max()))
print("by wave index (mean over tiles): prologue+integrate | A broad | A barrier | C add contacts")
for wv in range(nw):
    print("  wave %2d  %8.0f %8.0f %8.0f %8.0f" % (wv, t[:, wv, 4].mean(), t[:, wv, 5].mean(), t[:, wv, 6].mean(), t[:, wv, 9].mean()))
