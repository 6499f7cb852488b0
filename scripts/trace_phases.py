"""Per-wave phase timeline of one step_kernel launch (VMAS_TRACE=1): s_memtime stamps
0 start | 1 loads issued+landed | 2 after load barrier | 3 end of gather | 4 after barrier | 5 end"""
import os, sys, ctypes
os.environ["VMAS_TRACE"] = "1"
os.environ.setdefault("VMAS_HIP_LIB", "libvmas_hip_trace.so")  # -DVMAS_PROFILE -DVMAS_TRACE build of the same sources
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import bench
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
sc, w = bench.build_world(B, torch.device("cuda", 0), 4, lanes, 0)
be = w._get_backend()
be.set_queues(1)
forces = bench.pack_forces(w, bench.make_actions(100, 4, B, 1234), torch.device("cuda", 0))
be.step_n(60, forces[:60]); torch.cuda.synchronize()
be.step_n(1, forces[60:61]); torch.cuda.synchronize()
tiles = (B + 63) // 64
buf = np.zeros(tiles * 16 * 16, np.uint64)
lib = be.lib
lib.vmas_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
assert lib.vmas_debug_trace(be._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
full = buf.reshape(tiles, 16, 16).astype(np.int64)[:, :lanes]
t = full[:, :, :6]
t0 = t[:, :, 0].min()
print("lanes", lanes, "tiles", tiles, "kernel span (cycles, s_memtime @100MHz?)", t[:, :, 5].max() - t0)
span = (t[:, :, 5].max(axis=1) - t[:, :, 0].min(axis=1))
print("per-tile span: mean %.0f min %d max %d" % (span.mean(), span.min(), span.max()))
gat = t[:, :, 3] - t[:, :, 2]
print("gather per wave: mean %.0f; slowest wave of a tile: mean %.0f (imbalance %.2fx)" % (gat.mean(), gat.max(axis=1).mean(), gat.max(axis=1).mean() / gat.mean()))
integ = t[:, :, 5] - t[:, :, 4]
print("integrate per wave: mean %.0f; slowest wave of a tile: mean %.0f" % (integ.mean(), integ.max(axis=1).mean()))
for b in (tiles // 2, 3):
    print("tile", b)
    for wv in range(lanes):
        r = t[b, wv] - t0
        print("  wave %2d start %6d | load %6d | bar %6d | gather %6d | bar %6d | integrate %6d" % (wv, r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
d = t - t[:, :, 0:1]
print("mean per-phase over all waves:", (t[:, :, 1:] - t[:, :, :-1]).mean(axis=(0, 1)))
print("max  per-phase over all waves:", (t[:, :, 1:] - t[:, :, :-1]).max(axis=(0, 1)))
print("start spread:", t[:, :, 0].max() - t0, " end min/max:", t[:, :, 5].min() - t0, t[:, :, 5].max() - t0)

g = full[:, :, 6:16].astype(np.float64)
print("gather breakdown, mean cycles per wave: grab+segdesc %.0f | prologue %.0f | SS %.0f (n=%.1f) | LS %.0f (n=%.1f) | BS %.0f (n=%.1f) | other %.0f (n=%.1f)" % (
    g[:, :, 0].mean(), g[:, :, 1].mean(), g[:, :, 2].mean(), g[:, :, 6].mean(), g[:, :, 3].mean(), g[:, :, 7].mean(),
    g[:, :, 4].mean(), g[:, :, 8].mean(), g[:, :, 5].mean(), g[:, :, 9].mean()))
for c, nm in enumerate(("SS", "LS", "BS", "other")):
    n = g[:, :, 6 + c].sum()
    if n: print("  per-item %s: %.0f cycles (count per tile %.1f)" % (nm, g[:, :, 2 + c].sum() / n, n / tiles))
