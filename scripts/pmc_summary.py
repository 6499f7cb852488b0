"""Summarise the rocprofv3 passes of scripts/gpu_counters.sh: per kernel of OUR library (step / lidar / post / mask
kernels), mean counters per dispatch and per wave, achieved GB/s and GFLOP/s against the MI355X peaks, and which
resource binds.  usage: pmc_summary.py WORKDIR KERNEL_STATS_CSV BYTES_PER_ENV FLOP_PER_ENV ENVS "cmd" [RATED_KERNEL]

Only the kernel FAMILY named by RATED_KERNEL - the kernel's name up to its template arguments, compared exactly
("step_kernel", "step_kernel_spec", "step_kernel_spec_multi", "step_kernel_compact", "lidar_compact_kernel", "vmas_rt_lean" ...; optionally ":physics" / ":env" behind it - a launch with or
without Environment.step stages, told by its DevEnv / NoEnv argument; default "step_kernel_spec") - gets an achieved-GB/s / GFLOP/s / traffic-over-algorithmic line: the per-environment bytes and flop
on the command line are that family's in that run.  Every other kernel of the trace is listed with its counters and NO
rating (round 2 rated a physics kernel with LIDAR flops, round 3's substring match rated env-step kernels with physics
bytes).  A rated fraction above 1 is an error of the inputs and aborts.

Peaks (MI355X_MICROARCH.md): HBM 8.0 TB/s; fp32 vector 157.3 TFLOP/s (256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz).
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B; calibrated in this library's own 4 B/lane row pattern,
profiles/r01_traffic_calibration.txt), WRITE_SIZE taken as is.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles.
"""
import collections
import csv
import glob
import sys

work, stats_csv, bpe, fpe, envs, cmd = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
rated = sys.argv[7] if len(sys.argv) > 7 else "step_kernel_spec"
OURS = ("step_kernel", "lidar_kernel", "lidar_compact_kernel", "collision_kernel", "post_kernel", "pair_mask_kernel", "ingest_kernel", "query_kernel", "reset_kernel",
        "rollout", "vmas_rt_")


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def family(name):
    f = short(name).split("<")[0].strip().replace("compact::", "")
    for suffix in ("_t0", "_t1"):  # run-time specialisations (specialize.py): vmas_rt_lean_t0, vmas_rt_multi_e1_o1_t0
        if f.startswith("vmas_rt_") and f.endswith(suffix):
            f = f[: -len(suffix)]
    return f


def variant(name):
    """'env': the launch carries Environment.step stages (action ingest / a post-step epilogue: a DevEnv argument);
    'physics': plain World.step."""
    n = short(name)
    if n.startswith("vmas_rt_multi_e"):
        return "physics" if n.startswith("vmas_rt_multi_e0") else "env"
    return "env" if "DevEnv" in name else "physics"


def is_rated(name):
    fam, _, var = rated.partition(":")
    return family(name) == fam and (not var or variant(name) == var)


acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
full = {}  # short name -> full name (with the argument list: the Env type tells physics from env launches)
for d in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"{work}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not any(o in k for o in OURS):
                continue
            full[k] = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
dur = {}
try:
    for r in csv.DictReader(open(stats_csv)):
        dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]), float(r["MinNs"]), float(r["MaxNs"]), float(r["StdDev"]))
        full.setdefault(short(r["Name"]), r["Name"])
except Exception as e:  # noqa: BLE001
    print("no kernel stats:", e)
# per-dispatch view of the kernel trace: medians, and the same over the LAST HALF of a kernel's calls - the profiler's mean
# over a short run is taken before the clocks of a fresh process are up (profiles/README.md) and rates the kernel too low
per_dispatch = {}
for f in glob.glob(f"{work}/trace/**/*kernel_trace.csv", recursive=True):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(o in k for o in OURS):
            full.setdefault(k, r["Kernel_Name"])
            rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for k, v in rows.items():
        v.sort()
        d = [e - b for b, e in v]
        per = [v[i + 1][0] - v[i][0] for i in range(len(v) - 1)]
        med = lambda x: sorted(x)[len(x) // 2] if x else float("nan")
        per_dispatch[k] = (med(d), med(d[len(d) // 2:]), med(per), med(per[len(per) // 2:]), (v[-1][1] - v[0][0]) / 1e6)
print(f"# {cmd}")
print(f"# envs {envs}, algorithmic bytes/env-step {bpe:g}, flop/env-step {fpe:g} (SURVEY.md section 8d)")
for k in sorted(set(acc) | set(per_dispatch), key=lambda k: -dur.get(k, (0,))[0]):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()} if k in acc else {}
    if k in meta:
        g = meta[k]
        print(f"\n== {k}\n   grid {g[0]} threads, block {g[1]}, LDS {g[2]} B, VGPR {g[3]}, SGPR {g[4]}")
    else:
        print(f"\n== {k}\n   (kernel trace only: no counter pass saw it)")
    if k in dur:
        avg, calls, mn, mx, sd = dur[k]
        print(f"   duration (kernel trace): avg {avg / 1e3:.2f} us over {calls} calls (min {mn / 1e3:.2f}, max {mx / 1e3:.2f}, sd {sd / 1e3:.2f})")
    if k in per_dispatch:
        m, m2, pm, pm2, span = per_dispatch[k]
        print(f"   per dispatch: duration median {m / 1e3:.2f} us (last half of the calls {m2 / 1e3:.2f}), launch-to-launch median "
              f"{pm / 1e3:.2f} us (last half {pm2 / 1e3:.2f}); first to last call {span:.1f} ms")
    waves = c.get("SQ_WAVES", 0)
    for n in sorted(c):
        per_wave = f"  per wave {c[n] / waves:.1f}" if waves and n.startswith("SQ_") and n != "SQ_WAVES" else ""
        print(f"   {n:22s} {c[n]:14.6g}{per_wave}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        print(f"   HBM traffic per launch  {traffic / 1e6:.2f} MB (FETCH x2 + WRITE)")
    if k in dur and is_rated(full.get(k, k)):
        t = dur[k][0] * 1e-9
        gbs, gfs = bpe * envs / t / 1e9, fpe * envs / t / 1e9
        print(f"   achieved (algorithmic)  {gbs:.0f} GB/s = {gbs / 8000:.3f} of HBM peak | {gfs:.0f} GFLOP/s = {gfs / 157300:.3f} of fp32 vector peak")
        assert gbs / 8000 <= 1.0 and gfs / 157300 <= 1.0, f"{k}: a fraction above the roof - wrong bytes / flop per environment for this kernel"
        if k in per_dispatch:  # the same by the median duration of the last half of the calls (the clocks are up by then)
            t2 = per_dispatch[k][1] * 1e-9
            print(f"   by the last half's median duration ({t2 * 1e6:.2f} us): {bpe * envs / t2 / 1e9:.0f} GB/s = {bpe * envs / t2 / 8e12:.3f} of HBM peak  <- the sustained figure")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            print(f"   traffic / algorithmic   {traffic / (bpe * envs):.3f}")
    elif k in dur:
        print(f"   (not rated: the per-environment bytes / flop of this run belong to the family {rated!r})")
    if waves and "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        parts = []
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if n in c:
                parts.append(f"{n[3:]} {c[n] / wc:.2f}")
        print("   share of wave-cycles    " + ", ".join(parts))
        if "SQ_BUSY_CYCLES" in c and "SQ_INSTS_VALU" in c:
            # VALU issue pressure: a wave64 fp32 VALU instruction occupies a SIMD for 2 cycles (1 issue quad-cycle ~ 4 cyc);
            # SQ_BUSY_CYCLES is summed over the XCDs' SQs (quad-cycles)
            print(f"   VALU insts / SALU insts {c['SQ_INSTS_VALU'] / max(c.get('SQ_INSTS_SALU', 1), 1):.2f}")
