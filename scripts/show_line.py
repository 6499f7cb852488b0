"""Print the fields of a bench.py JSON line that a GPU call's log should show (the file keeps everything)."""
import json
import sys

path, brief = sys.argv[1], len(sys.argv) > 2
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    sys.exit(f"{path}: no JSON line ({e})")


def r(x, n=2):
    return round(x, n) if isinstance(x, (int, float)) else x


es = d.get("environment_step", {})
print("line:", {"value": r(d["value"], 0), "ms_per_step": r(d["ms_per_step"], 5), "frac": r(d["roofline"]["frac"], 4),
                "traffic_x": r(d["roofline"].get("traffic_over_algorithmic"), 3),
                "env_step_us": r(es.get("us_per_step")), "env_gpu_us": r(es.get("gpu_us_per_step")),
                "bound_us": r(es.get("bound", {}).get("gpu_us_per_step")), "rollout_us": r(es.get("rollout", {}).get("us_per_step"))})
ws = d.get("world_step")
if ws:
    print("world_step:", {"value": r(ws["value"], 0), "ms_per_step": r(ws["ms_per_step"], 5), "frac": r(ws["roofline"]["frac"], 4),
                          "traffic_x": r(ws["roofline"].get("traffic_over_algorithmic"), 3)})
    print("headline:", d.get("headline"), {k: r(v, 5) for k, v in d["repeats"].items() if k != "note"})
if brief:
    sys.exit(0)
a = d.get("attached_reference", {})
print("attached:", {k: r(v) for k, v in a.items() if k not in ("value_is",)})
print("parity:", {k: v for k, v in (d.get("parity") or {}).items() if k != "note"})
c = d.get("cpu_baseline", {})
print("cpu:", {"value": r(c.get("value"), 0), "cores": c.get("cores"), "gpu_over_cpu": r(c.get("gpu_over_cpu"), 0),
               "env_step": {k: r(v) for k, v in c.get("env_step", {}).items() if k != "note"}})
for name, o in (d.get("other_configs") or {}).items():
    if "error" in o:
        print(name, "ERROR", o["error"][:300])
        continue
    e = o.get("environment_step", {})
    at = o.get("attached_reference", {})
    print(name, {"us": r(o["us_per_step"]), "frac": r(o["roofline"]["frac"], 3), "traffic_x": r(o["roofline"].get("traffic_over_algorithmic"), 3), "env_us": r(e.get("us_per_step")), "env_gpu_us": r(e.get("gpu_us_per_step")),
                 "env_frac": r(e.get("roofline_frac"), 3), "bound_us": r(e.get("bound_us_per_step")), "rollout_us": r(e.get("rollout_us_per_step")),
                 "gpu_over_cpu": r(o.get("gpu_over_cpu"), 0), "env_over_cpu": r(e.get("gpu_over_cpu"), 0)})
    w_ = o.get("world_step")
    if w_:
        print("   world_step:", {"value": r(w_["value"], 0), "us": r(w_["us_per_step"]), "frac": r(w_["roofline"]["frac"], 3),
                                 "traffic_x": r(w_["roofline"].get("traffic_over_algorithmic"), 3), "gpu_over_cpu": r(w_.get("gpu_over_cpu"), 0)})
    print("   attached:", {k: r(v) for k, v in at.items() if k not in ("value_is", "unit")})
    print("   parity:", {k: v for k, v in (o.get("parity") or {}).items() if k != "note"}, "cpu:", o.get("cpu_reference"))
