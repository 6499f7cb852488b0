"""Markdown tables of one default bench.py line (DESIGN.md section 6 / README): python scripts/round_table.py LINE.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])


def us(x, n=1):
    return "-" if x is None else f"{x:.{n}f}"


def G(x):
    return f"{x / 1e9:.2f} G"


rows = []
a = d.get("attached_reference", {})
cb = d.get("cpu_baseline", {})
ws = d.get("world_step", {})
es = d.get("environment_step", {})
rows.append(("balance n_agents=4 32 768", d["ms_per_step"] * 1e3, d["value"], a, d["roofline"], ws.get("ms_per_step", 0) * 1e3, ws.get("roofline", {}),
             es, cb.get("gpu_over_cpu"), (cb.get("world_step") or {}).get("gpu_over_cpu"), d.get("parity") or {}))
names = {"transport": "transport 16 384", "transport_2pkg": "transport n_packages=2 16 384", "navigation": "navigation n_agents=8 65 536",
         "football": "football 5v5 131 072"}
for k, o in (d.get("other_configs") or {}).items():
    if "error" in o:
        continue
    w_ = o.get("world_step", {})
    rows.append((names[k], o["us_per_step"], o["value"], o.get("attached_reference", {}), o.get("roofline", {}), w_.get("us_per_step"),
                 w_.get("roofline", {}), o.get("environment_step", {}), o.get("gpu_over_cpu"), w_.get("gpu_over_cpu"), o.get("parity") or {}))

print("| configuration | `env.step` of the attached reference, asserts kept (= `value`) | asserts deferred / off | K-step rollout, per step | "
      "one-launch kernel: frac of 8 TB/s (traffic / algorithmic) | ÷ same-run CPU reference `env.step` |")
print("|---|---|---|---|---|---|")
for name, step_us, value, a, rf, _, _, _, over, _, _ in rows:
    print(f"| {name} | **{us(step_us)} µs** = {G(value)} env-steps/s | {us(a.get('env_step_deferred_validate_us'))} / {us(a.get('env_step_no_validate_us'))} µs | "
          f"{us(a.get('rollout_us_per_step'))} µs | {us(rf.get('frac'), 3)} ({us(rf.get('traffic_over_algorithmic'), 2)} x) | {over:,.0f} x |".replace(",", "\u202f"))
print()
print("| configuration | `World.step` (physics only, HIP events) | frac of 8 TB/s (traffic) | native `Environment.step` from Python (wall / GPU) | `step_bound` / rollout per step | ÷ CPU reference `World.step` |")
print("|---|---|---|---|---|---|")
for name, _, _, _, _, w_us, w_rf, es, _, w_over, _ in rows:
    b = es.get("bound", {}).get("gpu_us_per_step", es.get("bound_us_per_step"))
    r = es.get("rollout", {}).get("us_per_step", es.get("rollout_us_per_step"))
    print(f"| {name} | {us(w_us, 2)} µs | {us(w_rf.get('frac'), 3)} ({us(w_rf.get('traffic_over_algorithmic'), 2)} x) | {us(es.get('us_per_step'))} / {us(es.get('gpu_us_per_step'))} µs | "
          f"{us(b)} / {us(r)} µs | {(w_over or 0):,.0f} x |".replace(",", "\u202f"))
print()
print("| configuration | teacher-forced: values beyond 1e-5 / compared (reference steps) | max abs error | per-environment control: values beyond / steps with one | free-running drift (steps: max, fraction beyond 1e-3) |")
print("|---|---|---|---|---|")
for name, *_, p in rows:
    if not p or "error" in p:
        continue
    c = p.get("per_environment_form_control", {})
    f = p.get("free_running_drift", {})
    print(f"| {name} ({p.get('envs')} envs) | **{p.get('values_beyond_1e-5')} / {p.get('values_compared'):,}** ({p.get('teacher_forced_steps')}) | {p.get('teacher_forced_max_abs'):.1e} | "
          f"{c.get('values_beyond_1e-5')} / {c.get('steps_with_a_value_beyond')} | {f.get('steps')}: {f.get('max_abs'):.2g}, {f.get('frac_beyond_1e-3'):.1e} |".replace(",", "\u202f"))
