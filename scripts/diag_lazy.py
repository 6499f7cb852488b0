"""Diagnostic (GPU): the lazy exact broad phase against the launch-per-substep form on a jittered fixture batch, step by step.
python scripts/diag_lazy.py soup_solid 512 [seed]  -> where the two first differ, and what the oracle says there."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load
from test_hip_parity import make_batch, _hip, _up, _dev
from oracle.oracle import Oracle

name, B = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 41
g = load(name)
st0, ft0, jfr_np, eg_np = make_batch(g, B, seed=seed)
o = Oracle(g.spec)
spec = g.spec
print("pairs", len(spec.pairs), "substeps", spec.substeps, "entities", spec.n_entities, "types", sorted(set(p.type for p in spec.pairs)))
hw = {m: _hip(spec, B) for m in ("launches", "lazy")}
for m in hw:
    _up(hw[m], st0, ft0)
jfr = {m: _dev(hw[m], jfr_np, B) for m in hw}
eg = {m: _dev(hw[m], eg_np, B) for m in hw}
print("form", hw["lazy"].exact_form(), "lanes", hw["lazy"].lanes_per_env, "spec", hw["lazy"].specialized, "compact", hw["lazy"].compact)
for step in range(5):
    pre = hw["launches"].state.clone()
    pre_ft = hw["launches"].agent_ft.clone()
    masks = []
    hw["launches"].step_exact_launches(joint_fixed_rot=jfr["launches"], entity_gravity=eg["launches"])
    hw["lazy"].step_exact(joint_fixed_rot=jfr["lazy"], entity_gravity=eg["lazy"])
    a, b = hw["launches"].state[:, :, :B].cpu().numpy(), hw["lazy"].state[:, :, :B].cpu().numpy()
    neq = (a.view(np.uint32) != b.view(np.uint32))
    print(f"step {step}: {int(neq.sum())} words differ; status {hw['lazy'].exact_status()}")
    if neq.any():
        idx = np.argwhere(neq)
        envs = np.unique(idx[:, 2])
        print("  envs", envs[:20], "entities", np.unique(idx[:, 0]), "fields", np.unique(idx[:, 1]))
        with np.errstate(invalid="ignore"):
            d = np.abs(a - b)
        print("  max abs diff", np.nanmax(d), "nan mismatch", int((np.isnan(a) != np.isnan(b)).sum()))
        # the oracle from the same pre-state
        st = pre[:, :, :B].cpu().numpy().copy(); ft = pre_ft[: max(spec.n_agents, 1), :, :B].cpu().numpy().copy()[: spec.n_agents]
        want = st.copy()
        st_s = st.copy()
        for s in range(spec.substeps):
            m = o.pair_mask(want, B)
            off = [p for p in range(len(spec.pairs)) if not (m[p >> 5] >> (p & 31)) & 1]
            # band events in the differing envs
            for p in off:
                f = o.pair_forces(want, p, B)
                hit = np.nonzero(np.nan_to_num(np.abs(f).max(0), nan=1.0) > 0)[0]
                if len(hit):
                    print(f"  substep {s}: pair {p} type {spec.pairs[p].type} ({spec.pairs[p].a},{spec.pairs[p].b}) off for the batch, non-zero force in envs {hit[:10]}")
            o.step(want, ft, B, m, None if jfr_np is None else jfr_np, None if eg_np is None else eg_np, s, 1, 8)
        with np.errstate(invalid="ignore"):
            print("  |launches - oracle| max", np.nanmax(np.abs(a - want)), " |lazy - oracle| max", np.nanmax(np.abs(b - want)))
            e = envs[0]
            k = idx[idx[:, 2] == e][:6]
            for (en, f, _) in k:
                print(f"   env {e} entity {en} field {f}: launches {a[en, f, e]!r} lazy {b[en, f, e]!r} oracle {want[en, f, e]!r} pre {st_s[en, f, e]!r}")
        break
