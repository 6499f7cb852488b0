"""The C oracle against the LIVE reference at batch sizes the golden fixtures do not cover (they are 4-16 envs).

For the five BASELINE worlds (+ waterfall for joints) a reference environment of >= 2048 envs is driven with seeded
random actions; around selected steps the reference's own World.step is bracketed: (state, agent forces) before,
state after.  The oracle is run teacher-forced from the same inputs with the reference's broad-phase semantics
(batch-global mask per substep) and must land on the reference's state within the north-star tolerance
(abs 1e-5 + rel 1e-5) with NO sensitivity allowance - strict (waterfall, whose joint links have I ~ 1e-4, gets the measured
1-ulp-libm conditioning of golden_util.ulp_sensitivity and reports how many values needed it: a few tens in 4 x 10^6); worlds with several substeps are teacher-forced per
substep (the state entering each substep is recorded at core.py:2006).  CPU only, a few seconds per world."""
import numpy as np
import pytest
import torch

from golden_util import compare_state, ulp_sensitivity
from ref_backend import pack_ft, pack_state, per_env_arrays

pytestmark = pytest.mark.reference

SENSITIVE = ("waterfall",)  # light jointed links: one correct libm differs from another by more than 1e-5 of ang_vel
CASES = [
    ("balance", dict(n_agents=4), 2048, 24),
    ("balance", dict(n_agents=3), 2048, 12),
    ("transport", dict(), 2048, 24),
    ("transport", dict(n_packages=2), 2048, 16),
    ("navigation", dict(n_agents=8), 2048, 12),
    ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), 2048, 12),
    ("waterfall", dict(), 2048, 16),
]


@pytest.mark.parametrize("scenario,kw,B,steps", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_oracle_matches_live_reference_large_batch(scenario, kw, B, steps):
    from oracle import ref
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.spec import spec_from_world

    torch.manual_seed(0)
    torch.set_num_threads(8)
    env = ref.make_env(scenario, num_envs=B, device="cpu", seed=0, continuous_actions=True, **kw)
    w = env.world
    spec = spec_from_world(w)
    o = Oracle(spec)
    orig_step, orig_env_force = w.step, w._apply_vectorized_enviornment_force
    rec = {}

    def env_force():  # called once per substep (core.py:2006): the state entering that substep
        rec["sub"].append(pack_state(w))
        return orig_env_force()

    def step():
        rec["sub"], rec["ft_in"] = [], pack_ft(w)
        rec["jfr"], rec["eg"] = per_env_arrays(w, spec)
        orig_step()
        rec["sub"].append(pack_state(w))
        rec["ft_out"] = pack_ft(w)

    w.step, w._apply_vectorized_enviornment_force = step, env_force
    g = torch.Generator().manual_seed(1234)
    worst, checked, stats = 0.0, 0, {}
    for t in range(steps):
        acts = [(torch.rand(B, a.action_size, generator=g) * 2 - 1) * a.action.u_range_tensor for a in env.agents]
        env.step(acts)
        if t % 4 != 3:
            continue
        assert len(rec["sub"]) == spec.substeps + 1
        ft = rec["ft_in"].copy()
        for s in range(spec.substeps):  # teacher-forced per substep, the reference's batch-global broad phase
            st = rec["sub"][s].copy()
            kws = dict(batch=B, pair_mask=o.pair_mask(st, B), joint_fixed_rot=rec["jfr"], entity_gravity=rec["eg"],
                       first_substep=s, n_substeps=1, threads=8)
            sens = ulp_sensitivity(lambda a, b: o.step(a, b, **kws), st, ft) if scenario in SENSITIVE else None
            o.step(st, ft, **kws)
            worst = max(worst, compare_state(st, rec["sub"][s + 1], f"{scenario}[t={t},s={s}] oracle vs live reference, "
                                             f"{B} envs", atol=1e-5, rtol=1e-5, sens=sens, stats=stats))
        if ft.size:
            compare_state(ft, rec["ft_out"], f"{scenario}[t={t}] clamped agent force/torque", atol=1e-6, rtol=1e-6)
        checked += 1
    assert checked >= 3
    print(f"{scenario} {kw} {B} envs: max |oracle - reference| = {worst:.2e} over {checked} steps; "
          f"{stats['needed_sens']} of {stats['values']} values beyond the plain 1e-5 tolerance")
    if scenario not in SENSITIVE:
        assert stats["needed_sens"] == 0
    else:
        assert stats["needed_sens"] <= 1e-4 * stats["values"]


@pytest.mark.parametrize("crowd", [1.0, 0.25, 0.05])
def test_oracle_lidar_matches_live_reference_on_crowded_rotated_worlds(crowd):
    """World.cast_rays (core.py:1662-1786, ray - sphere core.py:1414-1490) of navigation n_agents=8 at 1024 environments on
    states the fixtures do not hold: positions scaled towards the origin (crowd 0.25: many agents within range of each
    other; 0.05: agents OVERLAP, so sensors sit inside other agents' spheres and the measured distance goes negative) and
    every agent rotated at random (the sensor's angles rotate with it, sensors.py:118).  The oracle's cast must be the
    reference's within 1e-5 - the states the lane-compacted HIP cast is pinned on through the plain kernel."""
    from oracle import ref
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.spec import spec_from_world

    B = 1024
    torch.manual_seed(3)
    env = ref.make_env("navigation", num_envs=B, device="cpu", seed=3, n_agents=8)
    w = env.world
    g = torch.Generator().manual_seed(7)
    for a in w.agents:
        a.set_pos(a.state.pos * crowd, batch_index=None)
        a.set_rot((torch.rand(B, 1, generator=g) * 2 - 1) * 3.0, batch_index=None)
    spec = spec_from_world(w)
    o = Oracle(spec)
    ents = list(w.entities)
    L = spec.lidars
    assert len(L) == 8
    got = o.cast_rays(pack_state(w), batch=B, threads=8)
    k, worst, below_zero = 0, 0.0, 0
    for agent in w.agents:
        for sensor in agent.sensors:
            if not hasattr(sensor, "_angles"):
                continue
            m = w.cast_rays(agent, sensor._angles + agent.state.rot, max_range=sensor._max_range, entity_filter=sensor.entity_filter)
            assert ents.index(agent) == L[k].entity
            want = m.T.numpy()
            below_zero += int((want < 0).sum())
            worst = max(worst, compare_state(got[k, : want.shape[0], :B], want, f"navigation crowd={crowd} sensor {k}", atol=1e-5, rtol=1e-5))
            k += 1
    assert k == 8
    if crowd <= 0.05:
        assert below_zero > 0, "the crowded case is meant to put sensors inside spheres"
    print(f"navigation lidar, crowd {crowd}: max |oracle - reference| = {worst:.2e}, {below_zero} negative distances")
