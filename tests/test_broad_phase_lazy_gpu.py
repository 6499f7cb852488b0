"""The reference's broad-phase rule in the path that is actually benchmarked (round-5 review, row a6 / weak #1).

``World.collides`` (core.py:2788-2803) drops a pair for the WHOLE batch iff no environment's bounding circles overlap,
re-decided at every substep.  Evaluating every static pair per environment - every default above 1 024 environments until
round 5 - differs from that in an environment that sits in a pair's band (circles apart, narrow-phase force non-zero) while
no environment of the batch overlaps, and at the BASELINE sizes that DOES happen (football 8 192: two states of three).

Since round 6 every host path asks for the rule itself and the library runs its LAZY form inside the step launch
(csrc/vmas_env_device.h): optimistic passes, the batch's words only for a tile that has to know.  Checked here:

* against the LIVE reference (the byte-compiled copy under oracle/_ref, run on the host cores of the GPU box): >= 100
  reference steps per configuration at 8 192 / 16 384 / full size, each replayed teacher-forced as ONE native World.step -
  strict 1e-5 abs + 1e-5 rel, ZERO values beyond it - with the per-environment form on the same inputs as the NEGATIVE
  control (it must trip over the events the review found);
* the waiting path: a batch whose environments ALL sit in a band with the pair off for the whole batch (every tile has to
  wait for every tile and make its pass again) at one tile per CU, and beyond what is resident (no hang: the wait is bounded);
* bitwise the launch-per-substep form (pair_mask launch + one-substep launches with that mask) - in tests/test_hip_parity.py,
  whose exact-form tests now run this form.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from golden_util import load

pytestmark = pytest.mark.gpu

FOOTBALL = dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)
# (scenario, kwargs, environments, reference steps, the per-environment form must differ somewhere)
CONFIGS = [
    ("balance", dict(n_agents=4), 32768, 100, False),       # BASELINE config 2 (the review found no event there)
    ("transport", {}, 16384, 100, True),                    # config 3
    ("transport", dict(n_packages=2), 16384, 100, None),    # config 3 with box-box pairs
    ("football", FOOTBALL, 8192, 100, True),                # the driver line's parity sample size
    ("football", FOOTBALL, 16384, 100, True),               # config 5's per-GPU shard
    ("football", FOOTBALL, 131072, 12, None),               # config 5 on one GPU: more tiles than are resident
]


def _record(line):
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "broad_phase_lazy.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass


def _beyond(got, want, tol=1e-5):
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    sane = np.isfinite(err) & (np.abs(want) < 1e3)
    bad = sane & (err > tol + tol * np.abs(want))
    return int(bad.sum()), float(err[sane].max()) if sane.any() else 0.0, int(sane.sum())


@pytest.mark.reference
@pytest.mark.parametrize("name,kw,B,steps,control_differs", CONFIGS, ids=[f"{c[0]}{'-' + str(len(c[1])) if c[1] else ''}-{c[2]}" for c in CONFIGS])
def test_default_step_follows_the_reference_rule_on_reference_driven_states(name, kw, B, steps, control_differs):
    from oracle import ref
    from ref_backend import pack_ft, pack_state
    from vectorizedmultiagentsimulator_amd.backend import HipWorld
    from vectorizedmultiagentsimulator_amd.spec import spec_from_world

    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    env = ref.make_env(name, num_envs=B, device="cpu", seed=0, continuous_actions=True, **kw)
    w = env.world
    spec = spec_from_world(w)
    hw = HipWorld(spec, B, "cuda:0")
    assert hw.exact_form() == 1, "a world with line / box pairs runs the lazy form inside the launch"
    rec = {}
    orig_step = w.step

    def step():
        rec["pre"], rec["ft"] = pack_state(w), pack_ft(w)
        orig_step()
        rec["post"] = pack_state(w)

    w.step = step
    g = torch.Generator().manual_seed(1234)
    nA = spec.n_agents
    tot = dict(values=0, beyond=0, worst=0.0, control_beyond=0, control_worst=0.0, control_states=0)
    t0 = time.time()
    with torch.no_grad():
        for t in range(steps):
            acts = [(torch.rand(B, a.action_size, generator=g) * 2 - 1) * a.action.u_range_tensor for a in env.agents]
            env.step(acts)
            pre, ft, post = torch.from_numpy(rec["pre"]), torch.from_numpy(rec["ft"]), rec["post"]
            for form in ("exact", "per_environment"):
                hw.state[:, :, :B].copy_(pre)
                if nA:
                    hw.agent_ft[:nA, :, :B].copy_(ft)
                hw.step(exact=(form == "exact"))
                got = hw.state[:, :, :B].cpu().numpy()
                n_bad, worst, n = _beyond(got, post)
                if form == "exact":
                    tot["values"] += n; tot["beyond"] += n_bad; tot["worst"] = max(tot["worst"], worst)
                else:
                    tot["control_beyond"] += n_bad; tot["control_worst"] = max(tot["control_worst"], worst)
                    tot["control_states"] += 1 if n_bad else 0
    assert hw.exact_status() == 0, "a tile gave up waiting for the batch's words"
    _record({"scenario": name, "kw": {k: str(v) for k, v in kw.items()}, "envs": B, "reference_steps": steps,
             "seconds": round(time.time() - t0, 1), **tot})
    assert tot["beyond"] == 0, (f"{name} {kw} x {B}: {tot['beyond']} of {tot['values']} values beyond 1e-5 between the default step "
                                f"and the reference over {steps} reference steps (max {tot['worst']:.2e})")
    if control_differs is True:
        assert tot["control_beyond"] > 0, "the per-environment form was expected to leave the reference's trajectory here"
    if control_differs is False:
        assert tot["control_beyond"] == 0
    hw.close()


@pytest.mark.parametrize("B", [64, 16384, 200000])
def test_every_tile_waits_and_makes_its_pass_again(B):
    """`band_4env` (made by the reference: a sphere just beyond the end of a line and off the corner of a box - outside the
    bounding circles, inside the contact distance) tiled to B environments, steps where NO environment overlaps: every tile
    has a band event on a pair that is off for the whole batch, so every tile waits for every tile and gathers again.  One
    tile; one tile per CU; and more tiles than are resident at once - where a waiting tile keeps its CU from tiles it waits
    for: the wait is bounded, the call after says so instead of hanging, and a batch that is NOT adversarial goes through."""
    from vectorizedmultiagentsimulator_amd.backend import HipWorld, VmasHipError

    g = load("band_4env")
    hw = HipWorld(g.spec, B, "cuda:0")
    assert hw.exact_form() == 1
    nA = g.spec.n_agents
    rep = lambda a: torch.from_numpy(np.ascontiguousarray(a)).repeat_interleave((B + g.B - 1) // g.B, dim=-1)[..., :B]  # noqa: E731
    checked = 0
    for t in range(g.T):
        if g.masks[t].any():
            continue  # (only the steps whose pairs are all off for the whole batch)
        hw.state[:, :, :B].copy_(rep(g.state0[t]))
        hw.agent_ft[:nA, :, :B].copy_(rep(g.ft_in[t]))
        t0 = time.time()
        hw.step(exact=True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        status = hw.exact_status()
        if B <= 16384:
            assert status == 0
            got = hw.state[:, :, :B].cpu()
            want = rep(g.state1[t])
            n_bad, worst, _ = _beyond(got.numpy(), want.numpy())
            assert n_bad == 0, f"band_4env x {B} step {t}: {n_bad} values beyond 1e-5 (max {worst:.2e})"
            checked += 1
        else:
            assert dt < 60.0, "a grid beyond what is resident must not hang"
            if status != 0:  # every resident tile waited for tiles that could not start: flagged, and the next call fails loudly
                with pytest.raises(VmasHipError, match="gave up"):
                    hw.step(exact=True)
            checked += 1
    assert checked >= 1
    hw.close()


def test_a_gated_launch_carries_the_lazy_form_and_a_refused_one_keeps_the_slots_clean():
    """`vmas_world_step_env_gated` (the reference's action asserts without an idle queue) with the exact broad phase: allowed
    since the lazy form has no barrier count that the host advances.  A refused launch does not publish - but it zeroes the set
    of slots the NEXT launch uses, so the steps behind it still see the batch's words: refuse one step in a run of exact steps
    on the band fixture and require the reference's numbers after it."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    env = make_env("balance", num_envs=2048, device="cuda:0", seed=3, n_agents=4)
    assert env.world.exact_broad_phase and env.world._get_backend().exact_form() == 1
    twin = make_env("balance", num_envs=2048, device="cuda:0", seed=3, n_agents=4, exact_broad_phase=True)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for k in range(12):
        acts = [(torch.rand(2048, 2, device="cuda:0", generator=g) * 2 - 1) for _ in env.agents]
        if k == 5:
            bad = [a.clone() for a in acts]
            bad[1][7, 0] = float("nan")
            with pytest.raises(AssertionError):
                env.step(bad)
        o1 = env.step(acts)
        o2 = twin.step(acts)
        assert torch.equal(torch.stack(o1[0]), torch.stack(o2[0])), f"step {k}"
    assert env.world._get_backend().exact_status() == 0
