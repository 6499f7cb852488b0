"""The DEFAULT broad phase at BASELINE sizes against the reference's batch-global rule (round-2 review, weak #2).

Above ``core.EXACT_AUTO_BELOW`` environments ``World.step()`` evaluates every static pair per environment; the reference
processes a pair - for all environments - iff SOME environment of the batch has the pair's bounding circles overlapping
(``World.collides``, core.py:2797-2801).  The two agree whenever, for every pair that exerts a force in some environment,
some environment of the batch overlaps - "at large batches always", which this file CHECKS instead of arguing:

for BASELINE configs 2-5 at their full sizes, on states taken from real 100-step rollouts of the scenario (reset law +
random actions + the scenario's own dynamics, not jittered fixture columns), the product's default step on the GPU is
compared, over ALL environments, with the oracle's ``step_exact`` (mask = the reference's rule, re-decided per substep).
Values beyond 1e-5 must number ZERO.  Also recorded: which static pairs no environment overlaps (skipped by the reference
for the whole batch) and the largest force the per-environment evaluation finds for those pairs (recorded in
gpurun_out/broad_phase_full_size.jsonl; 0 means the two rules are not merely close but identical on that state).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = [  # BASELINE.json configs 2-5, full size (4 and 5 are 8-GPU configs: whole batch AND the per-GPU shard)
    ("balance", dict(n_agents=4), 32768),
    ("transport", {}, 16384),
    ("transport", dict(n_packages=2), 16384),
    ("navigation", dict(n_agents=8), 65536),
    ("navigation", dict(n_agents=8), 8192),
    ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), 131072),
    ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), 16384),
]


def _record(line):
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..")), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "broad_phase_full_size.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name,kw,B", CONFIGS)
def test_default_broad_phase_equals_the_batch_global_rule_on_rollout_states(name, kw, B):
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.core import EXACT_AUTO_BELOW
    from vectorizedmultiagentsimulator_amd.environment import make_env

    env = make_env(name, num_envs=B, device="cuda:0", seed=7, validate_actions=False, **kw)
    w = env.world
    spec, be = w.spec, w._get_backend()
    o = Oracle(spec)
    threads = min(os.cpu_count() or 8, 64)
    g = torch.Generator(device="cuda:0").manual_seed(77)
    checked = 0
    for t in range(101):
        acts = [(torch.rand(B, env.get_agent_action_size(a), device="cuda:0", generator=g) * 2 - 1)
                * a.action.u_range_tensor_on(env.device) for a in env.agents]
        if t in (0, 1, 25, 100):  # right after the reset, early, mid-rollout, at its end
            # the step under test, from this rollout state: ingest the actions (so that agent_ft is what a step would use),
            # then the DEFAULT physics step of a world of this size on a copy of the state
            env._ingest(acts, False) if env._ingest is not None else env._ingest_torch(acts)
            st0 = w._state.clone()
            ft0 = w._agent_ft.clone()
            st_np = st0[:, :, :B].cpu().numpy().copy()
            ft_np = ft0[: spec.n_agents, :, :B].cpu().numpy().copy()
            assert B < EXACT_AUTO_BELOW or not w.exact_broad_phase  # the default at this size: per-environment evaluation
            be.step()  # == World.step() of the default configuration
            got = w._state[:, :, :B].cpu().numpy()
            w._state.copy_(st0)
            w._agent_ft.copy_(ft0)
            want = st_np.copy()
            masks = []
            for s in range(spec.substeps):  # Oracle.step_exact, keeping the masks
                m = o.pair_mask(want, B)
                masks.append(m.copy())
                o.step(want, ft_np, B, m, None, None, s, 1, threads)
            err = np.abs(got - want)
            lim = 1e-5 + 1e-5 * np.abs(want)
            n_bad = int((err > lim).sum())
            # pairs the reference skips for the whole batch (no environment's bounding circles overlap), first substep
            off = [p for p in range(len(spec.pairs)) if not (masks[0][p >> 5] >> (p & 31)) & 1]
            worst_off = 0.0
            for p in off:
                f = o.pair_forces(st_np, p, B)
                worst_off = max(worst_off, float(np.nanmax(np.abs(f))))
            _record({"scenario": name, "kw": {k: str(v) for k, v in kw.items()}, "envs": B, "rollout_step": t,
                     "values": int(err.size), "beyond_1e-5": n_bad, "max_abs_err": float(err.max()),
                     "pairs": len(spec.pairs), "pairs_no_env_overlaps": len(off), "max_force_of_those_pairs": worst_off})
            assert n_bad == 0, (f"{name} B={B} rollout step {t}: {n_bad} of {err.size} values beyond 1e-5 between the default "
                                f"(per-environment) broad phase and the reference's batch-global rule (max {err.max():.2e})")
            checked += 1
        env.step(acts)
    assert checked == 4
