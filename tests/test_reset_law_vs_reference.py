"""The reset LAW of the native scenarios against the reference's own ``reset_world_at`` (round-2 review, f4).

The GPU tests tie the masked-reset kernel (vmas_env_reset_where) to this package's own vectorised reset
(tests/test_env_fused_gpu.py: constraints + moments on 20 000 environments).  The missing link - that this reset samples
what the REFERENCE samples - is closed here on the CPU: 20 000 environments reset by the reference (its
``ScenarioUtils.spawn_entities_randomly`` / ``find_random_pos_for_entity`` rejection loop, utils.py:241-319, and each
scenario's ``reset_world_at``: balance.py:86-216, transport.py:87-129, navigation.py:137-198, football.py:162-171) against
20 000 environments reset by the native scenario classes (whose rejection runs a fixed number of device-side rounds
instead of ``while torch.any(overlaps)``): per entity the first two moments and the range of x and y, the constraints of
the placement (minimum distances, offsets), and the pairwise mean distances - independent random streams, so compared
within the sampling error of 20 000 draws.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference

N = 20000
CASES = [("balance", dict(n_agents=4)), ("transport", {}), ("transport", dict(n_packages=2)), ("navigation", dict(n_agents=8)),
         ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False))]


@pytest.fixture(scope="module")
def vmas():
    from oracle import ref

    ref.import_vmas()
    return ref


def _stats(world):
    out = {}
    for e in world.entities:
        p = e.state.pos.double().cpu().numpy()
        r = e.state.rot.double().cpu().numpy()
        out[e.name] = dict(mean=p.mean(0), std=p.std(0), lo=p.min(0), hi=p.max(0), rot=(r.min(), r.max()), pos=p)
    return out


@pytest.mark.parametrize("name,kw", CASES)
def test_native_reset_samples_the_reference_law(vmas, name, kw):
    from vectorizedmultiagentsimulator_amd.environment import make_env

    ref = vmas.make_env(name, num_envs=N, device="cpu", seed=1, **kw)
    ours = make_env(name, num_envs=N, device="cpu", seed=2, **kw)
    a, b = _stats(ref.world), _stats(ours.world)
    assert list(a) == list(b), "same entities in the same order"
    se = 5.5 * math.sqrt(2.0 / N)  # 5.5 standard errors of the DIFFERENCE of two means of N draws (per unit of standard deviation)
    for nm in a:
        ra, rb = a[nm], b[nm]
        scale = np.maximum(np.maximum(ra["std"], rb["std"]), 1e-9)
        fixed = (ra["std"] < 1e-7).all()
        if fixed:  # a fixed pose (walls, floor): identical numbers
            assert np.allclose(ra["mean"], rb["mean"], atol=1e-6) and (rb["std"] < 1e-6).all(), f"{name}: {nm} is fixed in the reference"
        else:
            assert (np.abs(ra["mean"] - rb["mean"]) <= se * scale + 1e-6).all(), f"{name}: mean of {nm}: {ra['mean']} vs {rb['mean']}"
            assert (np.abs(ra["std"] - rb["std"]) <= 2 * se * scale + 1e-6).all(), f"{name}: std of {nm}: {ra['std']} vs {rb['std']}"
            span = np.maximum(ra["hi"] - ra["lo"], 1e-9)
            assert (np.abs(ra["lo"] - rb["lo"]) <= 0.01 * span + 1e-6).all() and (np.abs(ra["hi"] - rb["hi"]) <= 0.01 * span + 1e-6).all(), (
                f"{name}: range of {nm}: [{ra['lo']}, {ra['hi']}] vs [{rb['lo']}, {rb['hi']}]")
        assert np.allclose(ra["rot"], rb["rot"], atol=1e-6), f"{name}: rotation of {nm}: {ra['rot']} vs {rb['rot']}"
    # pairwise structure: mean and minimum distance of every pair of entities (the rejection law shows up in the minimum)
    names = list(a)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            da = np.linalg.norm(a[names[i]]["pos"] - a[names[j]]["pos"], axis=1)
            db = np.linalg.norm(b[names[i]]["pos"] - b[names[j]]["pos"], axis=1)
            sd = max(da.std(), db.std(), 1e-9)
            assert abs(da.mean() - db.mean()) <= se * sd + 1e-6, f"{name}: mean distance {names[i]} - {names[j]}: {da.mean()} vs {db.mean()}"
            if da.std() > 1e-7:
                # the lower tail (where a rejection law shows): the 1 % quantile - 200 of the 20 000 draws, ~7 % sampling noise -
                # and the hard minimum of a law with a minimum distance
                qa, qb = np.quantile(da, 0.01), np.quantile(db, 0.01)
                assert abs(qa - qb) <= 0.25 * max(qa, qb) + 5e-3, f"{name}: lower-tail distance {names[i]} - {names[j]}: {qa} vs {qb}"


@pytest.mark.parametrize("name,kw", [("navigation", dict(n_agents=8)), ("transport", dict(n_packages=2))])
def test_native_reset_keeps_the_minimum_distance_everywhere(vmas, name, kw):
    """The reference's rejection loop runs until NO environment overlaps; the native one runs a fixed number of rounds.  At
    20 000 environments not one placement may violate the minimum distance."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    ours = make_env(name, num_envs=N, device="cpu", seed=5, **kw)
    prog = ours.scenario.fused_reset_program()
    placed = []
    for op in prog["ops"]:
        if op[0] != "uniform":
            continue
        _, ent, xb, yb, min_dist, avoid_from = op[:6]
        p = ent.state.pos
        for q in placed[avoid_from:]:
            assert (torch.linalg.vector_norm(p - q, dim=1) >= min_dist).all(), f"{name}: {ent.name} closer than {min_dist}"
        placed.append(p)
