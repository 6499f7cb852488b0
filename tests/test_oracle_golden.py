"""Pins the CPU oracle (oracle/vmas_oracle.c) against the reference's own outputs.

Each fixture holds (state0, forces, per-substep broad-phase masks) -> state1 tuples
recorded from the unmodified reference (tests/golden/make_golden.py).  The oracle is
run teacher-forced from state0 through all substeps of ONE World.step and must land
on state1 within 1e-5 (north_star tolerance; see golden_util.tolerances).
"""
import numpy as np
import pytest

from golden_util import FIXTURES, compare_state, load, tolerances, ulp_sensitivity
from oracle.oracle import Oracle


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_step_matches_reference(name):
    g = load(name)
    o = Oracle(g.spec)
    worst = 0.0
    for t in range(g.T):
        st = np.ascontiguousarray(g.state0[t]).copy()
        ft = np.ascontiguousarray(g.ft_in[t]).copy()
        jfr = None if g.jfr is None else np.ascontiguousarray(g.jfr[t])
        eg = None if g.egrav is None else np.ascontiguousarray(g.egrav[t])
        for s in range(g.spec.substeps):
            if g.sub is not None:  # teacher-force every substep
                st = np.ascontiguousarray(g.sub[t, s]).copy()
            kw = dict(pair_mask=np.ascontiguousarray(g.masks[t, s]), joint_fixed_rot=jfr, entity_gravity=eg,
                      first_substep=s, n_substeps=1)
            sens = ulp_sensitivity(lambda a, b: o.step(a, b, **kw), st, ft)
            o.step(st, ft, **kw)
            want = g.state1[t] if g.sub is None else g.sub[t, s + 1]
            worst = max(worst, compare_state(st, want, f"{name}[t={t},s={s}] state", sens=sens,
                                             **tolerances(g.spec)))
        compare_state(ft, g.ft_out[t], f"{name}[t={t}] agent force/torque", atol=1e-6, rtol=1e-6)
    print(f"{name}: max abs err {worst:.3e}")


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_pair_mask_matches_reference(name):
    """The oracle's batch-global broad phase reproduces the reference's
    World.collides decisions on the recorded entry state (substep 0)."""
    g = load(name)
    o = Oracle(g.spec)
    for t in range(g.T):
        m = o.pair_mask(np.ascontiguousarray(g.state0[t]))
        assert np.array_equal(m[: g.masks.shape[-1]], g.masks[t, 0]), f"{name}[t={t}]"


@pytest.mark.parametrize("name", [n for n in FIXTURES if load(n).lidar is not None])
def test_oracle_lidar_matches_reference(name):
    g = load(name)
    o = Oracle(g.spec)
    for t in range(g.T):
        out = o.cast_rays(np.ascontiguousarray(g.state0[t]))
        compare_state(out, g.lidar[t], f"{name}[t={t}] lidar", atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_queries_match_reference(name):
    """World.get_distance / World.is_overlapping (core.py:1822-1969) for every shape combination."""
    g = load(name)
    if not g.queries:
        pytest.skip("no query pairs")
    o = Oracle(g.spec)
    kinds = np.array([k == "overlap" for k, _, _ in g.queries])
    flips = 0
    for t in range(g.T):
        st = np.ascontiguousarray(g.state0[t])
        with np.errstate(invalid="ignore"):
            ok = np.isfinite(st).all(axis=(0, 1)) & (np.abs(st) < 1e3).all(axis=(0, 1))  # sane environments only
        out = o.queries(st, g.queries)[:, : g.B]
        want = g.query[t]
        compare_state(out[~kinds][:, ok], want[~kinds][:, ok], f"{name}[t={t}] distances", atol=2e-6, rtol=1e-5)
        flips += int((out[kinds][:, ok] != want[kinds][:, ok]).sum())
    # an overlap flag may flip only where the distance is within rounding of zero
    assert flips <= 2, f"{name}: {flips} overlap flags differ"


def test_band_fixture_pins_the_batch_global_broad_phase():
    """`band_4env` (made by the reference): every environment has a sphere just beyond the end of a line / the corner
    of a box - outside the bounding circles, inside the contact distance.  Free-running with the batch-global broad
    phase (step_exact) the oracle lands on the reference's numbers in every step; evaluating every pair per environment
    instead (the large-batch default) gives a visibly different velocity on the steps where NO environment overlaps -
    which is why small batches run exact by default (core.EXACT_AUTO_BELOW)."""
    g = load("band_4env")
    o = Oracle(g.spec)
    differs = 0
    for t in range(g.T):
        st, ft = np.ascontiguousarray(g.state0[t]).copy(), np.ascontiguousarray(g.ft_in[t]).copy()
        o.step_exact(st, ft)
        compare_state(st, g.state1[t], f"band_4env[t={t}] exact free-running", atol=1e-5, rtol=1e-5)
        st2, ft2 = np.ascontiguousarray(g.state0[t]).copy(), np.ascontiguousarray(g.ft_in[t]).copy()
        o.step(st2, ft2)  # no mask: every static pair evaluated per environment
        err = np.abs(st2 - g.state1[t]).max()
        if not g.masks[t].any():
            assert err > 1e-2, f"t={t}: per-environment evaluation should differ in the band, max err {err:.2e}"
            differs += 1
    assert differs == g.T // 2
