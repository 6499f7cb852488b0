"""Loading of the committed reference fixtures (tests/golden/*.npz, written by
tests/golden/make_golden.py from the reference itself)."""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from vectorizedmultiagentsimulator_amd.spec import WorldSpec

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_ALL = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
FIXTURES = [n for n in _ALL if not n.startswith("envstep_")]  # World.step fixtures
ENVSTEP_FIXTURES = [n for n in _ALL if n.startswith("envstep_")]  # Environment.step outputs (obs/rew/done)


@dataclass
class Golden:
    name: str
    spec: WorldSpec
    state0: np.ndarray  # [T, E, 6, B]
    ft_in: np.ndarray  # [T, A, 3, B]
    masks: np.ndarray  # [T, S, W] uint32
    state1: np.ndarray
    ft_out: np.ndarray
    jfr: Optional[np.ndarray]  # [T, J, B]
    egrav: Optional[np.ndarray]  # [T, E, 2, B]
    lidar: Optional[np.ndarray]  # [T, L, R, B]
    sub: Optional[np.ndarray] = None  # [T, S+1, E, 6, B]
    query: Optional[np.ndarray] = None  # [T, Q, B]
    queries: Optional[list] = None  # [(kind, a, b)]

    @property
    def T(self):
        return self.state0.shape[0]

    @property
    def B(self):
        return self.state0.shape[-1]


def load(name: str) -> Golden:
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    spec = WorldSpec.from_json(str(z["spec"]))
    g = lambda k: (z[k] if k in z.files else None)  # noqa: E731
    jfr = g("jfr")
    if jfr is not None and jfr.shape[1] == 0:
        jfr = None
    return Golden(name, spec, z["state0"], z["ft_in"], z["masks"], z["state1"], z["ft_out"], jfr, g("egrav"), g("lidar"), g("sub"), g("query"),
                  [tuple(q) for q in __import__("json").loads(str(z["queries"]))] if "queries" in z.files else None)


def tolerances(spec: WorldSpec):
    """Per-field absolute tolerance for ONE teacher-forced step.

    north_star: 1e-5 fp32.  pos/vel/rot are O(1) quantities -> abs 1e-5.  ang_vel of
    very light bodies is the one amplified quantity: d(ang_vel) = torque / I * dt with
    I as small as m*L^2/12 ~ 1e-4 for joint links, so its
    tolerance is 1e-5 relative to the magnitude (abs 1e-5 + rel 1e-5 * |x|), and the
    rot it integrates into follows with the same relative slack.
    """
    return dict(atol=1e-5, rtol=1e-5)


def ulp_sensitivity(step_fn, state: np.ndarray, ft: np.ndarray, n: int = 3, seed: int = 0) -> np.ndarray:
    """Elementwise conditioning of one (sub)step at fp32 resolution.

    ``step_fn(state, ft)`` advances copies in place (the oracle).  Every input is moved
    by exactly one ulp in a random direction and every sin/cos/exp/log1p RESULT inside the
    oracle by -1/0/+1 ulp (three correct libms - SLEEF in torch, glibc here, ocml on the
    GPU - differ in the last bit), ``n`` times; the result is the largest
    deviation of each output element from the unperturbed run.  Two *correct* fp32
    implementations (different libm for sin/cos/exp/log1p) cannot agree better than a
    small multiple of this: e.g. a joint anchor difference ``pos_a + R(rot_a) d_a -
    (pos_b + R(rot_b) d_b)`` cancels to ~1e-3, so a 1-ulp change of cos(rot) moves the
    force direction by 1e-4 relative, and a link with I = m L^2/12 ~ 1e-4 turns that
    into 1e-5..1e-4 of ang_vel in ONE substep (SURVEY.md App. C-2 measured the same
    on the reference itself: waterfall 1.2e-3 after one step).
    """
    from oracle.oracle import set_jitter

    rng = np.random.default_rng(seed)
    base_s, base_f = state.copy(), ft.copy()
    step_fn(base_s, base_f)
    sens = np.zeros_like(base_s)
    try:
        for i in range(n):
            # (a) inputs moved by one ulp, (b) libm results moved by -1/0/+1 ulp
            s2 = np.nextafter(state, np.where(rng.random(state.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
            f2 = np.nextafter(ft, np.where(rng.random(ft.shape) < 0.5, -np.inf, np.inf).astype(np.float32))
            s2, f2 = np.ascontiguousarray(s2, np.float32), np.ascontiguousarray(f2, np.float32)
            set_jitter(0x9E3779B1 + 7919 * (i + 1) + seed)
            step_fn(s2, f2)
            with np.errstate(invalid="ignore"):
                sens = np.fmax(sens, np.abs(s2 - base_s))
    finally:
        set_jitter(0)
    return sens


#: fixtures of the five BASELINE configs: parity must hold at the plain north-star tolerance, no allowance used
BASELINE_FIXTURES = ("balance_n3", "balance_n4", "transport", "transport_2pkg", "navigation_n8", "football_5v5")


def compare_state(got: np.ndarray, want: np.ndarray, name: str, atol=1e-5, rtol=1e-5, sens=None, sens_mult=8.0, stats=None):
    """|got - want| <= atol + rtol*|want| (+ sens_mult * 1-ulp conditioning if given).  ``stats`` (a dict) counts in
    ``stats["needed_sens"]`` how many values were only accepted thanks to the conditioning allowance and in
    ``stats["values"]`` how many were compared."""
    with np.errstate(invalid="ignore"):
        err = np.abs(got - want)
    lim = atol + rtol * np.abs(want)
    if stats is not None:
        stats["values"] = stats.get("values", 0) + int(err.size)
        stats["needed_sens"] = stats.get("needed_sens", 0) + int((err > lim).sum())
        # the largest error among values of physical magnitude, and how many are not: a blown-up environment (the reference
        # itself reaches 1e24 in the dense `soup_*` worlds) makes the plain maximum meaningless as a parity figure
        with np.errstate(invalid="ignore"):
            physical = np.isfinite(want) & np.isfinite(got) & (np.abs(want) < 1e3)
        stats["blown_up"] = stats.get("blown_up", 0) + int(err.size - physical.sum())
        if physical.any():
            stats["worst_physical"] = max(stats.get("worst_physical", 0.0), float(err[physical].max()))
    if sens is not None:
        lim = lim + sens_mult * sens
    bad = err > lim
    # NaN != NaN
    bad |= np.isnan(got) != np.isnan(want)
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(
            f"{name}: {int(bad.sum())} values off; first at {tuple(idx)} got {got[tuple(idx)]!r} "
            f"want {want[tuple(idx)]!r}; max abs err {np.nanmax(err):.3e}"
        )
    return float(np.nanmax(err)) if err.size else 0.0
