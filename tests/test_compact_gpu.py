"""The lane-compacted step kernel (csrc/vmas_compact.h: broad phase per environment, narrow phase per CONTACT) against
the schedule interpreter, bit for bit, and against the reference's recorded numbers.

Every golden fixture whose world qualifies (pairs all sphere-sphere / line-sphere, no joints: 15 of the 45 - football,
navigation, a rotating line with torques (wheel), friction and force ranges (give_way), per-environment gravity
(wind_flocking), torque dynamics (drone, diff_drive) ...) is stepped with the kernel forced on and off.  The golden
teacher-forced tests (test_hip_parity.py) run football through it by default."""
import numpy as np
import pytest
import torch

from golden_util import FIXTURES, compare_state, load, tolerances, ulp_sensitivity
from test_hip_parity import _dev, _down, _hip, _up, make_batch

pytestmark = pytest.mark.gpu


def _eligible(name):
    s = load(name).spec
    return (not s.joints) and len(s.pairs) > 0 and all(p.type in (0, 1) for p in s.pairs) and len(s.entities) <= 64


ELIGIBLE = [n for n in FIXTURES if _eligible(n)]


def _bits(t):
    return t.contiguous().view(torch.int32)


def _bits_nan(t):
    """Bits with every NaN made the same NaN: which NaN (sign, payload) a sum of several NaNs carries depends on how the sum
    is associated, and step_kernel adds an entity's item list segment by segment while the compacted kernel adds term by
    term in the reference's order - on finite values the two agree bit for bit, on NaNs both are NaN."""
    return _bits(torch.where(torch.isnan(t), torch.full_like(t, float("nan")), t))


def _pair(g, B, lanes=0):
    a, b = _hip(g.spec, B, lanes), _hip(g.spec, B, lanes)
    a.set_compact(1)
    b.set_compact(0)
    assert a.compact and not b.compact
    return a, b


def test_eligible_fixture_list():
    assert {"football_5v5", "navigation_n8", "all_wheel", "give_way", "wind_flocking"} <= set(ELIGIBLE) and len(ELIGIBLE) >= 12


@pytest.mark.parametrize("name", ELIGIBLE)
@pytest.mark.parametrize("B", [1000, 64 * 6])
def test_compact_kernel_is_bitwise_the_interpreter(name, B):
    """Free running for several steps (all substeps fused), batch with and without a tail tile; per-environment gravity
    where the fixture has it; states AND the clamped agent forces must carry the same bits."""
    g = load(name)
    st0, ft0, jfr_np, eg_np = make_batch(g, B, seed=31)
    assert jfr_np is None
    a, b = _pair(g, B)
    for hw in (a, b):
        _up(hw, st0, ft0)
    eg = _dev(a, eg_np, B)
    rng = np.random.default_rng(3)
    for t in range(6):
        f = torch.from_numpy((ft0 * (1 + 0.3 * rng.normal(0, 1, ft0.shape))).astype(np.float32))
        for hw in (a, b):
            if ft0.shape[0]:
                hw.agent_ft[: ft0.shape[0], :, :B].copy_(f)
            hw.step(entity_gravity=eg)
        assert torch.equal(_bits(a.state), _bits(b.state)), f"{name}: state differs at step {t}"
        assert torch.equal(_bits(a.agent_ft), _bits(b.agent_ft)), f"{name}: clamped forces differ at step {t}"


@pytest.mark.parametrize("name", ELIGIBLE)
def test_compact_kernel_matches_the_reference_fixture(name):
    """Teacher-forced on the reference's recorded (state, forces, recorded broad-phase mask) -> state, substep by substep:
    the kernel's recorded-mask and partial-substep paths against the reference's own numbers."""
    from oracle.oracle import Oracle

    g = load(name)
    o = Oracle(g.spec)
    hw = _hip(g.spec, g.B)
    hw.set_compact(1)
    T, S = g.state0.shape[0], g.spec.substeps
    for t in range(min(T, 6)):
        st0, ft0 = np.ascontiguousarray(g.state0[t]), np.ascontiguousarray(g.ft_in[t])
        _up(hw, st0, ft0)
        eg = _dev(hw, None if g.egrav is None else np.ascontiguousarray(g.egrav[t]), g.B)
        for s in range(S):
            mask = torch.from_numpy(np.ascontiguousarray(g.masks[t, s]).view(np.int32)).cuda()
            hw.step(pair_mask=mask, entity_gravity=eg, first_substep=s, n_substeps=1)
        got, _ = _down(hw, g.B, g.spec.n_agents)

        def ostep(a_, b_):
            for s in range(S):
                o.step(a_, b_, pair_mask=np.ascontiguousarray(g.masks[t, s]), first_substep=s, n_substeps=1,
                       entity_gravity=None if g.egrav is None else np.ascontiguousarray(g.egrav[t]))

        sens = ulp_sensitivity(ostep, st0, ft0)
        compare_state(got, g.state1[t], f"{name}[t={t}] compact kernel vs reference", sens=sens)


@pytest.mark.parametrize("name", ["football_5v5", "navigation_n8", "all_multi_give_way"])
@pytest.mark.parametrize("B", [700, 4096])
def test_compact_exact_broad_phase_forms(name, B):
    """The reference's batch-global broad phase: inside the launch (grid barrier per substep) == the explicit mask +
    substep launches, both bitwise the interpreter's."""
    g = load(name)
    st0, ft0, _, _ = make_batch(g, B, seed=5)
    a, b = _pair(g, B)
    c = _hip(g.spec, B)
    c.set_compact(1)
    for hw in (a, b, c):
        _up(hw, st0, ft0)
    for t in range(4):
        a.step_exact()
        b.step_exact()
        c.step_exact_launches()
        assert torch.equal(_bits(a.state), _bits(b.state)), f"{name}: in-launch exact, compact vs interpreter, step {t}"
        assert torch.equal(_bits(a.state), _bits(c.state)), f"{name}: in-launch vs launch-per-substep, step {t}"
    assert a.exact_status() == 0


@pytest.mark.parametrize("name", ["football_5v5", "navigation_n8"])
def test_compact_overflowing_tiles_take_the_rounds_path(name):
    """Non-finite poses pass every broad-phase test: a tile full of them needs more contact slots than the list has and
    is processed in rounds of CAP / 64 pairs.  NaN-for-NaN the interpreter's bits (the reference lets a non-finite pose
    poison every pair it is in)."""
    g = load(name)
    B = 64 * 5
    st0, ft0, _, _ = make_batch(g, B, seed=8)
    dyn = [i for i, e in enumerate(g.spec.entities) if e.flags & 3]
    st0[dyn[0], 0, 64:128] = np.nan          # one whole tile: a NaN x of one agent in every environment
    st0[dyn[1], 1, 128:192:2] = np.inf       # half of another
    st0[dyn[-1], 4, 200] = np.nan            # a lone NaN rotation elsewhere
    a, b = _pair(g, B)
    for hw in (a, b):
        _up(hw, st0, ft0)
    for t in range(3):
        a.step()
        b.step()
        assert torch.equal(_bits_nan(a.state), _bits_nan(b.state)), f"{name}: step {t}"
    assert torch.isnan(a.state[:, :, 64:128]).any() and torch.isfinite(a.state[:, :, :64]).all()


@pytest.mark.parametrize("name", ["football_5v5", "give_way", "all_wheel"])
def test_compact_rollout_and_step_n_forms(name):
    """vmas_world_rollout (several steps in one launch, the tile resident in LDS) and vmas_world_step_n over two queues
    (sub-ranges of the batch) on the compacted kernel == its single steps."""
    g = load(name)
    B, n = 64 * 9 + 17, 5
    st0, ft0, _, _ = make_batch(g, B, seed=12)
    rng = np.random.default_rng(1)
    outs = []
    forces = None
    for mode in ("steps", "rollout", "step_n_2q"):
        hw = _hip(g.spec, B)
        hw.set_compact(1)
        _up(hw, st0, ft0)
        if forces is None:
            forces = torch.zeros(n, *hw.agent_ft.shape, device="cuda")
            if ft0.shape[0]:
                for k in range(n):
                    forces[k, : ft0.shape[0], :, :B] = torch.from_numpy((ft0 * (1 + 0.2 * rng.normal(0, 1, ft0.shape))).astype(np.float32))
        f = forces.clone()
        if mode == "steps":
            for k in range(n):
                hw.agent_ft.copy_(f[k])
                hw.step()
                f[k].copy_(hw.agent_ft)
        elif mode == "rollout":
            hw.rollout(n, f)
        else:
            hw.set_queues(2)
            hw.step_n(n, f)
        torch.cuda.synchronize()
        outs.append((hw.state.clone(), f))
    for other, what in ((outs[1], "rollout"), (outs[2], "step_n over two queues")):
        assert torch.equal(_bits(outs[0][0]), _bits(other[0])), f"{name}: {what} state"
        assert torch.equal(_bits(outs[0][1][:, :, :, :B]), _bits(other[1][:, :, :, :B])), f"{name}: {what} clamped forces"


def test_football_default_is_the_compact_kernel():
    g = load("football_5v5")
    assert _hip(g.spec, 4096).compact
    g = load("navigation_n8")
    assert not _hip(g.spec, 4096).compact  # (28 pairs: its world-specialised kernel stays)
    g = load("balance_n4")
    hw = _hip(g.spec, 4096)
    hw.set_compact(1)
    assert not hw.compact  # boxes and line-line pairs: the world does not qualify


def test_the_library_leaves_the_compacted_kernel_when_its_contact_lists_overflow():
    """vmas_world_set_compact(-1), the default: the compacted kernel counts its contacts and the host sends a world whose
    tiles are dense with them to the interpreter, which is faster there (and probes again later).  Football driven into the
    walls by one held action is such a world.  The choice depends on the states alone: two such worlds stay bitwise identical;
    and whatever kernel made the last state, one more step of it is the oracle's within the north-star tolerance."""
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.environment import make_env

    kw = dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)
    B = 4096
    envs = [make_env("football", num_envs=B, device="cuda:0", seed=2, validate_actions=False, **kw) for _ in range(2)]
    acts = [envs[0].get_random_action(a) for a in envs[0].agents]
    for env in envs:
        for _ in range(20):
            env.step([a.clone() for a in acts])
    a, b = (e.world._get_backend() for e in envs)
    assert a.compact and b.compact
    for chunk in range(12):  # the last action held: 12 x 100 steps
        a.step_n(100)
        for _ in range(100):  # (the same launches one by one: the bookkeeping counts launches, not calls)
            b.step()
        sa, sb = envs[0].world._state, envs[1].world._state
        assert torch.equal(sa.view(torch.int32), sb.view(torch.int32)), f"two runs differ after {100 * (chunk + 1)} held steps"
    st = a.compact_stats()
    print("compact stats after 1200 held steps:", st)
    assert st["tiles"] > 0 and st == b.compact_stats()
    assert st["switches"] >= 1, f"a held action packs the bodies against the walls: the compacted kernel should have been left ({st})"
    # one more step against the oracle, every environment
    spec = envs[0].world.spec
    nE, nA = spec.n_entities, spec.n_agents
    s0 = envs[0].world._state[:nE, :, :B].cpu().numpy().copy()
    f0 = envs[0].world._agent_ft[:nA, :, :B].cpu().numpy().copy()
    Oracle(spec).step(s0, f0, threads=16)
    a.step()
    got = envs[0].world._state[:nE, :, :B].cpu().numpy()
    err = np.abs(got - s0)
    assert (err <= 1e-5 + 1e-5 * np.abs(s0)).all(), f"max err {err.max():.3g}"


def _contacts_per_entity(spec, state, B):
    """[E, B] how many of an entity's pairs are in contact (within the distance below which the penalty force is non-zero,
    core.py:2836) - sphere-sphere and line-sphere pairs, the worlds this kernel serves."""
    E = spec.entities
    pos = state[:, 0:2, :B]
    rot = state[:, 4, :B]
    n = np.zeros((len(E), B), np.int32)
    LMD = 4.0 / 600.0
    for p in spec.pairs:
        a, b = p.a, p.b
        if p.type == 0:
            d = np.hypot(pos[a, 0] - pos[b, 0], pos[a, 1] - pos[b, 1]) - (E[a].radius + E[b].radius)
        else:  # a = line, b = sphere: distance of the centre to the segment
            c, s_ = np.cos(rot[a]), np.sin(rot[a])
            dx, dy = pos[b, 0] - pos[a, 0], pos[b, 1] - pos[a, 1]
            t = np.clip(dx * c + dy * s_, -E[a].length / 2, E[a].length / 2)
            d = np.hypot(dx - t * c, dy - t * s_) - (E[b].radius + LMD)
        hit = d < 0
        n[a] += hit
        n[b] += hit
    return n


@pytest.mark.parametrize("mode", [1, 0], ids=["compacted", "interpreter"])
def test_each_football_kernel_holds_the_contract_against_the_oracle_in_dense_contact(mode):
    """The two football kernels are not bitwise equal to each other where an entity has three or more simultaneous contacts
    (include/vmas_hip.h, vmas_world_set_compact) - so EACH is pinned against the oracle there: 4096 environments driven
    into the walls by one held action for 600 steps (bodies stacked against walls and each other), then one step of the
    pinned kernel against the oracle's step of the same state, every environment, |err| <= 1e-5 (abs + rel) with NO
    sensitivity allowance."""
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.environment import make_env

    kw = dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)
    B = 4096
    env = make_env("football", num_envs=B, device="cuda:0", seed=4, validate_actions=False, **kw)
    acts = [env.get_random_action(a) for a in env.agents]
    for _ in range(20):
        env.step([a.clone() for a in acts])
    be = env.world._get_backend()
    be.set_compact(0)  # (the approach to the dense state on ONE kernel, whichever is under test afterwards)
    be.step_n(600)     # the last action held
    spec = env.world.spec
    nE, nA = spec.n_entities, spec.n_agents
    worst = 0.0
    for rep in range(3):  # three consecutive steps, each teacher-forced from the device state
        s0 = env.world._state[:nE, :, :B].cpu().numpy().copy()
        f0 = env.world._agent_ft[:nA, :, :B].cpu().numpy().copy()
        if rep == 0:
            cnt = _contacts_per_entity(spec, s0, B)
            dense_envs = int((cnt.max(axis=0) >= 3).sum())
            print(f"dense-contact state: {dense_envs} of {B} environments have an entity with >= 3 contacts, max {int(cnt.max())}")
            assert dense_envs >= B // 20, f"only {dense_envs} environments reached three simultaneous contacts on one entity"
        want = s0.copy()
        Oracle(spec).step(want, f0.copy(), threads=16)
        be.set_compact(mode)
        assert be.compact == bool(mode)
        be.step()
        got = env.world._state[:nE, :, :B].cpu().numpy()
        assert np.isfinite(got).all()
        err = np.abs(got - want)
        worst = max(worst, float(err.max()))
        bad = err > 1e-5 + 1e-5 * np.abs(want)
        assert not bad.any(), f"step {rep}: {int(bad.sum())} values beyond 1e-5 (max err {err.max():.3g})"
    print(f"max |{'compacted' if mode else 'interpreter'} - oracle| over three dense-contact steps: {worst:.3g}")
