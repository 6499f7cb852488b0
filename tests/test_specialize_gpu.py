"""Run-time specialisation on the GPU: the kernels compiled for a world (specialize.py -> vmas_world_load_spec) against
the schedule interpreter, bit for bit - worlds the library has NO built-in specialisation for: balance n_agents=3 (BASELINE
config 1's world), transport with two packages (box-box), jointed worlds (joint_passage, ball_trajectory), friction / force
ranges (give_way), a rotating line (wheel), and one attached reference scenario.  (Schedules of more than 40 item records -
football, waterfall - are refused: unrolled they outgrow the instruction cache.)  Code objects come from the on-disk cache when
__graft_entry__.build() (or an earlier test) made them, else hipcc compiles them here (~10 s each)."""
import os

import numpy as np
import pytest
import torch

from golden_util import FIXTURES, load
from test_hip_parity import _hip, _up, make_batch

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int32)


_BATCH = {"balance_n3": 4096, "transport_2pkg": 1024, "all_joint_passage_size": 700, "ball_trajectory": 1000, "give_way": 4096,
          "all_wheel": 64 * 7 + 3}  # (scripts/prebuild_test_specs.py compiles the same geometries ahead of time)


@pytest.mark.parametrize("name", FIXTURES)
def test_runtime_specialisation_is_bitwise_the_interpreter(name):
    """EVERY golden fixture: the world's own kernels against the schedule interpreter.  Skipped - with the reason - where
    the library has a built-in specialisation at this geometry, where the launch is not a plain one (per-environment
    joint rotations / gravity) and where the size guard refuses the schedule (football, waterfall, ...)."""
    from vectorizedmultiagentsimulator_amd.specialize import SpecializeError

    B = _BATCH.get(name, 647)
    g = load(name)
    st0, ft0, jfr_np, eg_np = make_batch(g, B, seed=17)
    if jfr_np is not None or eg_np is not None:
        pytest.skip("per-environment inputs run the interpreter (PLAIN launches only are specialised)")
    a, b = _hip(g.spec, B), _hip(g.spec, B)
    if a.specialized:
        pytest.skip("a built-in specialisation serves this geometry (tests/test_hip_parity.py pins it)")
    try:
        ok = a.specialize()
    except SpecializeError as e:
        pytest.skip(f"refused by the size guard: {e}")
    if not ok:
        pytest.skip("refused by the size guard (more item records than a specialisation may unroll, or an item list outside LDS)")
    assert a.specialized and not b.specialized
    for hw in (a, b):
        _up(hw, st0, ft0)
    rng = np.random.default_rng(5)
    for t in range(6):
        f = torch.from_numpy((ft0 * (1 + 0.3 * rng.normal(0, 1, ft0.shape))).astype(np.float32))
        for hw in (a, b):
            if ft0.shape[0]:
                hw.agent_ft[: ft0.shape[0], :, :B].copy_(f)
            hw.step()
        assert torch.equal(_bits(a.state), _bits(b.state)), f"{name}: state differs at step {t}"
        assert torch.equal(_bits(a.agent_ft), _bits(b.agent_ft)), f"{name}: clamped forces differ at step {t}"
    # several steps per launch (the multi form) and a switch back to the interpreter
    forces = torch.zeros(4, *a.agent_ft.shape, device="cuda")
    for hw in (a, b):
        hw.rollout(4, forces.clone())
    assert torch.equal(_bits(a.state), _bits(b.state)), f"{name}: rollout"
    a.set_specialized(False)
    assert not a.specialized
    for hw in (a, b):
        hw.step()
    assert torch.equal(_bits(a.state), _bits(b.state))


def test_a_code_object_of_another_world_is_refused():
    from vectorizedmultiagentsimulator_amd import _abi as A
    from vectorizedmultiagentsimulator_amd import specialize as S
    from vectorizedmultiagentsimulator_amd.backend import VmasHipError

    g3, g2 = load("balance_n3"), load("transport_2pkg")
    a, b = _hip(g3.spec, 4096), _hip(g2.spec, 4096)
    meta, words = S.schedule(a._h)
    path = S.code_object(S.render(meta, words, int(g3.spec.substeps), 0))
    lib = A.load_library()
    assert lib.vmas_world_load_spec(b._h, path.encode()) != 0 and "another schedule" in A.last_error()
    assert not b.specialized
    assert lib.vmas_world_load_spec(a._h, path.encode()) == 0 and a.specialized


def test_make_env_specialize_runs_the_one_launch_step_on_its_own_kernel():
    """make_env(..., specialize=True): balance n_agents=3 - ingest prologue + physics + balance epilogue as ONE launch of the
    world's own kernel - against the same environment on the interpreter: bitwise."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    B = 4096
    a = make_env("balance", num_envs=B, device="cuda:0", seed=3, n_agents=3, validate_actions=False, specialize=True)
    b = make_env("balance", num_envs=B, device="cuda:0", seed=3, n_agents=3, validate_actions=False, specialize=False)
    assert a.world._get_backend().specialized and not b.world._get_backend().specialized and a._one_launch
    assert torch.equal(a.world._state, b.world._state)
    for t in range(10):
        acts = [a.get_random_action(ag) for ag in a.agents]
        o1, r1, d1, _ = a.step([u.clone() for u in acts])
        o2, r2, d2, _ = b.step(acts)
        assert torch.equal(_bits(a.world._state), _bits(b.world._state)), f"t={t}"
        for x, y in zip(o1 + r1, o2 + r2):
            assert torch.equal(_bits(x), _bits(y)), f"t={t}"
        assert torch.equal(d1, d2)


def test_attach_specialize_on_a_reference_scenario():
    """attach(env, specialize=True): a scenario only the reference has (`dropout`) on its own compiled kernel, side by side
    with the same attached environment on the interpreter."""
    from oracle import ref
    from vectorizedmultiagentsimulator_amd.adapter import attach

    if not ref.available():
        pytest.skip("neither /root/reference nor oracle/_ref present")
    envs = [ref.make_env("dropout", num_envs=2048, device="cuda:0", seed=0) for _ in range(2)]
    for e, f in zip(envs[0].world.entities, envs[1].world.entities):
        f.set_pos(e.state.pos, batch_index=None)
    ha, hb = attach(envs[0], specialize=True), attach(envs[1])
    assert ha.backend.specialized and not hb.backend.specialized
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for t in range(8):
        acts = [(torch.rand(2048, a.action_size, device="cuda:0", generator=g) * 2 - 1) for a in envs[0].agents]
        envs[0].step([u.clone() for u in acts])
        envs[1].step(acts)
        assert torch.equal(_bits(ha.state), _bits(hb.state)), f"t={t}"
    ha.detach()
    hb.detach()


def test_large_schedules_are_not_specialised():
    from vectorizedmultiagentsimulator_amd import specialize as S

    g = load("waterfall")
    hw = _hip(g.spec, 512)
    assert hw.specialize() is False and not hw.specialized
    with pytest.raises(S.SpecializeError, match="too large"):
        S.specialize(hw, strict=True)


def test_navigation_between_the_built_in_geometries_specialises_at_run_time():
    """navigation n_agents=8 has built-in specialisations for 4 waves per tile (65 536 environments) and 16 (up to 16 384);
    in between the planner chooses 8 - served by the run-time specialiser, one-launch step (ingest + physics + LIDAR /
    observation / reward epilogue with > 64 KB of LDS) included: bitwise the interpreter."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    B = 24576
    a = make_env("navigation", num_envs=B, device="cuda:0", seed=5, n_agents=8, validate_actions=False, specialize=True)
    b = make_env("navigation", num_envs=B, device="cuda:0", seed=5, n_agents=8, validate_actions=False, specialize=False)
    assert a.world._get_backend().lanes_per_env == 8
    assert a.world._get_backend().specialized and not b.world._get_backend().specialized and a._one_launch
    for t in range(8):
        acts = [a.get_random_action(ag) for ag in a.agents]
        oa, ra, da, _ = a.step([x.clone() for x in acts])
        ob, rb, db, _ = b.step(acts)
        assert torch.equal(_bits(a.world._state), _bits(b.world._state)), f"state differs at step {t}"
        assert torch.equal(_bits(torch.stack(oa)), _bits(torch.stack(ob))), f"observations differ at step {t}"
        assert torch.equal(_bits(torch.stack(ra)), _bits(torch.stack(rb))) and torch.equal(da, db), f"rewards / done differ at step {t}"


def test_make_env_takes_a_cached_specialisation_and_never_compiles_by_default(tmp_path, monkeypatch):
    """make_env(specialize=None), the default: a world whose code object is in the cache runs its own kernels; one whose is
    not keeps the interpreter and NOTHING is compiled (an empty cache directory stays empty)."""
    from vectorizedmultiagentsimulator_amd import specialize as S
    from vectorizedmultiagentsimulator_amd.environment import make_env

    hit = make_env("balance", num_envs=4096, device="cuda:0", seed=3, n_agents=3, validate_actions=False)  # (PREBUILD)
    assert hit.world._get_backend().specialized, "balance n_agents=3 at 4096 environments is pre-compiled by build()"
    monkeypatch.setattr(S, "CACHE", str(tmp_path))
    miss = make_env("balance", num_envs=4096, device="cuda:0", seed=3, n_agents=3, validate_actions=False)
    assert not miss.world._get_backend().specialized and not any(tmp_path.iterdir())


def test_a_corrupt_cache_entry_never_breaks_make_env(tmp_path, monkeypatch):
    """make_env(specialize=None) is an optional fast path: a cache entry the library refuses (truncated by a crash, foreign
    bytes under the name) warns, is dropped and leaves the world on the interpreter - same results; with specialize=True the
    same entry is replaced by a fresh compile on the next request."""
    import warnings

    from vectorizedmultiagentsimulator_amd import specialize as S
    from vectorizedmultiagentsimulator_amd.environment import make_env

    monkeypatch.setattr(S, "CACHE", str(tmp_path))
    probe = make_env("balance", num_envs=4096, device="cuda:0", seed=3, n_agents=3, validate_actions=False, specialize=False)
    be = probe.world._get_backend()
    meta, words = S.schedule(be._h)
    src = S.render(meta, words, int(be.spec.substeps), 1)
    bad = S.cache_path(src, str(tmp_path))
    with open(bad, "wb") as f:
        f.write(b"not a code object" * 100)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        env = make_env("balance", num_envs=4096, device="cuda:0", seed=3, n_agents=3, validate_actions=False)
    assert not env.world._get_backend().specialized
    assert any("world-specialised kernel not used" in str(w.message) for w in caught), [str(w.message) for w in caught]
    assert not os.path.exists(bad), "the refused entry should have been dropped"
    acts = [env.get_random_action(a) for a in env.agents]
    ra, rb = env.step(acts), probe.step([a.clone() for a in acts])
    assert all(torch.equal(x, y) for x, y in zip(ra[0], rb[0]))
