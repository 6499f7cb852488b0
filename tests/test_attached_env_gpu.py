"""attach(vmas.make_env(..., device="cuda")) with the fused one-launch ``Environment.step`` (attached_env.py) beside an
UNTOUCHED reference environment on the CPU - SURVEY.md 8b's boundary through the reference's own objects (VERDICT r4
row g).  Teacher-forced: before every step the attached environment is given the CPU reference's state (entity state
through the reference's setters, the scenario's shaping terms in place), both take the same actions, and what
``env.step`` returns must agree - observations at 1e-5 abs+rel (north_star), rewards at that bound scaled by the
magnitude of the shaping terms they are differences of (tests/test_env_fused_gpu.py), dones and flags exactly.  The
configurations are those of the nine ``envstep_*`` fixtures.  The reference: oracle/_ref on the GPU box."""
import ast
import os

import numpy as np
import pytest
import torch

from golden_util import ENVSTEP_FIXTURES, GOLDEN_DIR

pytestmark = [pytest.mark.gpu, pytest.mark.reference]

ATOL, RTOL = 1e-5, 1e-5
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vmas():
    from oracle import ref

    ref.import_vmas()
    return ref


def _cfg(fixture):
    G = np.load(os.path.join(GOLDEN_DIR, fixture + ".npz"))
    ms = int(G["max_steps"])
    return fixture.split("_")[1], ast.literal_eval(str(G["kwargs"])), (None if ms < 0 else ms)


SHAPING_SCALE = {"balance": 10.0, "transport": 10.0, "navigation": 1.0, "football": 2.0}


def _terms(env, scenario):
    """The scenario-side persistent tensors a step reads: (object, name) pairs."""
    sc = env.scenario
    if scenario == "balance":
        return [(sc, "global_shaping")]
    if scenario == "transport":
        return [(p, n) for p in sc.packages for n in ("global_shaping", "on_goal")]
    if scenario == "navigation":
        return [(a, "pos_shaping") for a in env.world.agents]
    return [(sc.ball, n) for n in ("pos_shaping_blue", "pos_shaping_red", "pos_shaping_agent_blue", "pos_shaping_agent_red")]


def _force_state(ref, att, scenario):
    """Teacher forcing: the CPU reference's state and shaping terms into the attached environment (in place)."""
    for ea, eb in zip(ref.world.entities, att.world.entities):
        eb.set_pos(ea.state.pos.to(DEV), batch_index=None)
        eb.set_vel(ea.state.vel.to(DEV), batch_index=None)
        eb.set_rot(ea.state.rot.to(DEV), batch_index=None)
        eb.set_ang_vel(ea.state.ang_vel.to(DEV), batch_index=None)
    for aa, ab in zip(ref.world.agents, att.world.agents):  # (football observes the previous step's agent forces)
        if aa.state.force is not None:
            ab.state.force = aa.state.force.to(DEV)
        if aa.state.torque is not None:
            ab.state.torque = aa.state.torque.to(DEV)
    for (oa, n), (ob, _) in zip(_terms(ref, scenario), _terms(att, scenario)):
        getattr(ob, n).copy_(getattr(oa, n).to(DEV))
    att.steps.copy_(ref.steps.to(DEV))


def _force_state_same_device(src, dst, scenario):
    """``dst`` := ``src`` (two environments on the GPU): entity state, agent forces, shaping terms, step counter."""
    for ea, eb in zip(src.world.entities, dst.world.entities):
        eb.set_pos(ea.state.pos.clone(), batch_index=None)
        eb.set_vel(ea.state.vel.clone(), batch_index=None)
        eb.set_rot(ea.state.rot.clone(), batch_index=None)
        eb.set_ang_vel(ea.state.ang_vel.clone(), batch_index=None)
    for aa, ab in zip(src.world.agents, dst.world.agents):
        if aa.state.force is not None:
            ab.state.force = aa.state.force.clone()
        if aa.state.torque is not None:
            ab.state.torque = aa.state.torque.clone()
    for (oa, n), (ob, _) in zip(_terms(src, scenario), _terms(dst, scenario)):
        getattr(ob, n).copy_(getattr(oa, n))
    dst.steps.copy_(src.steps)


def _close(got, want, what, scale=1.0):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    bad = err > ATOL * scale + RTOL * np.abs(want)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} off, max err {err.max():.3g} at {np.argwhere(bad)[:4].tolist()}"
    return float(err.max()) if err.size else 0.0


def _actions(env, g):
    if env.continuous_actions:
        return [(torch.rand(env.num_envs, a.action_size, generator=g) * 2 - 1) * a.action.u_range_tensor.cpu() for a in env.agents]
    return [torch.randint(0, int(np.prod(a.discrete_action_nvec)), (env.num_envs, 1), generator=g) for a in env.agents]


def _compare_step(out_ref, out_att, scenario, what):
    o1, r1, d1, i1 = out_ref
    o2, r2, d2, i2 = out_att
    scale = SHAPING_SCALE[scenario]
    assert isinstance(out_att, list) and isinstance(o2, list) and isinstance(r2, list) and isinstance(i2, list)
    for k, (a, b) in enumerate(zip(o1, o2)):
        _close(b.cpu().numpy(), a.numpy(), f"{what} obs[{k}]")
    for k, (a, b) in enumerate(zip(r1, r2)):
        _close(b.cpu().numpy(), a.numpy(), f"{what} rew[{k}]", scale=scale)
    assert torch.equal(d1, d2.cpu()), f"{what} done: {(d1 != d2.cpu()).sum()} differ"
    for k, (ia, ib) in enumerate(zip(i1, i2)):
        assert set(ia) == set(ib), f"{what} info[{k}] keys {sorted(ia)} vs {sorted(ib)}"
        for name in ia:
            a, b = ia[name], ib[name].cpu()
            if a.dtype == torch.bool:
                assert torch.equal(a, b), f"{what} info[{k}][{name}]"
            else:
                _close(b.numpy(), a.numpy(), f"{what} info[{k}][{name}]", scale=scale)


@pytest.mark.parametrize("fixture", ENVSTEP_FIXTURES)
def test_attached_fused_env_step_equals_the_reference_teacher_forced(vmas, fixture):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    scenario, kw, max_steps = _cfg(fixture)
    B, T = 70, 40
    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=0, max_steps=max_steps, **kw)
    att = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, max_steps=max_steps, **kw)
    h = attach(att, fused=True)  # (True: a fallback to the reference's tensor-op step would fail the test, not pass it)
    assert h.fused is not None and att.__dict__["step"] == h.fused.step and h.exact_broad_phase
    g = torch.Generator().manual_seed(11)
    n_done = 0
    with torch.no_grad():
        for t in range(T):
            _force_state(ref, att, scenario)
            acts = _actions(ref, g)
            out_ref = ref.step([a.clone() for a in acts])
            out_att = att.step([a.to(DEV) for a in acts])
            _compare_step(out_ref, out_att, scenario, f"{fixture} t={t}")
            assert torch.equal(att.steps.cpu(), ref.steps)
            # the world the step left behind (what the NEXT reference-side call - reset_at, render, a query - would see)
            for ea, eb in zip(ref.world.entities, att.world.entities):
                _close(eb.state.pos.cpu().numpy(), ea.state.pos.numpy(), f"{fixture} t={t} {ea.name}.pos")
                _close(eb.state.vel.cpu().numpy(), ea.state.vel.numpy(), f"{fixture} t={t} {ea.name}.vel")
            n_done += int(out_ref[2].sum())
            if t == 12:  # a partial reset through the reference's own reset_at: observations by its tensor ops on the views
                ref.reset_at(3)
                oa = att.reset_at(3)
                assert len(oa) == len(att.agents) and oa[0].shape == out_att[0][0].shape  # (the reference returns every environment's)
            if t == 25:
                ref.reset()
                ob = att.reset()
                assert len(ob) == len(att.agents) and ob[0].shape == out_att[0][0].shape
    h.detach()
    assert "step" not in att.__dict__
    att.step([a.to(DEV) for a in _actions(att, g)])  # the reference's own Environment.step works again


@pytest.mark.parametrize("scenario,kw", [("balance", dict(n_agents=4)), ("navigation", dict(n_agents=4)),
                                         ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False))])
def test_attached_fused_env_free_running_tracks_the_reference(vmas, scenario, kw):
    """No teacher forcing: same start, same actions, 30 steps - the drift stays that of two fp32 implementations of a
    chaotic system (the plumbing bound of tests/test_adapter_reference.py), rewards and dones follow."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 130
    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=0, **kw)
    att = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, **kw)
    h = attach(att, fused=True)
    _force_state(ref, att, scenario)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for t in range(30):
            acts = _actions(ref, g)
            o1, r1, d1, _ = ref.step([a.clone() for a in acts])
            o2, r2, d2, _ = att.step([a.to(DEV) for a in acts])
            for a, b in zip(o1, o2):
                assert torch.allclose(a, b.cpu(), atol=5e-3, rtol=1e-2), f"{scenario} obs diverged at step {t}: {(a - b.cpu()).abs().max()}"
            assert (d1 == d2.cpu()).float().mean() > 0.98
    h.detach()


def test_attached_fused_dict_spaces_truncation_discrete_and_clamp(vmas):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 40
    # dict spaces + terminated / truncated split
    kw = dict(n_agents=3)
    ref = vmas.make_env("balance", num_envs=B, device="cpu", seed=0, max_steps=4, dict_spaces=True, terminated_truncated=True, **kw)
    att = vmas.make_env("balance", num_envs=B, device=DEV, seed=0, max_steps=4, dict_spaces=True, terminated_truncated=True, **kw)
    h = attach(att, fused=True)
    g = torch.Generator().manual_seed(2)
    for t in range(6):
        _force_state(ref, att, "balance")
        acts = {a.name: u for a, u in zip(ref.agents, _actions(ref, g))}
        o1, r1, te1, tr1, i1 = ref.step({k: v.clone() for k, v in acts.items()})
        o2, r2, te2, tr2, i2 = att.step({k: v.to(DEV) for k, v in acts.items()})
        assert list(o2) == list(o1) and list(r2) == list(r1) and list(i2) == list(i1)
        for k in o1:
            _close(o2[k].cpu().numpy(), o1[k].numpy(), f"dict obs {k} t={t}")
            _close(r2[k].cpu().numpy(), r1[k].numpy(), f"dict rew {k} t={t}", scale=10.0)
        assert torch.equal(te1, te2.cpu()) and torch.equal(tr1, tr2.cpu()), t
        assert bool(tr1.all()) == (t >= 3)
    h.detach()
    # discrete actions, and clamped continuous ones out of range
    for env_kw, bad_scale in ((dict(continuous_actions=False), None), (dict(clamp_actions=True), 3.0)):
        ref = vmas.make_env("transport", num_envs=B, device="cpu", seed=0, **env_kw)
        att = vmas.make_env("transport", num_envs=B, device=DEV, seed=0, **env_kw)
        h = attach(att, fused=True)
        for t in range(5):
            _force_state(ref, att, "transport")
            acts = _actions(ref, g)
            if bad_scale:
                acts = [a * bad_scale for a in acts]
            out_ref = ref.step([a.clone() for a in acts])
            out_att = att.step([a.to(DEV) for a in acts])
            _compare_step(out_ref, out_att, "transport", f"transport {env_kw} t={t}")
        h.detach()


def test_attached_fused_validates_like_the_reference(vmas):
    """environment.py:621,651-653: NaN / out-of-range actions raise BEFORE the world is touched; validate_actions=False
    drops the check (and its host sync)."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    att = vmas.make_env("balance", num_envs=33, device=DEV, seed=0, n_agents=3)
    h = attach(att, fused=True)
    good = [att.get_random_action(a) for a in att.agents]
    att.step(good)
    before = h.state.clone()
    steps = att.steps.clone()
    bad = [a.clone() for a in good]
    bad[1][5, 0] = float("nan")
    with pytest.raises(AssertionError):
        att.step(bad)
    bad = [a.clone() for a in good]
    bad[2][7, 1] = 1.5
    with pytest.raises(AssertionError):
        att.step(bad)
    with pytest.raises(AssertionError):
        att.step(good[:2])
    assert torch.equal(h.state, before) and torch.equal(att.steps, steps), "a refused action must not touch the world"
    att.step(good)  # and the environment goes on
    h.detach()
    h = attach(att, fused=True, validate_actions=False)
    att.step(good)
    h.detach()
    # deferred: the step launch flags the bad action itself, the NEXT step (or check_actions) raises - no synchronisation
    h = attach(att, fused=True, validate_actions="deferred")
    att.step(good)
    att.step(good)
    h.fused.check_actions()  # nothing to report
    bad = [a.clone() for a in good]
    bad[0][3, 1] = float("nan")
    att.step(bad)  # does not raise: the launch is asynchronous
    torch.cuda.synchronize()
    with pytest.raises(AssertionError, match="NaN"):
        att.step(good)
    att.reset()  # (the NaN went through one step of environment 3)
    att.step(good)
    bad = [a.clone() for a in good]
    bad[1][0, 0] = -1.25
    att.step(bad)
    with pytest.raises(AssertionError, match="out of its range"):
        h.fused.check_actions()
    h.fused.check_actions()  # the flag was consumed
    h.detach()


def test_attached_fused_outputs_are_fresh_and_scenario_attributes_follow(vmas):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    att = vmas.make_env("balance", num_envs=130, device=DEV, seed=0, n_agents=4)
    h = attach(att, fused=True)
    acts = [att.get_random_action(a) for a in att.agents]
    o1, r1, d1, i1 = att.step(acts)
    keep = [x.clone() for x in o1]
    o2, r2, _, _ = att.step(acts)
    assert all(torch.equal(a, b) for a, b in zip(o1, keep)), "a later step must not overwrite returned observations"
    assert o1[0].data_ptr() != o2[0].data_ptr()
    sc = att.scenario
    assert torch.equal(sc.ground_rew + sc.pos_rew, r2[0])  # the scenario's own attributes are the step's
    # ... and its tensor-op methods see the same world: get_from_scenario by the reference's own code agrees with the kernel
    obs = att.get_from_scenario(get_observations=True, get_rewards=False, get_infos=False, get_dones=True)
    for a, b in zip(obs[0], o2):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-6)
    # a static change mid-episode rebuilds the native world AND the fused layer on it
    att.world.agents[0].mass = 2.5
    att.step(acts)
    assert h.refreshes == 1 and h.fused._backend is h.backend
    h.detach()


def test_attach_falls_back_with_a_reason_where_no_kernel_covers_the_scenario(vmas):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    att = vmas.make_env("waterfall", num_envs=16, device=DEV, seed=0)
    h = attach(att)
    assert h.fused is None and "no fused post-step kernel" in h.fused_reason and "step" not in att.__dict__
    att.step([att.get_random_action(a) for a in att.agents])
    h.detach()
    with pytest.raises(NotImplementedError):
        attach(att, fused=True)
    att = vmas.make_env("football", num_envs=16, device=DEV, seed=0, n_blue_agents=2, n_red_agents=2)  # AgentPolicy opponents
    h = attach(att)
    assert h.fused is None and "ai_red_agents" in h.fused_reason
    att.step([att.get_random_action(a) for a in att.agents])
    h.detach()


FOOTBALL_5V5 = dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)


@pytest.mark.parametrize("scenario,kw,B,steps", [
    ("balance", dict(n_agents=4), 32768, 20),       # BASELINE config 2
    ("transport", {}, 16384, 20),                   # config 3
    ("transport", dict(n_packages=2), 16384, 20),   # config 3 with box-box pairs
    ("navigation", dict(n_agents=8), 8192, 20),     # config 4's per-GPU shard
    ("football", FOOTBALL_5V5, 16384, 20),          # config 5's per-GPU shard
    ("football", FOOTBALL_5V5, 131072, 1),          # config 5 on one GPU
], ids=["balance-32768", "transport-16384", "transport2-16384", "navigation-8192", "football-16384", "football-131072"])
def test_attached_fused_at_benchmark_size_against_the_reference(vmas, scenario, kw, B, steps):
    """BASELINE sizes through the reference's own objects: `steps` teacher-forced ``env.step`` calls against the CPU reference of
    the same batch, strict 1e-5 on the observations.  The reference's batch-global broad phase (World.collides, core.py:2788-
    2803) is what the attached step follows at every size (the lazy form inside the launch); until round 5 these sizes ran every
    pair per environment, which the round-5 review showed leaves the reference's trajectory on configs 3 and 5 (an environment
    in a pair's band while no environment of the batch overlaps: 0.1-0.3 N the reference does not apply)."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=0, **kw)
    att = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, **kw)
    h = attach(att, fused=True, validate_actions=False)
    assert h.exact_broad_phase and h.backend.exact_form() in (0, 1), "the reference's rule, inside the step launch"
    assert h.fused.ingest_in_step
    if B <= 16384 or scenario != "football":
        assert h.fused.one_launch
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for t in range(steps):
            _force_state(ref, att, scenario)
            acts = _actions(ref, g)
            out_ref = ref.step([a.clone() for a in acts])
            out_att = att.step([a.to(DEV) for a in acts])
            _compare_step(out_ref, out_att, scenario, f"{scenario} {B} t={t}")
    assert h.backend.exact_status() == 0
    h.detach()


# configurations beyond the fixtures': every scenario kwarg the reference accepts that changes what the ingest or the epilogue
# computes, a few at a time (teacher-forced like the fixture test, 10 steps each)
WIDER = [
    ("balance", dict(n_agents=2, package_mass=1.5), {}),
    ("balance", dict(n_agents=6, random_package_pos_on_line=False), dict(max_steps=5)),
    ("balance", dict(n_agents=3), dict(continuous_actions=False)),
    ("transport", dict(n_agents=2, n_packages=3, package_width=0.2, package_length=0.1, package_mass=10), {}),
    ("transport", dict(n_agents=6), dict(clamp_actions=True, max_steps=7)),
    ("navigation", dict(n_agents=2, lidar_range=0.5, n_lidar_rays=7), {}),
    ("navigation", dict(n_agents=6, shared_rew=False, agent_collision_penalty=-0.3, final_reward=0.5, pos_shaping_factor=2.0), dict(max_steps=6)),
    ("navigation", dict(n_agents=4, collisions=False, agents_with_same_goal=4), {}),
    ("navigation", dict(n_agents=4, collisions=False, agents_with_same_goal=2, split_goals=True, observe_all_goals=True), {}),
    ("navigation", dict(n_agents=3, enforce_bounds=True, world_spawning_x=0.6, world_spawning_y=0.6, agent_radius=0.05), dict(continuous_actions=False)),
    ("football", dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False, agent_size=0.03, ball_size=0.025, goal_size=0.5, pitch_length=2.4,
                      pitch_width=1.2, dense_reward=True), {}),  # (1 v 1 is refused by the reference itself: torch.cat of no teammates)
    ("football", dict(n_blue_agents=4, n_red_agents=4, ai_red_agents=False, observe_teammates=False), dict(max_steps=8)),
    ("football", dict(n_blue_agents=2, n_red_agents=3, ai_red_agents=False, observe_teammates=False, observe_adversaries=False), {}),
    ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False, spawn_in_formation=True, u_multiplier=0.2, max_speed=0.3,
                      ball_mass=0.5, scoring_reward=10.0, pos_shaping_factor_ball_goal=5.0, distance_to_ball_trigger=0.2), dict(continuous_actions=False)),
]


@pytest.mark.parametrize("scenario,kw,env_kw", WIDER)
def test_attached_fused_wider_configurations_teacher_forced(vmas, scenario, kw, env_kw):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 40
    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=1, **env_kw, **kw)
    att = vmas.make_env(scenario, num_envs=B, device=DEV, seed=1, **env_kw, **kw)
    h = attach(att)
    assert h.fused is not None, f"{scenario} {kw}: expected the one-launch step, got the fallback ({h.fused_reason})"
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for t in range(10):
            _force_state(ref, att, scenario)
            acts = _actions(ref, g)
            out_ref = ref.step([a.clone() for a in acts])
            out_att = att.step([a.to(DEV) for a in acts])
            _compare_step(out_ref, out_att, scenario, f"{scenario} {kw} {env_kw} t={t}")
            assert torch.equal(att.steps.cpu(), ref.steps)
    h.detach()


@pytest.mark.parametrize("scenario,kw", [("balance", dict(n_agents=4)), ("transport", dict(n_packages=2)),
                                         ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False))])
def test_attached_fused_rollout_is_bitwise_k_single_steps(vmas, scenario, kw):
    """``handle.fused.rollout`` (K steps of the reference's environment in ONE launch) against K ``env.step`` calls of a twin."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B, K = 200, 7
    a = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, max_steps=5, **kw)
    b = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, max_steps=5, **kw)
    ha, hb = attach(a, fused=True, validate_actions=False), attach(b, fused=True, validate_actions=False)
    _force_state_same_device(a, b, scenario)
    g = torch.Generator(device=DEV).manual_seed(3)
    acts = [(torch.rand(K, B, ag.action_size, device=DEV, generator=g) * 2 - 1) * 0.9 for ag in a.agents]
    want = [a.step([u[k] for u in acts]) for k in range(K)]
    got = hb.fused.rollout(acts)
    for k in range(K):
        obs, rew, done, info = want[k]
        for i in range(len(a.agents)):
            assert torch.equal(got["obs"][k, i], obs[i]), f"{scenario}: obs of agent {i} differs at step {k}"
            assert torch.equal(got["rew"][k, i], rew[i]), f"{scenario}: reward of agent {i} differs at step {k}"
        assert torch.equal(got["done"][k], done)
    assert got["done"][4].all() and torch.equal(a.steps, b.steps)  # the time limit fell inside the rollout
    assert torch.equal(ha.state, hb.state)
    # the same rollout stored straight into a gather buffer (shard.NativeRollout): the kernel writes the caller's tensors
    from vectorizedmultiagentsimulator_amd.shard import EnvShard, NativeRollout
    _force_state_same_device(a, b, scenario)
    nr = NativeRollout(EnvShard(B, 0, 1), hb.fused.rollout_fields(K), torch.device(DEV))
    a_again = [a.step([u[k] for u in acts]) for k in range(K)]
    got2 = hb.fused.rollout(acts, out=nr.fields)
    assert got2["obs"].data_ptr() == nr.fields["obs"].data_ptr()
    assert all(torch.equal(nr.gather()["obs"][0][k, i], a_again[k][0][i]) for k in range(K) for i in range(len(a.agents)))
    ob, rb, db, _ = b.step([u[0] for u in acts])  # and single steps go on from where the rollout left the world
    oa, ra, da, _ = a.step([u[0] for u in acts])
    assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and all(torch.equal(x, y) for x, y in zip(ra, rb))
    ha.detach(); hb.detach()


RESET_CASES = [("balance", dict(n_agents=4)), ("transport", dict(n_packages=2)), ("navigation", dict(n_agents=4)),
               ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False))]


@pytest.mark.parametrize("scenario,kw", RESET_CASES)
def test_attached_fused_reset_where_resets_exactly_the_masked_environments(vmas, scenario, kw):
    """``handle.fused.reset_where(mask)`` on the reference's environment: the masked environments get a fresh initial state by the
    scenario's own placement rules (bounds, zero velocities, step counter, shaping terms consistent with the new positions -
    checked with the REFERENCE's formulas on the views), the others keep their bits; and the environment steps on, teacher-forced
    against the CPU reference from the state the reset left."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 300
    att = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, **kw)
    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=0, **kw)
    h = attach(att, fused=True)
    g = torch.Generator().manual_seed(23)
    for _ in range(6):
        att.step([a.to(DEV) for a in _actions(att, g)])
    before, steps_before = h.state.clone(), att.steps.clone()
    mask = torch.zeros(B, dtype=torch.bool)
    mask[::3] = True
    mask[B - 1] = True
    obs = h.fused.reset_where(mask.to(DEV), return_observations=True)
    assert len(obs) == len(att.agents) and obs[0].shape[0] == B
    m = mask.to(DEV)
    st = h.state[:, :, :B]
    assert torch.equal(st[:, :, ~m], before[:, :, :B][:, :, ~m]), "an unmasked environment was touched"
    assert not torch.equal(st[:, 0:2, m], before[:, :, :B][:, 0:2, m])
    assert torch.equal(att.steps[m], torch.zeros_like(att.steps[m])) and torch.equal(att.steps[~m], steps_before[~m])
    assert float(st[:, 2:4, m].abs().max()) == 0.0 and float(st[:, 5, m].abs().max()) == 0.0  # World.reset: velocities zeroed
    sc, w = att.scenario, att.world
    if scenario == "balance":
        want = torch.linalg.vector_norm(sc.package.state.pos - sc.package.goal.state.pos, dim=1) * sc.shaping_factor
        assert torch.allclose(sc.global_shaping[m], want[m], atol=1e-4, rtol=1e-5)
        assert float(sc.package.goal.state.pos[m, 1].min()) >= 0.0 and float(sc.package.goal.state.pos[m, 0].abs().max()) <= 1.0
        assert torch.allclose(sc.line.state.pos[m, 1], torch.full_like(sc.line.state.pos[m, 1], -w.y_semidim + sc.agent_radius * 2))
    if scenario == "transport":
        for p in sc.packages:
            want = torch.linalg.vector_norm(p.state.pos - p.goal.state.pos, dim=1) * sc.shaping_factor
            assert torch.allclose(p.global_shaping[m], want[m], atol=1e-4, rtol=1e-5)
            assert float(p.state.pos[m].abs().max()) <= sc.world_semidim
        pos = torch.stack([a.state.pos for a in w.agents], dim=1)[m]  # agents at least two radii apart
        d = torch.cdist(pos, pos) + torch.eye(len(w.agents), device=DEV) * 10
        assert float(d.min()) >= sc.agent_radius * 2 - 1e-6
    if scenario == "navigation":
        for a in w.agents:
            want = torch.linalg.vector_norm(a.state.pos - a.goal.state.pos, dim=1) * sc.pos_shaping_factor
            assert torch.allclose(a.pos_shaping[m], want[m], atol=1e-5, rtol=1e-5)
        ents = torch.stack([e.state.pos for e in w.entities], dim=1)[m]
        d = torch.cdist(ents, ents) + torch.eye(len(w.entities), device=DEV) * 10
        assert float(d.min()) >= sc.min_distance_between_entities - 1e-6 and float(ents.abs().max()) <= max(sc.world_spawning_x, sc.world_spawning_y)
    if scenario == "football":
        assert float(sc.ball.state.pos[m].abs().max()) == 0.0 and not bool(sc._done[m].any())
        for a in sc.blue_agents:
            assert float(a.state.pos[m, 0].max()) <= sc.agent_size + 1e-6
        for a in sc.red_agents:
            assert float(a.state.pos[m, 0].min()) >= -sc.agent_size - 1e-6 and torch.allclose(a.state.rot[m], torch.full_like(a.state.rot[m], torch.pi))
        want = torch.linalg.vector_norm(sc.ball.state.pos - sc.right_goal_pos, dim=-1) * sc.pos_shaping_factor_ball_goal
        assert torch.allclose(sc.ball.pos_shaping_blue[m], want[m], atol=1e-5, rtol=1e-5)
        for lm in w.landmarks:  # walls and goal lines where the reference's reset_walls / reset_goals put them
            assert torch.equal(lm.state.pos[m], lm.state.pos[~m][:1].expand_as(lm.state.pos[m]))
    # a second reset of the same environments draws another episode
    again = h.state.clone()
    h.fused.reset_where(mask.to(DEV))
    assert not torch.equal(h.state[:, 0:2, :B][:, :, m], again[:, 0:2, :B][:, :, m]) and torch.equal(h.state[:, :, :B][:, :, ~m], again[:, :, :B][:, :, ~m])
    # ... and the environment goes on: teacher-forced against the CPU reference from the state the reset left
    for ea, eb in zip(att.world.entities, ref.world.entities):
        eb.set_pos(ea.state.pos.cpu(), batch_index=None); eb.set_vel(ea.state.vel.cpu(), batch_index=None)
        eb.set_rot(ea.state.rot.cpu(), batch_index=None); eb.set_ang_vel(ea.state.ang_vel.cpu(), batch_index=None)
    for (oa, n), (ob, _) in zip(_terms(att, scenario), _terms(ref, scenario)):
        getattr(ob, n).copy_(getattr(oa, n).cpu())
    for aa, ab in zip(att.world.agents, ref.world.agents):
        if aa.state.force is not None:
            ab.state.force = aa.state.force.cpu()
    ref.steps.copy_(att.steps.cpu())
    with torch.no_grad():
        for t in range(3):
            _force_state(ref, att, scenario)
            acts = _actions(ref, g)
            _compare_step(ref.step([a.clone() for a in acts]), att.step([a.to(DEV) for a in acts]), scenario, f"{scenario} after reset_where t={t}")
    h.detach()


FOOTBALL_KW = dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False)


@pytest.mark.parametrize("scenario,kw,B", [
    ("balance", dict(n_agents=4), 4096), ("transport", dict(n_packages=2), 4096),
    # (round 6) the kinds whose step is more than one kernel or carries a grid barrier:
    ("navigation", dict(n_agents=4), 4096),     # collision reduction at a grid barrier inside the launch (64 tiles)
    ("navigation", dict(n_agents=4), 20000),    # ... by a second kernel behind the step (313 tiles > 256 CUs)
    ("football", FOOTBALL_KW, 4096),            # the post-step as the compacted kernel's epilogue
    ("football", FOOTBALL_KW, 20000),           # ... as a second kernel
], ids=lambda v: v if isinstance(v, (str, int)) else "")
def test_attached_fused_gated_validation_at_benchmark_size(vmas, scenario, kw, B):
    """The reference's asserts cost no idle queue: the check is enqueued, the step launched GATED on its result - every kernel of
    it - and the host waits behind both (vmas_env_validate_begin / vmas_world_step_env_gated / vmas_env_validate_end; a refused
    launch's host-side advances are taken back by vmas_world_gated_refused).  Same behaviour: a refused action leaves the
    world, the step counter AND the scenario's attributes exactly as they were; good steps are bitwise those of the
    unvalidated path - before AND after a refusal (navigation's barrier number / mask alternation must not slip)."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    a = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, **kw)
    b = vmas.make_env(scenario, num_envs=B, device=DEV, seed=0, **kw)
    ha, hb = attach(a, fused=True, validate_actions=True), attach(b, fused=True, validate_actions=False)
    assert ha.fused.one_launch and ha.fused.launch.can_gate(ha.fused.post.kind)
    assert ha.exact_broad_phase and ha.backend.exact_form() <= 1
    _force_state_same_device(b, a, scenario)
    g = torch.Generator().manual_seed(4)
    for t in range(5):
        acts = [x.to(DEV) for x in _actions(a, g)]
        oa, ra, da, _ = a.step([x.clone() for x in acts])
        ob, rb, db, _ = b.step(acts)
        assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and all(torch.equal(x, y) for x, y in zip(ra, rb)) and torch.equal(da, db)
    assert torch.equal(ha.state, hb.state)
    before, steps = ha.state.clone(), a.steps.clone()
    sc = a.scenario
    attrs = {"balance": ("pos_rew", "ground_rew", "on_the_ground"), "transport": ("rew",), "navigation": ("pos_rew", "final_rew"),
             "football": ("_sparse_reward_blue", "_done")}[scenario]
    kept = [getattr(sc, n) for n in attrs]
    kept_vals = [x.clone() for x in kept]
    shaping = [getattr(o, n).clone() for o, n in _terms(a, scenario)]
    good = [x.to(DEV) for x in _actions(a, g)]
    for poison, what in ((float("nan"), "NaN"), (1.7, "out of its range")):
        bad = [x.clone() for x in good]
        bad[-1][B - 3, 0] = poison
        with pytest.raises(AssertionError, match=what):
            a.step(bad)
        torch.cuda.synchronize()
        assert torch.equal(ha.state, before) and torch.equal(a.steps, steps), "a refused action must not touch the world"
        assert all(getattr(sc, n) is k for n, k in zip(attrs, kept)), "the scenario's attributes are the previous step's again"
        assert all(torch.equal(x, y) for x, y in zip(kept, kept_vals))
        assert all(torch.equal(getattr(o, n), s) for (o, n), s in zip(_terms(a, scenario), shaping))
    for t in range(3):  # the gate is open again, and the steps behind a refusal are the unvalidated path's, bit for bit
        oa, ra, da, _ = a.step([x.clone() for x in good])
        ob, rb, db, _ = b.step(good)
        assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and all(torch.equal(x, y) for x, y in zip(ra, rb)), f"step {t} after the refusals"
        assert torch.equal(ha.state, hb.state) and torch.equal(a.steps, b.steps) and torch.equal(da, db)
        good = [x.to(DEV) for x in _actions(a, g)]
    torch.cuda.synchronize()
    assert ha.backend.exact_status() == 0
    ha.detach(); hb.detach()


def test_attached_fused_follows_scenario_parameters_written_after_attach(vmas):
    """The reference reads its scenario's parameters at every step; the kernels' descriptors are built once.  A write after
    attach() - ``scenario.shaping_factor = 50`` - reaches the kernel at the next step; one that leaves the kernels' coverage -
    football's ``dense_reward = False`` - hands ``env.step`` back to the reference (still on the native World.step)."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 64
    ref = vmas.make_env("balance", num_envs=B, device="cpu", seed=0, n_agents=3)
    att = vmas.make_env("balance", num_envs=B, device=DEV, seed=0, n_agents=3)
    h = attach(att, fused=True)
    g = torch.Generator().manual_seed(8)
    for t in range(6):
        if t == 3:
            for e in (ref, att):
                e.scenario.shaping_factor = 50
                e.scenario.fall_reward = -3
        _force_state(ref, att, "balance")
        if t == 3:  # (the cached shaping term is in units of the old factor on both sides: keep them equal, not meaningful)
            pass
        acts = _actions(ref, g)
        _compare_step(ref.step([a.clone() for a in acts]), att.step([a.to(DEV) for a in acts]), "balance", f"balance params t={t}")
    assert h.fused is not None and h.fused.post.desc.shaping_factor == 50 and h.fused.post.desc.fall_reward == -3
    h.detach()
    kw = dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False)
    ref = vmas.make_env("football", num_envs=B, device="cpu", seed=0, **kw)
    att = vmas.make_env("football", num_envs=B, device=DEV, seed=0, **kw)
    h = attach(att)
    assert h.fused is not None
    for t in range(4):
        if t == 2:
            for e in (ref, att):
                e.scenario.dense_reward = False
        _force_state(ref, att, "football")
        acts = _actions(ref, g)
        _compare_step(ref.step([a.clone() for a in acts]), att.step([a.to(DEV) for a in acts]), "football", f"football params t={t}")
    assert h.fused is None and "dense_reward" in h.fused_reason and "step" not in att.__dict__
    h.detach()


def test_attached_fused_follows_max_steps_written_after_attach(vmas):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    att = vmas.make_env("transport", num_envs=64, device=DEV, seed=0)
    h = attach(att, fused=True)
    acts = [att.get_random_action(a) for a in att.agents]
    for _ in range(3):
        _, _, d, _ = att.step(acts)
    assert not bool(d.all())
    att.max_steps = 5
    _, _, d4, _ = att.step(acts)
    _, _, d5, _ = att.step(acts)
    assert not bool(d4.all()) and bool(d5.all())
    h.detach()
