"""attach(env, fused=...) on the reference's own Environment - the parts that need no GPU: which scenarios /
configurations the one-launch ``Environment.step`` claims (attached_env.plan_fuse and its pieces), and the views that
serve the reference's objects under the names fused.py reads.  The step itself: tests/test_attached_env_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def vmas():
    from oracle import ref

    ref.import_vmas()
    return ref


COVERED = [("balance", dict(n_agents=4)), ("balance", dict(n_agents=3, package_mass=2)), ("transport", {}),
           ("transport", dict(n_packages=2)), ("navigation", dict(n_agents=4)),
           ("navigation", dict(n_agents=5, shared_rew=False, observe_all_goals=True)), ("navigation", dict(n_agents=3, collisions=False)),
           ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False)),
           ("football", dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False, spawn_in_formation=True))]


@pytest.mark.parametrize("scenario,kw", COVERED)
def test_covered_configurations_are_claimed(vmas, scenario, kw):
    from vectorizedmultiagentsimulator_amd import attached_env as AE

    env = vmas.make_env(scenario, num_envs=3, device="cpu", seed=0, **kw)
    profile, reason = AE.find_profile(env)
    assert profile is not None and profile.module_tail == scenario, reason
    assert profile.check(env) is None
    al = profile.aliases(env)
    scripts = al["fused_agent_scripts"]() if "fused_agent_scripts" in al else []
    assert AE._ingest_reason(env, {id(s["agent"]) for s in scripts}) is None
    assert AE.plan_fuse(env) == (None, "not a GPU environment")  # (a CPU environment never gets the kernels)
    # the names fused.py's post-step classes read resolve on the view
    view = AE._ScenarioView(env.scenario, al)
    if scenario == "balance":
        assert view.goal is env.scenario.package.goal and view.package is env.scenario.package
    if scenario == "transport":
        assert view.goal is env.world.landmarks[0]
    if scenario == "football":
        red0 = env.scenario.red_agents[0]
        assert view.fused_action_factors(red0) == [-1.0, 1.0] and view.fused_action_factors(env.scenario.blue_agents[0]) is None
        assert scripts[0]["agent"] is env.scenario.ball and len(scripts[0]["params"]) == 4
    view.pos_rew = torch.zeros(3)  # writes land on the reference's scenario
    assert env.scenario.pos_rew is view.pos_rew
    view._lidar_cache = 1
    assert not hasattr(env.scenario, "_lidar_cache")


REFUSED = [("football", dict(n_blue_agents=2, n_red_agents=2), "ai_red_agents"),
           ("football", dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False, dense_reward=False), "dense_reward"),
           ("football", dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False, enable_shooting=True), "enable_shooting"),
           ("waterfall", {}, "no fused post-step kernel"), ("wheel", {}, "no fused post-step kernel")]


@pytest.mark.parametrize("scenario,kw,why", REFUSED)
def test_uncovered_configurations_are_refused_with_a_reason(vmas, scenario, kw, why):
    from vectorizedmultiagentsimulator_amd import attached_env as AE

    env = vmas.make_env(scenario, num_envs=3, device="cpu", seed=0, **kw)
    profile, reason = AE.find_profile(env)
    if profile is not None:
        reason = profile.check(env)
    assert reason is not None and why in reason, reason


def test_overridden_scenario_methods_are_refused(vmas):
    from vectorizedmultiagentsimulator_amd import attached_env as AE

    base = vmas.scenario_class("balance")

    class Mine(base):
        def reward(self, agent):
            return super().reward(agent) * 2

    env = vmas.make_env(Mine(), num_envs=3, device="cpu", seed=0, n_agents=3)
    assert AE.find_profile(env)[0] is None  # (its module is not vmas.scenarios.balance: not the reference's scenario as shipped)
    env = vmas.make_env("balance", num_envs=3, device="cpu", seed=0, n_agents=3)
    env.scenario.reward = lambda agent: torch.zeros(3)
    profile, reason = AE.find_profile(env)
    assert profile is None and "instance" in reason


def test_ingest_refuses_what_the_kernel_does_not_do(vmas):
    from vectorizedmultiagentsimulator_amd import attached_env as AE

    env = vmas.make_env("balance", num_envs=3, device="cpu", seed=0, n_agents=3, continuous_actions=False, multidiscrete_actions=True)
    assert AE._ingest_reason(env, set()) == "multidiscrete actions"
    env = vmas.make_env("balance", num_envs=3, device="cpu", seed=0, n_agents=3)
    env.agents[0].action._u_noise = 0.1
    assert AE._ingest_reason(env, set()) == "action noise"
    env = vmas.make_env("football", num_envs=3, device="cpu", seed=0, n_blue_agents=2, n_red_agents=2, ai_red_agents=False)
    assert AE._ingest_reason(env, set()) == "scripted agents"  # (the ball, unless the profile supplies its device script)
    env = vmas.make_env("wheel", num_envs=3, device="cpu", seed=0)
    assert AE._ingest_reason(env, set()) is None or "dynamics" in AE._ingest_reason(env, set())


def test_attach_with_a_custom_backend_keeps_the_reference_env_step(vmas):
    """fused=None never engages without the HIP library behind the world (the CPU-oracle backend of the plumbing tests);
    fused=True then fails loudly."""
    from ref_backend import OracleBackend
    from vectorizedmultiagentsimulator_amd.adapter import attach

    env = vmas.make_env("balance", num_envs=3, device="cpu", seed=0, n_agents=3)
    h = attach(env, backend_factory=OracleBackend)
    assert h.fused is None and "custom backend" in h.fused_reason and "step" not in env.__dict__
    h.detach()
    with pytest.raises(NotImplementedError):
        attach(env, backend_factory=OracleBackend, fused=True)
    h = attach(env, backend_factory=OracleBackend, fused=False)
    assert h.fused is None and h.fused_reason == "fused=False"
    h.exact_broad_phase = False  # (read at the next step: ADVICE r4)
    env.step([env.get_random_action(a) for a in env.agents])
    h.detach()


@pytest.mark.parametrize("scenario,kw,n_ops,n_terms,n_flags", [("balance", dict(n_agents=4), 8, 1, 1), ("transport", dict(n_packages=2), 7, 2, 2),
                                                               ("navigation", dict(n_agents=4), 8, 4, 0),
                                                               ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False), 18, 6, 1)])
def test_reset_programs_evaluate_on_the_reference_objects(vmas, scenario, kw, n_ops, n_terms, n_flags):
    """``handle.fused.reset_where``: the spawn program is stated once (this package's scenario of the same name) and evaluated on
    the REFERENCE's scenario through the view - every entity it names is one of the reference world's, every term a tensor of the
    reference's objects."""
    from vectorizedmultiagentsimulator_amd import attached_env as AE

    env = vmas.make_env(scenario, num_envs=3, device="cpu", seed=0, **kw)
    for i, e in enumerate(env.world.entities):
        e.__dict__["_index"] = i
    profile, _ = AE.find_profile(env)
    prog = profile.reset_program(AE._ScenarioView(env.scenario, profile.aliases(env)))
    assert (len(prog["ops"]), len(prog["terms"]), len(prog["flags"])) == (n_ops, n_terms, n_flags)
    ents = set(map(id, env.world.entities))
    assert all(id(op[1]) in ents for op in prog["ops"])
    unplaced = ents - {id(op[1]) for op in prog["ops"]}  # (football's ball stays where World.reset zeroes it: the centre)
    assert unplaced == ({id(env.scenario.ball)} if scenario == "football" else set()), "every other entity is placed by the program"
    for t in prog["terms"]:
        x = t[0]() if callable(t[0]) else t[0]
        assert isinstance(x, torch.Tensor) and x.shape == (3,)
    # configurations whose reset is not a spawn program say so
    env = vmas.make_env("navigation", num_envs=3, device="cpu", seed=0, n_agents=4, collisions=False, agents_with_same_goal=4)
    profile, _ = AE.find_profile(env)
    assert profile.reset_program(AE._ScenarioView(env.scenario, profile.aliases(env))) is None
