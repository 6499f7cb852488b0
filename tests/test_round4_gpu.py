"""Round-4 additions (GPU): the output-set pool behind Environment.step, rollout() into caller-provided buffers and the
buffer the end-of-rollout gather sends (shard.NativeRollout), attach()'s pre-marshalled step call and its static-change
detection by __setattr__ hooks."""
import pytest
import torch

pytestmark = pytest.mark.gpu

eq = lambda x, y: torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8))  # noqa: E731
CASES = [("balance", dict(n_agents=4), 4096), ("transport", {}, 700), ("navigation", dict(n_agents=4), 2048),
         ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False), 1000)]


def _pair(name, kw, B, seed=3):
    from vectorizedmultiagentsimulator_amd.environment import make_env

    a = make_env(name, num_envs=B, device="cuda:0", seed=seed, validate_actions=False, **kw)
    b = make_env(name, num_envs=B, device="cuda:0", seed=seed, validate_actions=False, **kw)
    b.set_state([t.clone() for t in a.get_state()])
    return a, b


@pytest.mark.parametrize("name,kw,B", CASES)
def test_step_outputs_held_by_the_caller_are_never_overwritten(name, kw, B):
    """Environment.step recycles an output set only when the caller holds nothing of it: results kept over many steps keep
    their values (compared with clones taken at once), and a caller that drops them gets recycled storage."""
    a, _ = _pair(name, kw, B)
    g = torch.Generator(device="cuda:0").manual_seed(11)
    held, clones = [], []
    for k in range(12):
        acts = [(torch.rand(B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.9 for _ in a.agents]
        obs, rew, done, info = a.step(acts)
        held.append((obs, rew, done, info))
        clones.append(([o.clone() for o in obs], [r.clone() for r in rew], done.clone(),
                       [{k_: (v.clone() if torch.is_tensor(v) else v) for k_, v in d.items()} for d in info]))
    torch.cuda.synchronize()
    for (obs, rew, done, info), (cobs, crew, cdone, cinfo) in zip(held, clones):
        assert all(eq(x, y) for x, y in zip(obs, cobs)) and all(eq(x, y) for x, y in zip(rew, crew)) and eq(done, cdone)
        for d, cd in zip(info, cinfo):
            assert all(eq(d[k_], cd[k_]) for k_ in d if torch.is_tensor(d[k_]))
    assert len({o[0][0].data_ptr() for o in held}) == len(held), "a held output set was handed out twice"
    del held, clones, obs, rew, done, info
    ptrs = set()
    for k in range(10):  # the usual loop: two sets alternate
        obs, rew, done, info = a.step(acts)
        ptrs.add(obs[0].data_ptr())
    assert len(ptrs) <= 3, f"{len(ptrs)} distinct output sets for a caller that keeps one step's results"


@pytest.mark.parametrize("name,kw,B", CASES)
def test_pooled_steps_equal_steps_of_a_twin(name, kw, B):
    """The pooled outputs are the same numbers a twin environment returns whose results are cloned at once."""
    a, b = _pair(name, kw, B)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for k in range(8):
        acts = [(torch.rand(B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.9 for _ in a.agents]
        ra = a.step(acts)
        rb = b.step([u.clone() for u in acts])
        assert all(eq(x, y) for x, y in zip(ra[0], rb[0])) and all(eq(x, y) for x, y in zip(ra[1], rb[1])) and eq(ra[2], rb[2])
    assert eq(a.world._state, b.world._state)


@pytest.mark.parametrize("name,kw,B", CASES)
def test_rollout_into_the_gather_buffer_is_the_plain_rollout(name, kw, B):
    """Environment.rollout(out=NativeRollout.fields): the kernel stores straight into the buffer the end-of-rollout gather
    sends - bitwise what rollout() returns in tensors of its own; gather() at world size 1 hands the same views back."""
    from vectorizedmultiagentsimulator_amd.rollout import collect_native
    from vectorizedmultiagentsimulator_amd.shard import EnvShard, NativeRollout

    a, b = _pair(name, kw, B)
    if not getattr(a._post, "rollout_ok", True):
        pytest.skip("no multi-step launch for this world size")
    g = torch.Generator(device="cuda:0").manual_seed(7)
    K = 5
    roll = [(torch.rand(K, B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.8 for _ in a.agents]
    want = a.rollout(roll)
    shard = EnvShard(B, 0, 1)
    nr = collect_native(b, [u.clone() for u in roll], shard)
    assert isinstance(nr, NativeRollout)
    for name_, shape, dtype in b.rollout_fields(K):
        assert eq(nr.fields[name_], want[name_]), name_
    assert eq(a.world._state, b.world._state)
    got = nr.gather()
    assert got["obs"].shape == (1,) + tuple(want["obs"].shape) and got["obs"].data_ptr() == nr.fields["obs"].data_ptr()
    assert eq(nr.env_major(got, "obs"), want["obs"]) and eq(nr.env_major(got, "done"), want["done"])
    # a second rollout into the same buffer
    roll2 = [(torch.rand(K, B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.8 for _ in a.agents]
    want2 = a.rollout(roll2)
    collect_native(b, roll2, shard, into=nr)
    assert eq(nr.fields["obs"], want2["obs"]) and eq(nr.fields["rew"], want2["rew"])
    with pytest.raises(AssertionError):
        b.rollout(roll2, out={**nr.fields, "obs": nr.fields["obs"][:, :, :-1]})  # a wrong shape is refused, not written through


@pytest.mark.parametrize("kw,B", [(dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), 3000),
                                  (dict(n_blue_agents=3, n_red_agents=2, ai_red_agents=False), 1000)])
def test_football_forms_are_bitwise_each_other(kw, B):
    """football's Environment.step / rollout have two forms (include/vmas_debug_hip.h: one launch with the post-step as the step
    kernel's epilogue - rollouts, single steps up to one tile per CU; step kernel + stand-alone post-step kernel - single steps
    beyond) and the library's own choice between them: the same device functions, so the same bits - observations (three
    different store patterns), rewards, done, info terms, shaping terms, steps, state."""
    import ctypes as C

    from vectorizedmultiagentsimulator_amd.environment import make_env

    envs = [make_env("football", num_envs=B, device="cuda:0", seed=3, validate_actions=False, max_steps=9, **kw) for _ in range(3)]
    for form, env in zip((0, 1, -1), envs):  # (max_steps: the step limit is applied inside every form)
        env.set_state([t.clone() for t in envs[0].get_state()])
        be = env.world._get_backend()
        be.lib.vmas_debug_football_form.argtypes = [C.c_void_p, C.c_int32]
        assert be.lib.vmas_debug_football_form(be._h, form) == 0
    a = envs[0]
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for k in range(6):
        acts = [(torch.rand(B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.9 for _ in a.agents]
        ra = a.step(acts)
        for b in envs[1:]:
            rb = b.step([u.clone() for u in acts])
            assert all(eq(x, y) for x, y in zip(ra[0], rb[0])), f"observations, step {k}"
            assert all(eq(x, y) for x, y in zip(ra[1], rb[1])) and eq(ra[2], rb[2])
            for da, db in zip(ra[3], rb[3]):
                assert all(eq(da[k_], db[k_]) for k_ in da if torch.is_tensor(da[k_]))
            assert eq(a.world._state, b.world._state) and eq(a.steps, b.steps)
    for K in (5, 2):
        roll = [(torch.rand(K, B, 2, device="cuda:0", generator=g) * 2 - 1) * 0.8 for _ in a.agents]
        wa = a.rollout(roll)
        for form, b in enumerate(envs[1:], 1):
            wb = b.rollout([u.clone() for u in roll])
            for name_ in wa:
                assert eq(wa[name_], wb[name_]), (name_, form, K)
            assert eq(a.world._state, b.world._state) and eq(a.steps, b.steps), (form, K)
            assert eq(a.world._agent_ft, b.world._agent_ft), (form, K)
            assert eq(a.scenario.ball.pos_shaping_blue, b.scenario.ball.pos_shaping_blue)
        if K == 5:
            assert bool(wa["done"].any()), "the step limit should have ended episodes inside the rollout"


@pytest.mark.reference
def test_attached_step_is_one_foreign_call_and_sees_static_changes():
    """attach(): world.step() goes through the pre-marshalled stepper (no per-step fingerprint); a mass written through the
    reference's setter, a collision filter re-assigned and a shape swapped are still noticed by the NEXT step."""
    from oracle import ref
    from vectorizedmultiagentsimulator_amd.adapter import attach

    vmas = ref.import_vmas()
    env = ref.make_env("balance", num_envs=256, device="cuda:0", seed=0, continuous_actions=True, n_agents=3)
    twin = ref.make_env("balance", num_envs=256, device="cuda:0", seed=0, continuous_actions=True, n_agents=3)
    h = attach(env)
    assert h._fast_step is not None and not h._dirty
    acts = [env.get_random_action(a) for a in env.agents]
    for _ in range(3):
        env.step(acts)
        twin.step(acts)
    assert h.refreshes == 0
    for e, t in zip(env.world.entities, twin.world.entities):
        assert torch.allclose(e.state.pos, t.state.pos, atol=1e-5), e.name
    env.world.agents[0].mass = 3.5   # core.py:634-636
    twin.world.agents[0].mass = 3.5
    assert h._dirty
    env.step(acts)
    twin.step(acts)
    assert h.refreshes == 1 and not h._dirty
    for e, t in zip(env.world.entities, twin.world.entities):
        assert torch.allclose(e.state.pos, t.state.pos, atol=1e-5), e.name
    env.world.agents[1].collision_filter = lambda e: False
    twin.world.agents[1].collision_filter = lambda e: False
    env.step(acts)
    twin.step(acts)
    assert h.refreshes == 2
    for e, t in zip(env.world.entities, twin.world.entities):
        assert torch.allclose(e.state.pos, t.state.pos, atol=1e-5), e.name
    for _ in range(5):  # nothing written: no re-extraction
        env.step(acts)
    assert h.refreshes == 2
    h.detach()
    assert vmas is not None
