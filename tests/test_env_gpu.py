"""End-to-end on the GPU: make_env -> Environment.step through the native World (gpu)."""
import numpy as np
import pytest
import torch

from golden_util import compare_state, ulp_sensitivity

pytestmark = pytest.mark.gpu

CASES = [("balance", dict(n_agents=4), 16), ("transport", {}, 11), ("transport", dict(n_packages=2), 18),
         ("navigation", dict(n_agents=8), 18), ("football", dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), 88)]


@pytest.mark.parametrize("name,kw,obs_dim", CASES)
def test_env_rollout_and_physics_vs_oracle(name, kw, obs_dim):
    from oracle.oracle import Oracle
    from vectorizedmultiagentsimulator_amd.environment import make_env

    B = 300
    env = make_env(name, num_envs=B, device="cuda:0", seed=1, **kw)
    o = Oracle(env.world.spec)
    nA = len(env.world.agents)
    for t in range(30):
        acts = [env.get_random_action(a) for a in env.agents]
        st0 = env.world._state.cpu().numpy().copy()
        obs, rews, dones, infos = env.step(acts)
        assert len(obs) == len(env.agents) and obs[0].shape == (B, obs_dim), obs[0].shape
        assert rews[0].shape == (B,) and dones.shape == (B,) and dones.dtype == torch.bool
        assert all(torch.isfinite(x).all() for x in obs) and all(torch.isfinite(r).all() for r in rews)
        if t % 10 == 0:  # the step the environment just made == oracle step from the same state/forces
            ft = env.world._agent_ft.cpu().numpy().copy()[:nA]
            want = st0.copy()
            assert env.world.exact_broad_phase  # (the default below 1024 environments: the reference's broad phase)
            sens = ulp_sensitivity(lambda a, b: o.step_exact(a, b, batch=B), st0, ft)
            o.step_exact(want, ft.copy(), batch=B)
            got = env.world._state.cpu().numpy()
            compare_state(got[:, :, :B], want[:, :, :B], f"{name} env.step physics t={t}", sens=sens[:, :, :B])
    env.reset_at(3)
    obs = env.reset()
    assert obs[0].shape == (B, obs_dim)


def test_discrete_actions_map_like_the_reference():
    """3-way discretisation per dimension: 0 -> stay, then decrement / increment (environment.py:657-705)."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    env = make_env("balance", num_envs=9, device="cuda:0", continuous_actions=False, n_agents=3, seed=0)
    a = env.agents[0]
    env._set_action(torch.arange(9, device="cuda:0").unsqueeze(-1), a)
    u = (a.action.u / 0.7).cpu()
    want = torch.tensor([[0, 0], [0, -1], [0, 1], [-1, 0], [-1, -1], [-1, 1], [1, 0], [1, -1], [1, 1]], dtype=torch.float32)
    assert torch.allclose(u, want)


@pytest.mark.parametrize("fused", [False, None])
@pytest.mark.parametrize("name,kw", [("balance", dict(n_agents=4)), ("transport", {}), ("navigation", dict(n_agents=4)),
                                     ("football", dict(n_blue_agents=3, n_red_agents=3, ai_red_agents=False))])
def test_graph_captured_step_equals_eager(name, kw, fused):
    """Environment(graph=True) gives the same observations / rewards / dones as the eager path, step after
    step, from the very first call (the capture's warm-up steps are undone).  fused=False: the whole
    tensor-op step is one HIP graph; default: one-launch scenarios run as they are, navigation replays
    its post-physics launches."""
    from vectorizedmultiagentsimulator_amd.environment import make_env

    B = 200
    eager = make_env(name, num_envs=B, device="cuda:0", seed=5, validate_actions=False, fused=fused, **kw)
    graphed = make_env(name, num_envs=B, device="cuda:0", seed=5, validate_actions=False, graph=True, fused=fused, **kw)
    assert torch.equal(graphed.world._state, eager.world._state)  # same seed, same reset
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for t in range(15):
        acts = [torch.rand(B, 2, device="cuda:0", generator=g) * 2 - 1 for _ in eager.agents]
        o1, r1, d1, _ = eager.step(acts)
        o2, r2, d2, _ = graphed.step(acts)
        for a, b in zip(o1, o2):
            assert torch.equal(a, b), f"{name} obs differ at step {t}: {(a - b).abs().max()}"
        for a, b in zip(r1, r2):
            assert torch.equal(a, b), f"{name} rewards differ at step {t}"
        assert torch.equal(d1, d2) and torch.equal(eager.steps, graphed.steps)
        if t == 7:
            eager.reset(seed=21)
            graphed.reset(seed=21)


def test_rollout_collect_and_single_rank_gather():
    from vectorizedmultiagentsimulator_amd.environment import make_env
    from vectorizedmultiagentsimulator_amd.rollout import collect, gather_rollout
    from vectorizedmultiagentsimulator_amd.shard import EnvShard

    env = make_env("balance", num_envs=128, device="cuda:0", seed=2, n_agents=3, validate_actions=False)
    buf = collect(env, lambda obs: [env.get_random_action(a) for a in env.agents], 12)
    assert buf["obs"].shape == (12, 128, 3, 16) and buf["rew"].shape == (12, 128, 3) and buf["done"].shape == (12, 128)
    assert torch.isfinite(buf["obs"]).all()
    g = gather_rollout(buf, EnvShard(128, 0, 1))
    assert all(torch.equal(g[k], buf[k]) for k in buf)
    env = make_env("balance", num_envs=128, device="cuda:0", seed=2, n_agents=3, validate_actions=False, max_steps=5)
    buf = collect(env, lambda obs: [env.get_random_action(a) for a in env.agents], 12, auto_reset=True)
    assert buf["done"][4].all() and buf["done"][9].all() and not buf["done"][5].any()  # restarted by the time limit
