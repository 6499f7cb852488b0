"""N>1 path on CPU: two gloo processes shard a batch and gather a rollout."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vectorizedmultiagentsimulator_amd.shard import EnvShard, NativeRollout, PackedRollout, RolloutGather, max_over_ranks, shard_range


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 64, 32768, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_envs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = EnvShard.from_env(num_envs)
        assert (sh.rank, sh.world_size) == (rank, world)
        T, A, D = 5, 3, 4
        env = torch.arange(sh.lo, sh.hi, dtype=torch.float32)
        t = torch.arange(T, dtype=torch.float32)
        obs = (t[:, None, None, None] * 1000 + env[None, :, None, None] + torch.arange(A)[None, None, :, None] * 0.1
               + torch.arange(D)[None, None, None, :] * 0.01)
        rew = t[:, None, None] - env[None, :, None] + torch.arange(A)[None, None, :]
        done = (env[None, :] + t[:, None]) % 3 == 0
        out = RolloutGather(sh).gather({"obs": obs, "rew": rew, "done": done}, env_dim=1)
        genv = torch.arange(num_envs, dtype=torch.float32)
        want_obs = (t[:, None, None, None] * 1000 + genv[None, :, None, None] + torch.arange(A)[None, None, :, None] * 0.1
                    + torch.arange(D)[None, None, None, :] * 0.01)
        ok = torch.equal(out["obs"], want_obs)
        ok &= torch.equal(out["rew"], t[:, None, None] - genv[None, :, None] + torch.arange(A)[None, None, :])
        ok &= torch.equal(out["done"], (genv[None, :] + t[:, None]) % 3 == 0) and out["done"].dtype == torch.bool
        # the packed form: ONE [b, T, W] buffer per rank, environment axis first, ONE collective, results = views
        pr = PackedRollout(sh, T, A, D, "cpu")
        v = pr.views()
        v["obs"].copy_(obs.movedim(1, 0)); v["rew"].copy_(rew.movedim(1, 0)); v["done"].copy_(done.movedim(1, 0).float())
        calls = []
        orig = dist.all_gather_into_tensor
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            g = pr.gather()
        finally:
            dist.all_gather_into_tensor = orig
        ok &= len(calls) == 1
        ok &= torch.equal(g["obs"].movedim(0, 1), want_obs) and g["obs"].shape == (num_envs, T, A, D)
        ok &= torch.equal(g["rew"].movedim(0, 1), t[:, None, None] - genv[None, :, None] + torch.arange(A)[None, None, :])
        ok &= torch.equal(g["done"].movedim(0, 1) > 0.5, (genv[None, :] + t[:, None]) % 3 == 0)
        if num_envs % world == 0:  # equal shards: the results are views of the gathered buffer (no copy behind the collective)
            ok &= g["obs"].untyped_storage().data_ptr() == pr._full.untyped_storage().data_ptr()
        # the native form: the layout the K-step rollout kernel writes ([K, A, b, D] ...), one byte buffer per rank, ONE
        # collective; the gathered tensors carry the rank as a leading axis (rank order IS environment order)
        b = sh.local_envs
        fields = [("obs", (T, A, b, D), torch.float32), ("rew", (T, A, b), torch.float32), ("done", (T, b), torch.bool),
                  ("pos_rew", (T, b), torch.float32)]
        nr = NativeRollout(sh, fields, "cpu")
        nr.fields["obs"].copy_(obs.movedim(1, 2)); nr.fields["rew"].copy_(rew.movedim(1, 2)); nr.fields["done"].copy_(done)
        nr.fields["pos_rew"].copy_(rew[:, :, 0])
        calls = []
        dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            gn = nr.gather()
        finally:
            dist.all_gather_into_tensor = orig
        ok &= len(calls) == 1
        ok &= torch.equal(nr.env_major(gn, "obs"), want_obs.movedim(1, 2)) and nr.env_major(gn, "obs").shape == (T, A, num_envs, D)
        ok &= torch.equal(nr.env_major(gn, "rew"), (t[:, None, None] - genv[None, :, None] + torch.arange(A)[None, None, :]).movedim(1, 2))
        ok &= torch.equal(nr.env_major(gn, "done"), (genv[None, :] + t[:, None]) % 3 == 0) and nr.env_major(gn, "done").dtype == torch.bool
        ok &= torch.equal(nr.env_major(gn, "pos_rew"), t[:, None] - genv[None, :])
        if num_envs % world == 0:  # equal shards: [R, *shape] views of the gathered buffer
            ok &= gn["obs"].shape == (world, T, A, b, D) and gn["obs"].untyped_storage().data_ptr() == nr._full.untyped_storage().data_ptr()
            ok &= torch.equal(gn["obs"][rank], nr.fields["obs"])
        else:
            ok &= isinstance(gn["obs"], list) and [x.shape[2] for x in gn["obs"]] == [shard_range(num_envs, r, world)[1] - shard_range(num_envs, r, world)[0] for r in range(world)]
        slowest = max_over_ranks(float(rank + 1), "cpu")
        ok &= slowest == float(world)
        q.put((rank, bool(ok), sh.seed(0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_envs", [10, 11])  # equal and unequal shards
def test_two_rank_rollout_gather(num_envs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_envs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({s for _, _, s in res}) == 2  # distinct per-shard seeds


# ---------------------------------------------------------------------------------------------------------------
# The whole N > 1 pipeline on two ranks: EnvShard -> a real sharded environment (native object model, per-shard seed,
# scenario reset) -> rollout.collect -> rollout.gather_rollout.  No GPU here: the World.step of each shard runs on the
# CPU oracle injected as the backend (test infrastructure; the product has no CPU path).
def _sharded_env(shard):
    from ref_backend import OracleBackend
    from vectorizedmultiagentsimulator_amd.environment import make_env

    env = make_env("balance", num_envs=shard.local_envs, device="cpu", seed=shard.seed(0), n_agents=3)
    w = env.world
    w._backend = OracleBackend(w.spec, shard.local_envs, "cpu", w._packed_state(), w._packed_agent_ft())
    return env


def _policy_for(shard, n_agents):
    """Deterministic actions that depend on the GLOBAL environment index and the step."""
    step = [0]

    def policy(obs):
        t = step[0]
        step[0] += 1
        genv = torch.arange(shard.lo, shard.hi, dtype=torch.float32)
        return [torch.stack([torch.sin(0.37 * genv + 0.11 * t + a), torch.cos(0.23 * genv - 0.07 * t + a)], dim=1) * 0.9
                for a in range(n_agents)]

    return policy


def _rollout_worker(rank, world, port, num_envs, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vectorizedmultiagentsimulator_amd.rollout import collect, gather_rollout

        T = 5
        sh = EnvShard.from_env(num_envs)
        env = _sharded_env(sh)
        local = collect(env, _policy_for(sh, 3), T)
        assert local["obs"].shape == (T, sh.local_envs, 3, 16) and local["done"].dtype == torch.bool
        full = gather_rollout(local, sh)
        ok = full["obs"].shape == (T, num_envs, 3, 16) and full["rew"].shape == (T, num_envs, 3)
        # global environment order: every rank finds its own block where its shard range says
        for k in ("obs", "rew", "done"):
            ok &= torch.equal(full[k][:, sh.lo:sh.hi], local[k])
        # ... and the PEER's block is exactly what an environment of the peer's shard (its size, its seed, the same
        # policy of the global index) produces: recomputed here, bit for bit (the CPU oracle is deterministic)
        peer = EnvShard(num_envs, 1 - rank, world)
        want = collect(_sharded_env(peer), _policy_for(peer, 3), T)
        for k in ("obs", "rew", "done"):
            ok &= torch.equal(full[k][:, peer.lo:peer.hi], want[k])
        # the same rollout collected straight into the gather's layout (collect_packed): one collective, same numbers
        from vectorizedmultiagentsimulator_amd.rollout import collect_packed
        pr = collect_packed(_sharded_env(sh), _policy_for(sh, 3), T, sh)
        gp = pr.gather()
        ok &= torch.equal(gp["obs"].movedim(0, 1), full["obs"]) and torch.equal(gp["rew"].movedim(0, 1), full["rew"])
        ok &= torch.equal(gp["done"].movedim(0, 1) > 0.5, full["done"])
        differs = not torch.equal(local["obs"][0, 0], want["obs"][0, 0])  # per-shard seeds: different reset states
        q.put((rank, bool(ok), bool(differs), sh.seed(0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_envs", [8, 9])  # equal and unequal shards
def test_two_rank_sharded_environment_rollout_and_gather(num_envs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rollout_worker, args=(r, 2, port, num_envs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert all(d for _, _, d, _ in res), "both shards drew the same reset state: per-shard seeds are not applied"
    assert len({s for _, _, _, s in res}) == 2


def test_bench_dry_run_spawns_its_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment re-executes itself under torch.distributed.run;
    --dry-run keeps it on the CPU (gloo, no physics): the line says n_gpus 2 and the group really had 2 ranks."""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dry_run"] is True
    assert set(d["rollout_gather"]) == {"balance_cfg2", "navigation_cfg4", "football_cfg5"}


@pytest.mark.parametrize("config,per_gpu,agents,obs_dim", [("navigation", 8192, 8, 18), ("football", 16384, 10, 88)])
def test_bench_dry_run_eight_ranks_strong_scaling_plan(config, per_gpu, agents, obs_dim):
    """What the driver's 8-GPU run of BASELINE configs 4 / 5 does, minus the physics (no multi-GPU box is available to the
    builder): `bench.py --gpus 8 --config C --strong` launches 8 ranks, shards the configuration's batch into contiguous
    blocks of 8 192 / 16 384 environments, chunks the rollout so that a gathered buffer stays below 2 GB, and ONE
    all_gather_into_tensor per chunk returns all eight ranks' blocks in rank = environment order."""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--config", config, "--strong", "--dry-run",
                          "--no-gather"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "strong" and d["shard"] == [0, per_gpu]
    plan = d["sharding_plan"]
    assert plan["envs_per_gpu"] == per_gpu and plan["global_envs"] == 8 * per_gpu
    assert plan["bytes_per_step_all_ranks"] == 8 * per_gpu * (agents * obs_dim + agents + 1) * 4
    assert plan["chunk_bytes"] <= 2 << 30 and plan["steps_per_chunk"] >= 1
    assert (plan["steps_per_chunk"] + 1) * plan["bytes_per_step_all_ranks"] > 2 << 30 or plan["steps_per_chunk"] == 100
    assert plan["ranks_in_result"] == 8 and plan["rank_order_kept"] and plan["collectives_per_chunk"] == 1


def test_bench_refuses_to_claim_more_gpus_than_it_sees():
    """`python bench.py --gpus 2` where fewer than two devices are visible (here: none) must fail loudly instead of
    printing an n_gpus=2 line."""
    import subprocess
    import sys

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the real run is the driver's")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "--gpus 2 but only" in (out.stderr + out.stdout)
    assert not any(l.startswith("{") for l in out.stdout.splitlines())
