"""The N > 1 path on REAL devices: two ranks, backend nccl (= RCCL over xGMI), one GPU each.  Skipped where fewer than two
devices are visible (the single-GPU test boxes): the first multi-GPU box exercises RCCL without anyone remembering to.

* a sharded `balance` environment per rank on its own GPU (real libvmas_hip.so), a 4-step rollout collected straight into
  the packed gather layout, ONE all_gather_into_tensor: every rank finds its own block where its shard range says and
  the peer's block equal to what the peer computed (exchanged checksums);
* `python bench.py --gpus 2` end to end: the line says n_gpus 2, both ranks' step times, the gather rates.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _need_two():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_envs, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        sys.path.insert(0, ROOT)
        from vectorizedmultiagentsimulator_amd.environment import make_env
        from vectorizedmultiagentsimulator_amd.rollout import collect_packed
        from vectorizedmultiagentsimulator_amd.shard import EnvShard

        sh = EnvShard.from_env(num_envs)
        env = make_env("balance", num_envs=sh.local_envs, device=dev, seed=sh.seed(0), n_agents=4, validate_actions=False)
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        policy = lambda obs: [(torch.rand(sh.local_envs, 2, device=dev, generator=g) * 2 - 1) * 0.8 for _ in env.agents]  # noqa: E731
        T = 4
        pr = collect_packed(env, policy, T, sh)
        local = {k: v.clone() for k, v in pr.views().items()}
        full = pr.gather()
        torch.cuda.synchronize()
        ok = full["obs"].shape == (num_envs, T, 4, 16) and full["rew"].shape == (num_envs, T, 4)
        for k in ("obs", "rew", "done"):
            ok &= torch.equal(full[k][sh.lo:sh.hi], local[k])
        # the peer's block: its checksum, sent through a second (tiny) collective
        mine = torch.stack([local[k].double().sum() for k in ("obs", "rew", "done")])
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        peer = EnvShard(num_envs, 1 - rank, world)
        got = torch.stack([full[k][peer.lo:peer.hi].double().sum() for k in ("obs", "rew", "done")])
        ok &= torch.allclose(got, sums[1 - rank], rtol=1e-12, atol=1e-9)
        ok &= torch.isfinite(full["obs"]).all().item()
        # the fast path: K steps in ONE launch (Environment.rollout) storing straight into the buffer the gather sends
        from vectorizedmultiagentsimulator_amd.rollout import collect_native

        K = 6
        acts = [(torch.rand(K, sh.local_envs, 2, device=dev, generator=g) * 2 - 1) * 0.8 for _ in env.agents]
        snap = env.get_state()
        want = env.rollout([a.clone() for a in acts])
        want = {k: v.clone() for k, v in want.items()}
        env.set_state(snap)
        nr = collect_native(env, acts, sh)
        gn = nr.gather()
        torch.cuda.synchronize()
        for k in ("obs", "rew", "done", "pos_rew"):
            mine_k = gn[k][rank]
            ok &= torch.equal(mine_k, want[k])
        flat = nr.env_major(gn, "obs")
        ok &= flat.shape == (K, 4, num_envs, 16) and torch.equal(flat[:, :, sh.lo:sh.hi], want["obs"])
        mine = torch.stack([want[k].double().sum() for k in ("obs", "rew")])
        sums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sums, mine)
        got = torch.stack([gn[k][1 - rank].double().sum() for k in ("obs", "rew")])
        ok &= torch.allclose(got, sums[1 - rank], rtol=1e-12, atol=1e-9)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_envs", [4096, 4097])  # equal and unequal shards
def test_two_rank_nccl_sharded_rollout_gather(num_envs):
    _need_two()
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_envs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_bench_two_gpus_end_to_end():
    _need_two()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert len(d["per_rank_us_per_step"]) == 2 and all(t > 0 for t in d["per_rank_us_per_step"])
    assert set(d["rollout_gather"]) == {"balance_cfg2", "navigation_cfg4", "football_cfg5"}
    assert all(v["collectives_per_chunk"] == 1 and v["GBps_received_per_gpu"] > 0 for v in d["rollout_gather"].values())
    assert d["value"] > 1e9
    sr = d["sharded_rollout"]
    assert sr["ranks_in_result"] == 2 and len(sr["per_rank_rollout_us_per_step"]) == 2 and sr["GBps_received_per_gpu"] > 0


def test_bench_two_gpus_strong_scaling_config():
    """`--config navigation --gpus 2`: BASELINE config 4 sharded (strong scaling: 65 536 environments in all)."""
    _need_two()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "navigation", "--steps", "50",
                          "--warmup", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_envs"] == 65536
    assert d["config"]["num_envs_per_gpu"] == 32768 and len(d["per_rank_us_per_step"]) == 2


def test_bench_n_greater_1_code_path_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu` (testing mode: both ranks on cuda:0, process group over gloo): the N > 1 code path of the
    line - environment sharding, barrier + synchronize fences, max over ranks, the attached headline on EVERY rank, the
    per-rank roofline - runs on the one-GPU boxes too.  Its numbers mean nothing (two processes share a device); its shape
    is what the driver's N = 2 / 4 / 8 runs will print."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--no-gather", "--steps", "30",
                          "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and "share_gpu" in d
    assert d["config"]["global_envs"] == 2 * d["config"]["num_envs_per_gpu"] and len(d["per_rank_us_per_step"]) == 2
    assert d["headline"]["fused"] and d["headline"]["launches_per_env_step"] == 1, "the attached reference's one-launch step on every rank"
    assert d["value"] == pytest.approx(d["config"]["global_envs"] * d["headline"]["substeps"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert d["world_step"]["value"] > 0 and d["roofline"]["frac"] is not None and 0 < d["roofline"]["frac"] < 1

