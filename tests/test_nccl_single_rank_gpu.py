"""Backend `nccl` (= RCCL) on the ONE GPU the test boxes have: a single-rank process group, the K-step rollout kernel storing
straight into the buffer that `all_gather_into_tensor` then sends.  It is not a scaling test - with one rank the collective
moves nothing between devices - but it runs the whole N > 1 code path of the pipeline's only exchange on real
hardware: RCCL initialises on the box, takes the kernel's output buffer as it is on the caller's stream, and returns it
bit for bit.  The two-rank form of the same test is tests/test_nccl_two_ranks_gpu.py (skipped on one GPU)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sys.path.insert(0, ROOT)
        from vectorizedmultiagentsimulator_amd.environment import make_env
        from vectorizedmultiagentsimulator_amd.rollout import collect_native
        from vectorizedmultiagentsimulator_amd.shard import EnvShard, NativeRollout

        assert dist.get_backend() == "nccl"
        sh = EnvShard.from_env(4096)
        env = make_env("balance", num_envs=sh.local_envs, device=dev, seed=sh.seed(0), n_agents=4, validate_actions=False)
        g = torch.Generator(device=dev).manual_seed(7)
        K = 6
        acts = [(torch.rand(K, sh.local_envs, 2, device=dev, generator=g) * 2 - 1) * 0.8 for _ in env.agents]
        snap = env.get_state()
        want = {k: v.clone() for k, v in env.rollout([a.clone() for a in acts]).items()}
        env.set_state(snap)
        nr = NativeRollout.for_env(sh, env, K)
        nr.force_collective = True
        collect_native(env, acts, sh, into=nr)
        gathered = nr.gather()  # dist.all_gather_into_tensor on the kernel's own output buffer, same stream
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        ok = nr._full is not None and nr._full.data_ptr() != nr.local.data_ptr()
        for k in ("obs", "rew", "done", "pos_rew"):
            ok &= gathered[k].shape[0] == 1 and torch.equal(gathered[k][0], want[k])
        ok &= float(t.item()) == 1.0
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_single_rank_nccl_gathers_the_kernels_rollout_buffer():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    ok = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and ok
