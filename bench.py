#!/usr/bin/env python
"""bench.py - BASELINE.json's metric on the MI355X-native hot path.

metric : env-steps/sec (batch x substeps) on `balance`, 32768 envs per GPU, n_agents=4
step   : ONE World.step() (core.py:1972-2015) over the whole batch = one fused kernel launch; state and the
         pre-generated agent forces are resident in HBM before the timed region starts.  This is the north-star
         hot path and what `value` reports.  The same line carries, as a peer field, `env_step`: the rate through
         make_env('balance').step() (SURVEY.md 8d's definition: action ingest + World.step + reward / observation /
         done, ONE launch) with its own roofline.
timing : W warm-up steps, then exactly K steps between barrier+synchronize pairs; HIP events recorded on the launch
         stream right inside the two fences bracket the same K launches.  `ms_per_step` and `value` come from the
         events (MAX over ranks): with K = 20 the wall clock around a 0.2 ms region is mostly the cost of the fences
         themselves; the wall-clock figures are kept beside them (`wall`).
multi-GPU: `--gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N
         ranks (one per GPU, backend nccl = RCCL); under a launcher it reads RANK/LOCAL_RANK/WORLD_SIZE.  The batch
         is sharded by environment (weak scaling: 32768 envs per GPU), NO collective on the step path; the only
         exchange of the pipeline - the end-of-rollout all-gather of obs/rew/done - is timed
         separately (`rollout_gather`) and is not part of `value`.
roofline: achieved = algorithmic bytes per launch (384 B/env x envs) / average launch duration from
         the HIP events; achieved GFLOP/s beside it (1.7 kflop per env-step) and which bound binds.
cpu_baseline: the REFERENCE itself (VMAS, `kind: "reference"`): its World.step (core.py:1972) and its
         Environment.step (environment.py:325) on the host cores, device="cpu", torch threads = all cores, same
         initial state and actions, bounded to ~10 s each; rank 0, N=1 only.  The reference is imported from
         /root/reference when present, else from its byte-compiled build oracle/_ref (made by
         __graft_entry__.build()).  `cpu_port` = the C oracle with OpenMP (a scalar port, the conservative baseline).

Episodes are 100 steps long (actions ~ U(-1,1) * u_multiplier, reference law environment.py:536-548): every 100
steps the post-reset state is restored by a device-to-device copy inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPISODE = 100
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_GFLOPS = 157286.4  # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz (vector fp32, no MFMA)
FLOP_PER_ENV_STEP = 1700.0   # balance n_agents=4, SURVEY.md section 8d
POST_BYTES_PER_ENV = 273     # obs 4 x 16 x 4 + rew 4 x 4 + done 1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--clock-warmup", type=float, default=0.25,
                    help="seconds of untimed launches before the W warm-up steps (the GPU's clocks ramp up: see bench.py)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step window is timed this many times (each between its own fences); `value` = the median window")
    ap.add_argument("--num-envs", type=int, default=32768, help="environments PER GPU")
    ap.add_argument("--n-agents", type=int, default=4)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per environment (0 = library default)")
    ap.add_argument("--queues", type=int, default=0, help="HIP queues of vmas_world_step_n (0 = library's choice, 1..4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary measurements (env_step, persistent rollout)")
    ap.add_argument("--no-gather", action="store_true", help="skip the rollout all-gather timing (N > 1)")
    ap.add_argument("--fused", action="store_true",
                    help="time vmas_world_rollout (persistent launch, state resident in LDS) instead of one launch per step")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU plumbing check (tests): ranks, sharding and the rollout gather over gloo, NO physics, value = null")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py --gpus N`."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.dry_run:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible: refusing to report an "
                     f"n_gpus={args.gpus} line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def build_world(num_envs, device, n_agents, lanes, seed):
    import torch
    from vectorizedmultiagentsimulator_amd.scenarios.balance import Scenario

    torch.manual_seed(seed)
    sc = Scenario()
    w = sc.env_make_world(num_envs, device, n_agents=n_agents, lanes_per_env=lanes)
    sc.env_reset_world_at(None)
    return sc, w


def make_actions(n_steps, n_agents, num_envs, seed):
    """[n_steps, A, B, 2] ~ U(-1, 1) on the host (the reference's get_random_action law, u_range = 1)."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.rand(n_steps, n_agents, num_envs, 2, generator=g) * 2 - 1


def pack_forces(w, actions, device):
    """[n_steps, A, 3, ld] packed agent forces: u * u_multiplier (what Environment._set_action + Holonomic produce)."""
    import torch

    n_steps, A = actions.shape[0], len(w.agents)
    f = torch.zeros(n_steps, max(A, 1), 3, w._ld)
    for i, a in enumerate(w.agents):
        f[:, i, 0:2, : w.batch_dim] = actions[:, i].transpose(1, 2) * float(a.u_range) * float(a.u_multiplier)
    return f.to(device)


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_port(w, forces_cpu, state0_cpu, budget_s=6.0):
    """C oracle (scalar port of the reference algorithm, OpenMP over environments) on the host cores."""
    from oracle.oracle import Oracle

    o = Oracle(w.spec)
    B = w.batch_dim
    ncpu = os.cpu_count() or 1
    best = (0.0, 1)  # the thread count that is actually fastest on this host; 3 steps per candidate
    for th in sorted({1, 8, 16, 32, 64, 128, ncpu}):
        if th > ncpu:
            continue
        st = state0_cpu.copy()
        o.step(st, forces_cpu[0].copy(), batch=B, threads=th)
        t0 = time.perf_counter()
        for i in range(3):
            o.step(st, forces_cpu[i].copy(), batch=B, threads=th)
        rate = 3 * B / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, th)
    threads = best[1]
    st = state0_cpu.copy()
    n, t0 = 0, time.perf_counter()
    while True:
        if n % EPISODE == 0:
            st[...] = state0_cpu
        o.step(st, forces_cpu[n % forces_cpu.shape[0]].copy(), batch=B, threads=threads)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    return {"value": B * n * w.substeps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} World.step() of balance x {B} envs (same state/forces as the GPU run), {el:.1f} s, C oracle "
                      f"with OpenMP over environments"}


def cpu_reference(w, actions, state0_cpu, n_agents, budget_s=10.0):
    """The reference's own World.step and Environment.step on the host cores (device="cpu")."""
    import torch
    from oracle import ref

    B = w.batch_dim
    ncpu = os.cpu_count() or 1
    env = ref.make_env("balance", num_envs=B, device="cpu", seed=0, continuous_actions=True, n_agents=n_agents)
    world = env.world
    assert [e.name for e in world.entities] == [e.name for e in w.entities], "entity order differs from the reference's"
    st0 = torch.from_numpy(state0_cpu)

    def restore():
        for i, e in enumerate(world.entities):  # the GPU run's post-reset state, through the reference's own setters
            e.set_pos(st0[i, 0:2, :B].T.clone(), batch_index=None)
            e.set_vel(st0[i, 2:4, :B].T.clone(), batch_index=None)
            e.set_rot(st0[i, 4:5, :B].T.clone(), batch_index=None)
            e.set_ang_vel(st0[i, 5:6, :B].T.clone(), batch_index=None)

    def world_step(k):
        for i, a in enumerate(world.agents):  # what _set_action + Holonomic.process_action leave in the state
            a.state.force = actions[k % EPISODE, i] * a.u_range * a.u_multiplier
        world.step()

    def env_step(k):
        env.step([actions[k % EPISODE, i] * a.u_range for i, a in enumerate(env.agents)])

    # torch's intra-op thread count that is actually fastest on this host: World.step is ~2800 small aten calls per step,
    # and os.cpu_count() threads (the survey's plan) can be pathological - 256 threads measured 58 s per step on a pool
    # box against 50 ms with 8.  Probed in ascending order on World.step, stopping once a count is clearly slower.
    probed, best = {}, (float("inf"), 1)
    with torch.no_grad():
        for th in sorted({4, 8, 16, 32, 64, min(ncpu, 128), ncpu}):
            if th > ncpu:
                continue
            torch.set_num_threads(th)
            restore()
            world_step(0)
            t0 = time.perf_counter()
            world_step(1)
            dt = time.perf_counter() - t0
            probed[th] = round(dt * 1e3, 2)
            if dt < best[0]:
                best = (dt, th)
            elif dt > 1.3 * best[0]:
                break
    threads = best[1]
    torch.set_num_threads(threads)
    out = {}
    with torch.no_grad():
        for name, fn in (("world_step", world_step), ("env_step", env_step)):
            restore()
            for k in range(2):
                fn(k)
            restore()
            n, t0 = 0, time.perf_counter()
            while True:
                fn(n)
                n += 1
                el = time.perf_counter() - t0
                if el > budget_s or n >= EPISODE:
                    break
            out[name] = {"value": B * n * w.substeps / el, "ms_per_step": el / n * 1e3, "steps": n, "seconds": el}
    return {
        "value": out["world_step"]["value"], "unit": "env-steps/s", "cores": threads, "kind": "reference",
        "torch_threads": torch.get_num_threads(), "host_cpus": ncpu,
        "ms_per_world_step_by_threads": probed,
        "sample": f"{out['world_step']['steps']} World.step() (vmas/simulator/core.py:1972) of the reference's balance "
                  f"n_agents={n_agents} x {B} envs on device='cpu', same initial state and actions as the GPU run, "
                  f"{out['world_step']['seconds']:.1f} s ({out['world_step']['ms_per_step']:.1f} ms/step); 2 warm-up steps",
        "env_step": {**out["env_step"], "unit": "env-steps/s",
                     "note": "the reference's Environment.step (environment.py:325): ingest + World.step + reward/obs/done"},
        "reference_from": ref.root(),
    }


# ------------------------------------------------------------------------------------------------ gather
def time_rollout_gather(dist, shard_cls, packed_cls, rank, world_size, device, per_gpu_envs, t_steps=100,
                        max_chunk_bytes=2 << 30, shrink=1):
    """End-of-rollout all-gather, the ONLY collective of the pipeline, for the shapes of BASELINE configs
    2 / 4 / 5: every rank's rollout chunk is ONE packed buffer [b, T, W] (environment axis first: observations, rewards,
    done of a step side by side, shard.PackedRollout) and the exchange ONE all_gather_into_tensor per chunk, straight into
    the global buffer - no copies around it.  Chunks of steps keep the gathered buffer below ``max_chunk_bytes`` (config 5:
    46 GB per 100-step rollout on 8 GPUs)."""
    import torch

    shapes = {  # name: (envs per GPU, agents, obs dim)
        "balance_cfg2": (per_gpu_envs, 4, 16),
        "navigation_cfg4": (max(65536 // world_size // shrink, 1), 8, 18),
        "football_cfg5": (max(131072 // world_size // shrink, 1), 10, 88),
    }
    out = {}
    for name, (b, A, D) in shapes.items():
        shard = shard_cls(b * world_size, rank, world_size)
        step_bytes = b * (A * D + A + 1) * 4 * world_size
        tc = max(1, min(t_steps, max_chunk_bytes // step_bytes))
        pr = packed_cls(shard, tc, A, D, device)
        pr.gather()  # warm-up (communicator set-up, the output buffer)
        _sync(device)
        dist.barrier()
        t0 = time.perf_counter()
        done = 0
        while done < t_steps:
            res = pr.gather()
            done += tc
        _sync(device)
        el = time.perf_counter() - t0
        assert res["obs"].shape[0] == b * world_size
        del res, pr
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        gathered = step_bytes * done
        out[name] = {"ms_per_100_step_rollout": el * 1e3 * t_steps / done, "steps_per_chunk": tc, "envs_per_gpu": b,
                     "collectives_per_chunk": 1, "gathered_GB_per_100_steps": step_bytes * t_steps / 1e9,
                     "GBps_received_per_gpu": gathered * (world_size - 1) / world_size / el / 1e9}
    return out


def _sync(device):
    import torch

    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    maybe_spawn(args)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_size} ranks")
    from vectorizedmultiagentsimulator_amd.shard import EnvShard, PackedRollout, max_over_ranks

    dist = None
    if args.dry_run:
        device = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)
        ranks_seen = dist.get_world_size()
        t = torch.ones(1, device=device)
        dist.all_reduce(t)  # every rank really is in the group
        assert int(t.item()) == world_size == ranks_seen

    # weak scaling: the global batch is world_size x num_envs, sharded by environment; every rank steps its own
    # contiguous block, no collective on the step path
    shard = EnvShard(world_size * args.num_envs, rank, world_size)
    assert shard.local_envs == args.num_envs

    gather = None
    if world_size > 1 and not args.no_gather:
        gather = time_rollout_gather(dist, EnvShard, PackedRollout, rank, world_size, device,
                                     args.num_envs if not args.dry_run else 64, t_steps=100 if not args.dry_run else 4,
                                     shrink=512 if args.dry_run else 1)

    if args.dry_run:
        if rank == 0:
            print(json.dumps({"metric": "env-steps/sec (batch x substeps) on 'balance'", "value": None, "dry_run": True,
                              "n_gpus": world_size, "ranks_seen": ranks_seen, "backend": "gloo",
                              "shard": [shard.lo, shard.hi], "rollout_gather": gather}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    sc, w = build_world(shard.local_envs, device, args.n_agents, args.lanes, seed=shard.seed(0))
    be = w._get_backend()
    be.set_queues(args.queues)
    state0 = w._state.clone()
    actions = make_actions(EPISODE, args.n_agents, args.num_envs, 1234 + rank)
    forces = pack_forces(w, actions, device)
    stream = torch.cuda.current_stream()

    def run(n_steps, start=0, fused=args.fused):
        done = 0
        while done < n_steps:
            k = (start + done) % EPISODE
            if k == 0:
                w._state.copy_(state0)
            chunk = min(EPISODE - k, n_steps - done)
            (be.rollout if fused else be.step_n)(chunk, forces[k: k + chunk])
            done += chunk

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        """K steps between fences: (HIP-event seconds, wall seconds), each MAX over ranks."""
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        fn(n)
        ev1.record(stream)
        fence()
        wall = time.perf_counter() - t0
        own = ev0.elapsed_time(ev1) * 1e-3
        return max_over_ranks(own, device), max_over_ranks(wall, device), own

    # ---- secondary legs first: they also bring the GPU to its steady clocks before the headline region
    persistent = env_leg = None
    if not args.no_fused and not args.fused:
        run(EPISODE, fused=True)
        ev_s, wall_s, _ = timed(lambda n: run(n, fused=True), args.steps)
        persistent = {
            "value": world_size * args.num_envs * w.substeps * args.steps / ev_s, "unit": "env-steps/s",
            "us_per_step": ev_s / args.steps * 1e6,
            "note": "vmas_world_rollout: identical results bit for bit, state stays in LDS between steps (scripted / "
                    "pre-computed forces only); NOT the headline value",
        }
        try:
            from vectorizedmultiagentsimulator_amd.environment import make_env

            env = make_env("balance", num_envs=args.num_envs, device=device, seed=0, validate_actions=False,
                           n_agents=args.n_agents)
            acts = [env.get_random_action(a) for a in env.agents]

            def env_steps(n):
                for _ in range(n):
                    env.step(acts)

            env_steps(400)  # (the first few hundred steps carry one-time costs)
            n_env = max(min(args.steps, 2000), 200)
            ev_s, wall_s, _ = timed(env_steps, n_env)
            env.bind(acts)  # caller-owned action tensors, static outputs: one foreign call per step

            def env_steps_bound(n):
                for _ in range(n):
                    env.step_bound()

            env_steps_bound(100)
            ev_b, wall_b, _ = timed(env_steps_bound, n_env)
            per_env = be.step_bytes_per_env() + POST_BYTES_PER_ENV
            env_leg = {
                "value": world_size * args.num_envs * w.substeps * n_env / wall_s, "unit": "env-steps/s",
                "us_per_step": wall_s / n_env * 1e6, "gpu_us_per_step": ev_s / n_env * 1e6, "steps": n_env,
                "launches_per_step": 1 if env._one_launch else None,
                "bound": {"value": world_size * args.num_envs * w.substeps * n_env / wall_b, "us_per_step": wall_b / n_env * 1e6,
                          "gpu_us_per_step": ev_b / n_env * 1e6,
                          "note": "Environment.bind(actions) + step_bound(): the same step on caller-owned action tensors and "
                                  "static output buffers - a single foreign call per step, no host-side tensor work"},
                "roofline": {"bound": "hbm", "achieved": per_env * args.num_envs / (ev_s / n_env) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": per_env * args.num_envs / (ev_s / n_env) / 1e9 / HBM_PEAK_GBS,
                             "bytes_per_env": per_env},
                "note": "make_env('balance').step() driven from Python, fresh output tensors every step: action ingest "
                        "(prologue) + World.step + reward/observation/done/info (epilogue) in ONE kernel launch "
                        "(vmas_world_step_env); value from the wall clock (host included), roofline from HIP events",
            }
            del env
        except Exception as e:  # never let a secondary leg break the bench line
            env_leg = {"error": repr(e)}

    # ---- the same K launches on ONE queue (what rocprofv3's per-kernel durations describe); then the headline
    single = None
    if not args.fused and be.queues(min(EPISODE, args.steps)) > 1:
        be.set_queues(1)
        run(args.warmup)
        ev_1, wall_1, _ = timed(lambda n: run(n, start=args.warmup), args.steps)
        single = {"value": world_size * args.num_envs * w.substeps * args.steps / ev_1, "unit": "env-steps/s",
                  "us_per_step": ev_1 / args.steps * 1e6,
                  "note": "vmas_world_set_queues(1): one launch per step on one HIP queue; its time per step is the "
                          "kernel's launch-to-launch time and agrees with rocprofv3's per-kernel duration + launch gap"}
        be.set_queues(args.queues)
    # ---- headline: W warm-up steps, then exactly K World.step launches between fences - R times over (every window is a
    #      complete measurement by the contract; `value` is the MEDIAN window, min / max beside it).  In front of the W
    #      steps, untimed: `clock_warmup_s` seconds of the same launches - a process that has just been set up finds the
    #      GPU's clocks ramping (the first ~0.1 s of kernels reads up to several times slow: profiles/README.md), and with
    #      the driver's small K every window would fall inside that ramp.
    t_clock = time.perf_counter()
    while time.perf_counter() - t_clock < args.clock_warmup:
        run(EPISODE)
        torch.cuda.synchronize()
    run(args.warmup)
    windows = [timed(lambda n: run(n, start=args.warmup), args.steps) for _ in range(max(1, args.repeats))]
    order = sorted(range(len(windows)), key=lambda i: windows[i][0])
    ev_s, wall_s, _ = windows[order[len(order) // 2]]
    kernel_s = ev_s / args.steps
    n_queues = 1 if args.fused else be.queues(min(EPISODE, args.steps))
    # every rank's own time per step (the headline takes the slowest rank, window by window)
    per_rank_us = [ev_s / args.steps * 1e6]
    if dist is not None:
        mine = torch.tensor([windows[order[len(order) // 2]][2]], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(allr, mine)
        per_rank_us = [float(x.item()) / args.steps * 1e6 for x in allr]

    if rank == 0:
        bytes_per_env = be.step_bytes_per_env()
        ach = bytes_per_env * args.num_envs / kernel_s / 1e9
        gflops = FLOP_PER_ENV_STEP * args.num_envs / kernel_s / 1e9
        out = {
            "metric": "env-steps/sec (batch x substeps) on 'balance'",
            "value": world_size * args.num_envs * w.substeps * args.steps / ev_s,
            "unit": "env-steps/s",
            "n_gpus": world_size,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup, "clock_warmup_s": args.clock_warmup,
            "ms_per_step": kernel_s * 1e3,
            "repeats": {"windows": len(windows), "steps_per_window": args.steps,
                        "ms_per_step_min": min(w_[0] for w_ in windows) / args.steps * 1e3,
                        "ms_per_step_median": kernel_s * 1e3,
                        "ms_per_step_max": max(w_[0] for w_ in windows) / args.steps * 1e3,
                        "note": "each window = exactly `steps` launches between barrier + synchronize fences; `value` and "
                                "`ms_per_step` are the median window"},
            "per_rank_us_per_step": per_rank_us,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "value_is": "World.step() physics (north-star hot path), timed with HIP events inside the fences; "
                        "`env_step` = through Environment.step(); `wall` = the same K steps by the host clock",
            "wall": {"ms_per_step": wall_s / args.steps * 1e3,
                     "value": world_size * args.num_envs * w.substeps * args.steps / wall_s},
            "config": {
                "workload": f"balance n_agents={args.n_agents}, {args.num_envs} envs/GPU, World.step() physics only, "
                            f"random actions, {EPISODE}-step episodes",
                "num_envs_per_gpu": args.num_envs,
                "global_envs": world_size * args.num_envs,
                "substeps": w.substeps,
                "lanes_per_env": be.lanes_per_env,
                "launch": "persistent rollout (vmas_world_rollout)" if args.fused else (
                    "one launch per step" if n_queues == 1 else
                    f"one launch per step and per part of the batch: {n_queues} HIP queues, {n_queues} launches per "
                    f"World.step of the whole batch (vmas_world_step_n, environments are independent)"),
                "queues": n_queues,
                "kernel": ("step_kernel_spec<SpecBalance4>: the world-specialised form of the step kernel (schedule as "
                           "compile-time tables, generated from the library's planner; bitwise the interpreter's results)")
                if be.specialized else "step_kernel (interpreter of the schedule)",
                "parallelism": f"env-sharded x{world_size}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": "step_kernel_spec" if be.specialized else "step_kernel",
                "kernel_us": kernel_s * 1e6,
                "kernel_us_is": "time per World.step of the whole batch from the HIP events (region / K)" + (
                    "" if n_queues == 1 else f"; {n_queues} launches of {args.num_envs // n_queues} environments each "
                    "overlap in it - rocprofv3's per-launch durations add up to more than this, see `single_queue`"),
                "bytes_per_launch": bytes_per_env * args.num_envs // n_queues,
                "launches_per_step": n_queues,
                "gflops": gflops,
                "gflops_frac_of_fp32_vector_peak": gflops / FP32_PEAK_GFLOPS,
                "binds": "neither roof at this batch: one launch moves 12.6 MB (1.6 us at 8 TB/s) and 56 Mflop "
                         "(0.4 us at fp32 peak); the launch is a ~3 us launch floor plus one tile's dependent chain "
                         "(profiles/r02e_balance32768_physics_pmc_summary.txt, r02_balance32768_phase_trace.txt). At "
                         "1 M environments the same kernel reaches 50 % of the HBM roof. HBM is the nominal bound.",
            },
        }
        if single is not None:
            single["roofline_frac"] = bytes_per_env * args.num_envs / (single["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["single_queue"] = single
        if env_leg is not None:
            out["env_step"] = env_leg
        if persistent is not None:
            out["persistent_rollout"] = persistent
        if gather is not None:
            out["rollout_gather"] = gather
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 PMC passes (scripts/gpu_prof.sh)
            try:
                tr = json.load(open(tpath))
                if tr.get("num_envs") == args.num_envs:
                    out["roofline"]["traffic"] = tr["bytes_per_launch"]
                    out["roofline"]["traffic_source"] = ("from profiles/latest_traffic.json (rocprofv3 PMC passes of this kernel, "
                                                         "scripts/gpu_prof.sh) - NOT measured in this run")
                    out["roofline"]["traffic_note"] = tr["note"]
            except Exception:
                pass
        if world_size == 1 and not args.no_cpu_baseline:
            st0, f_cpu = state0.cpu().numpy(), forces.cpu().numpy()
            try:
                out["cpu_baseline"] = cpu_reference(w, actions, st0, args.n_agents)
            except Exception as e:  # the reference is not importable here: say so, keep the port
                out["cpu_baseline_error"] = repr(e)
            out["cpu_port"] = cpu_port(w, f_cpu, st0)
            if "cpu_baseline" not in out:
                out["cpu_baseline"] = dict(out["cpu_port"])
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["cpu_port"]["gpu_over_cpu"] = out["value"] / out["cpu_port"]["value"]
            if env_leg and "value" in env_leg and "env_step" in out["cpu_baseline"]:
                out["cpu_baseline"]["env_step"]["gpu_over_cpu"] = env_leg["value"] / out["cpu_baseline"]["env_step"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
