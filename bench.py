#!/usr/bin/env python
"""bench.py - BASELINE.json's metric on the MI355X-native hot path, one JSON line per run.

usage  : python bench.py [--config balance|transport|transport_2pkg|navigation|football] [--gpus N] [--strong|--weak]
                         [--steps K] [--warmup W]            (no flags: config 2 - balance, 32768 envs, n_agents=4 - on 1 GPU)
metric : env-steps/sec (batch x substeps) on the configuration's scenario (SURVEY.md 8d's table: cfg 2 balance 32768 envs,
         cfg 3 transport 16384 (+ the n_packages=2 box-box variant), cfg 4 navigation n_agents=8 65536, cfg 5 football 5v5 131072)
step   : ONE call of the REFERENCE's ``Environment.step`` (environment.py:325) - the object ``vmas.make_env(...,
         device="cuda")`` returns, after ``attach()`` with its defaults - over the whole batch: action ingest + World.step
         (core.py:1972-2015) + reward / observation / done / info in ONE kernel launch for the benchmark scenarios, the
         reference's action asserts kept, the reference's batch-global broad phase (lazy form, inside the launch).  This is
         north_star's own sentence ("through vmas.make_env(...).step") and what `value` reports since round 6; when the
         reference is not importable the line falls back to the next object.  `world_step` = World.step() physics alone (the
         headline of rounds 1-5: one launch per step, pre-recorded agent forces resident in HBM), `environment_step` = this
         package's own Environment.step (same kernel, native host objects) with its GPU-bound form (`bound`) and the
         K-steps-per-launch form (`rollout`) beside it.
timing : W warm-up steps, then exactly K steps between barrier+synchronize pairs, `--repeats` windows, median window, MAX
         over ranks.  The headline is rated by the WALL clock between the fences (the host that drives the reference's
         Python objects is part of that path); HIP events around the same K calls are kept beside it.  `world_step` is timed
         with HIP events recorded on the launch stream right inside the fences (behind a ~100 us untimed spin kernel that
         lets the host queue the launches up: K back-to-back steps, not the host's first-launch latency on an idle GPU) with
         the wall clock beside them (`world_step.wall`).
inputs : SURVEY.md 8d's protocol - per step and policy agent u ~ U(-u_range, u_range), pre-generated on the host with
         torch.Generator().manual_seed(1234 + rank) (the law of Environment.get_random_action, environment.py:536-548);
         episodes are 100 steps long: every 100 steps the post-reset state is restored by a device-to-device copy inside
         the timed region.  World.step's agent forces are what Environment._set_action + process_action made of those
         actions (recorded from one real episode, scripted agents included).
multi-GPU: `--gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
         (one per GPU, backend nccl = RCCL); under a launcher it reads RANK/LOCAL_RANK/WORLD_SIZE.  The batch is sharded by
         environment - `--weak` (default for balance: the configuration's batch PER GPU) or `--strong` (default for the
         configurations BASELINE.json quotes as "sharded across 8": the configuration's batch in total) - NO collective on
         the step path; the pipeline's only exchange, the end-of-rollout all-gather, is timed on the REAL
         sharded rollout (`sharded_rollout`: Environment.rollout writing K steps per launch straight into the buffer that
         ONE all_gather_into_tensor sends) and on the three BASELINE shapes (`rollout_gather`); neither is part of `value`.
roofline: the headline's dominant kernel = the one-launch Environment.step kernel: achieved = algorithmic bytes per launch
         (24 E + 12 A read, 24 E_dyn written per environment for the physics, + the actions read and the
         observations / rewards / done / info written: 657 B per environment for balance) / the launch-to-launch time of
         back-to-back env.step calls from HIP events (asserts off, so that nothing but the kernel is between the events).
         `traffic` = HBM bytes per launch by the PMC counters: the same env.step calls re-run (N = 1 only, two short child
         processes) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` - separate passes, counters only, FETCH_SIZE
         doubled (gfx950) - median per dispatch; null with the reason if rocprofv3 is not there (`--no-traffic` skips it).
         `world_step.roofline`: the same for the physics-only launch.
cpu_baseline: the REFERENCE itself (VMAS, `kind: "reference"`): its Environment.step (environment.py:325) on the host
         cores, device="cpu", torch threads = the fastest count on this host, same initial state and actions, bounded to
         ~10 s; `value` = those Environment.step calls, `world_step` = its World.step (core.py:1972) share of them (timed inside
         the same calls).  EVERY one of those reference steps is also a parity sample of this run (`parity`).  Rank 0, N=1
         only; configurations above 32768 environments are timed on the first 32768 (the reference's rate is flat in the
         batch there).  The reference is imported from /root/reference when present, else from its
         byte-compiled build oracle/_ref (made by __graft_entry__.build()).  `cpu_port` = the C oracle with OpenMP.
attached_reference: the drop-in boundary itself - attach(vmas.make_env(..., device="cuda")): the reference's
         own Environment on the GPU with its World.step rebound to the native kernel.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EPISODE = 100
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_GFLOPS = 157286.4  # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz (vector fp32, no MFMA)

# the configurations, their algorithmic flop per env-step and the bytes a fused post-step adds
CONFIGS = {
    "balance": dict(cfg=2, scenario="balance", envs=32768, kwargs=dict(n_agents=4), flop=1700.0, post_bytes=273, scaling="weak"),
    "transport": dict(cfg=3, scenario="transport", envs=16384, kwargs={}, flop=800.0, post_bytes=193, scaling="weak"),
    "transport_2pkg": dict(cfg=3, scenario="transport", envs=16384, kwargs=dict(n_packages=2), flop=4700.0, post_bytes=305,
                           scaling="weak"),
    "navigation": dict(cfg=4, scenario="navigation", envs=65536, kwargs=dict(n_agents=8), flop=1800.0, post_bytes=609,
                       env_flop=28800.0, scaling="strong"),
    "football": dict(cfg=5, scenario="football", envs=131072,
                     kwargs=dict(n_blue_agents=5, n_red_agents=5, ai_red_agents=False), flop=11900.0, post_bytes=3561,
                     scaling="strong"),
}
GATHER_SHAPES = {"balance": (4, 16), "transport": (4, 11), "transport_2pkg": (4, 18), "navigation": (8, 18), "football": (10, 88)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="balance", choices=sorted(CONFIGS))
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--strong", action="store_true", help="N GPUs share the configuration's batch (default for navigation / football)")
    ap.add_argument("--weak", action="store_true", help="every GPU steps the configuration's batch (default for balance / transport)")
    ap.add_argument("--clock-warmup", type=float, default=0.25,
                    help="seconds of untimed launches before the W warm-up steps (the GPU's clocks ramp up: see bench.py)")
    ap.add_argument("--gate-us", type=float, default=100.0,
                    help="untimed spin kernel in front of every timed window (microseconds; 0 = none): the K launches are "
                         "queued behind it, so the HIP events hold K back-to-back steps and not the host's first-launch latency")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step window is timed this many times (each between its own fences); `value` = the median window")
    ap.add_argument("--num-envs", type=int, default=0, help="environments PER GPU (0 = the configuration's)")
    ap.add_argument("--n-agents", type=int, default=0, help="balance / navigation: agents (0 = the configuration's)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per environment (0 = library default)")
    ap.add_argument("--queues", type=int, default=0, help="HIP queues of vmas_world_step_n (0 = library's choice, 1..4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary measurements (environment_step, persistent rollout)")
    ap.add_argument("--no-gather", action="store_true", help="skip the rollout all-gather timings (N > 1)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (balance, 1 GPU) only: skip the short lines of the other configurations")
    ap.add_argument("--no-attached", action="store_true", help="skip the attached_reference leg")
    ap.add_argument("--fused", action="store_true",
                    help="time vmas_world_rollout (persistent launch, state resident in LDS) instead of one launch per step")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip `roofline.traffic` (HBM bytes per launch by the PMC counters: two short re-runs of the physics "
                         "launches under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, N = 1 only)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # (the re-run itself: launches only)
    ap.add_argument("--traffic-attached", action="store_true", help=argparse.SUPPRESS)  # (... of the attached reference's env.step)
    ap.add_argument("--share-gpu", action="store_true",
                    help="TESTING ONLY: every rank on cuda:0, process group over gloo - runs the N > 1 code path (sharding, fences, "
                         "max-over-ranks, the attached headline on every rank) on a one-GPU box; the line says so and its value means nothing")
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
    if not args.dry_run and not args.share_gpu:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible: refusing to report an "
                     f"n_gpus={args.gpus} line from fewer devices")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def config_kwargs(name, n_agents=0):
    kw = dict(CONFIGS[name]["kwargs"])
    if n_agents and "n_agents" in kw:
        kw["n_agents"] = n_agents
    return kw


def make_actions(env, n_steps, seed):
    """[n_steps, A, B, size] ~ U(-u_range, u_range) on the host (the reference's get_random_action law, environment.py:536-548)."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed)
    A, B = len(env.agents), env.num_envs
    size = env.agents[0].action_size
    assert all(a.action_size == size for a in env.agents)
    u = torch.rand(n_steps, A, B, size, generator=g) * 2 - 1
    rng = torch.tensor([[float(x) for x in (a.action.u_range if isinstance(a.action.u_range, (list, tuple)) else [a.action.u_range] * size)]
                        for a in env.agents])
    return u * rng[None, :, None, :]


def record_episode_forces(env, acts_dev):
    """[EPISODE, A_all, 3, ld]: the agent-force rows Environment._set_action + process_action (scripted agents included)
    leave for World.step, step by step, over one real episode from the current state; the state is restored afterwards."""
    import torch

    snap = env.get_state()
    rows = []
    for k in range(acts_dev.shape[0]):
        env.step(list(acts_dev[k].unbind(0)))
        rows.append(env.world._packed_agent_ft().clone())
    env.set_state(snap)
    return torch.stack(rows).contiguous()


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_port(w, forces_cpu, state0_cpu, budget_s=6.0):
    """C oracle (scalar port of the reference algorithm, OpenMP over environments) on the host cores."""
    from oracle.oracle import Oracle

    o = Oracle(w.spec)
    B = state0_cpu.shape[-1]
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
            "sample": f"{n} World.step() x {B} envs (same state/forces as the GPU run), {el:.1f} s, C oracle with OpenMP over environments"}


def _pack_ref_state(world):
    """[E, 6, B] of a reference world (pos.x pos.y vel.x vel.y rot ang_vel per entity: the packed layout, host side)."""
    import torch

    return torch.stack([torch.cat([e.state.pos, e.state.vel, e.state.rot, e.state.ang_vel], dim=-1).T for e in world.entities]).clone()


def _pack_ref_ft(world):
    import torch

    return torch.stack([torch.cat([a.state.force, a.state.torque], dim=-1).T for a in world.agents]).clone()


def _err_stats(got, want, tol=1e-5):
    """max |got - want| over the values that are finite and of physical magnitude on both sides (a blown-up environment -
    the reference itself reaches 1e24 in dense soups - is counted, not compared), and how many exceed tol abs + tol rel."""
    import torch

    got, want = got.double(), want.double()
    sane = torch.isfinite(got) & torch.isfinite(want) & (want.abs() < 1e3) & (got.abs() < 1e3)
    err = (got - want).abs()[sane]
    beyond = int((err > tol + tol * want.abs()[sane]).sum())
    return (float(err.max()) if err.numel() else 0.0), beyond, int(sane.numel() - sane.sum()), int(sane.sum())


def cpu_reference(name, kw, w, actions, state0_cpu, budget_s=10.0, max_envs=32768, threads=None, parity=None):
    """The reference's own Environment.step - and, inside it, its World.step - on the host cores (device="cpu").
    ``parity`` (the GPU run's environment, its post-reset snapshot and device actions): the SAME reference steps are the
    checker of the native path in this very run (BASELINE.md section 3 "parity in the same run") - every 10th step's
    pre-step state and agent forces are kept (outside the timers), replayed as ONE native World.step each and compared
    with what the reference's step made of them (teacher-forced); and the native environment is run free from the same
    start with the same actions for as many steps as the reference made, its drift reported separately."""
    import torch
    from oracle import ref

    B = min(w.batch_dim, max_envs)
    ncpu = os.cpu_count() or 1
    sc = CONFIGS[name]["scenario"]
    env = ref.make_env(sc, num_envs=B, device="cpu", seed=0, continuous_actions=True, **kw)
    world = env.world
    assert [e.name for e in world.entities] == [e.name for e in w.entities], "entity order differs from the reference's"
    st0 = torch.from_numpy(state0_cpu)

    def restore():
        for i, e in enumerate(world.entities):  # the GPU run's post-reset state, through the reference's own setters
            e.set_pos(st0[i, 0:2, :B].T.clone(), batch_index=None)
            e.set_vel(st0[i, 2:4, :B].T.clone(), batch_index=None)
            e.set_rot(st0[i, 4:5, :B].T.clone(), batch_index=None)
            e.set_ang_vel(st0[i, 5:6, :B].T.clone(), batch_index=None)

    in_world = [0.0]
    hook_s = [0.0]
    samples, sample_now = [], [False]
    orig_step = world.step

    def timed_world_step():
        if sample_now[0]:
            th = time.perf_counter()
            # (Environment.step between two World.step calls writes forces only: the state entering this step is the one the
            #  previous sample left - kept once)
            pre = _pack_ref_state(world) if not samples else None
            ft = _pack_ref_ft(world)
            hook_s[0] += time.perf_counter() - th
        t0 = time.perf_counter()
        orig_step()
        in_world[0] += time.perf_counter() - t0
        if sample_now[0]:
            th = time.perf_counter()
            samples.append((pre, ft, _pack_ref_state(world)))
            hook_s[0] += time.perf_counter() - th

    world.step = timed_world_step

    def env_step(k):
        env.step([actions[k % EPISODE, i, :B] for i in range(len(env.agents))])

    # torch's intra-op thread count that is actually fastest on this host: World.step is ~2800 small aten calls per step,
    # and os.cpu_count() threads (the survey's plan) can be pathological - 256 threads measured 58 s per step on a pool
    # box against 50 ms with 8.  Probed in ascending order, stopping once a count is clearly slower.
    probed, best = {}, (float("inf"), 1)
    with torch.no_grad():
        for th in (sorted({4, 8, 16, 32, 64, min(ncpu, 128), ncpu}) if threads is None else [threads]):
            if th > ncpu:
                continue
            torch.set_num_threads(th)
            restore()
            env_step(0)
            t0 = time.perf_counter()
            env_step(1)
            dt = time.perf_counter() - t0
            probed[th] = round(dt * 1e3, 2)
            if dt < best[0]:
                best = (dt, th)
            elif dt > 1.3 * best[0]:
                break
        threads = best[1]
        torch.set_num_threads(threads)
        restore()
        for k in range(2):
            env_step(k)
        restore()
        in_world[0] = hook_s[0] = 0.0
        n, t0 = 0, time.perf_counter()
        while True:
            sample_now[0] = parity is not None  # (EVERY reference step is a parity sample: round-5 review)
            env_step(n)
            n += 1
            el = time.perf_counter() - t0 - hook_s[0]
            if (el > budget_s and n >= 2) or n >= EPISODE:
                break
        sample_now[0] = False
        final_ref = _pack_ref_state(world)
    sub = w.substeps
    par = None
    if parity is not None:
        try:
            par = same_run_parity(parity, w, B, samples, final_ref, n)
        except Exception as e:  # noqa: BLE001 (a parity leg that cannot run says so; it never breaks the line)
            par = {"error": repr(e)[:300]}
    return {
        "parity": par,
        "value": B * n * sub / in_world[0], "unit": "env-steps/s", "cores": threads, "kind": "reference",
        "torch_threads": torch.get_num_threads(), "host_cpus": ncpu, "envs": B,
        "ms_per_env_step_by_threads": probed,
        "sample": f"{n} Environment.step() (vmas/simulator/environment/environment.py:325) of the reference's {sc} {kw} x {B} envs on "
                  f"device='cpu', same initial state and actions as the GPU run, {el:.1f} s ({el / n * 1e3:.1f} ms/step), of which "
                  f"{in_world[0]:.1f} s inside World.step (core.py:1972; {in_world[0] / n * 1e3:.1f} ms/step = `value`); 2 warm-up steps",
        "env_step": {"value": B * n * sub / el, "ms_per_step": el / n * 1e3, "steps": n, "seconds": el, "unit": "env-steps/s",
                     "world_step_share": in_world[0] / el,
                     "note": "the reference's Environment.step (environment.py:325): ingest + World.step + reward/obs/done"},
        "reference_from": ref.root(),
    }


def same_run_parity(ctx, w, B, samples, final_ref, n_ref_steps):
    """The native path checked against the reference steps that were just timed (see cpu_reference).  Teacher-forced: each
    sampled pre-step state + agent forces -> ONE native World.step (the product's launch, the product's broad phase) ->
    compared with the reference's post-step state.  Free-running: the native Environment from the same post-reset state
    with the same actions, compared after the number of steps the reference made."""
    import torch

    genv, snapshot, acts_dev = ctx["env"], ctx["snapshot"], ctx["acts_dev"]
    be = w._get_backend()
    st, ft = w._packed_state(), w._packed_agent_ft()
    nA = len(w.agents)
    worst, beyond, blown, values = 0.0, 0, 0, 0
    ctl_worst, ctl_beyond, ctl_steps = 0.0, 0, 0
    last_post = None
    for pre, f, post in samples:
        pre = last_post if pre is None else pre
        last_post = post
        for form in ("product", "per_environment"):
            # The rule is batch-GLOBAL: the native batch must hold exactly the environments the reference's batch held.  Where
            # the reference was run on the first B of a larger batch (configurations above 32 768 / the short lines' 8 192) the
            # sample is TILED over the rest - the same set of environments, so the same pairs are on for the batch.  (Until
            # round 6 the rest kept its post-reset state: its overlaps switched pairs on that the reference's batch had off.)
            genv.set_state(snapshot)
            pre_d, f_d = pre.to(st.device), f.to(st.device)
            for lo in range(0, w.batch_dim, B):
                n_ = min(B, w.batch_dim - lo)
                st[:, :, lo:lo + n_].copy_(pre_d[:, :, :n_])
                ft[:nA, :, lo:lo + n_].copy_(f_d[:, :, :n_])
            w.invalidate_queries()
            if form == "product":  # what World.step() of this environment runs: the reference's batch-global rule
                assert w.exact_broad_phase
                be.step_exact()
                e, b, bl, v = _err_stats(st[:, :, :B].cpu(), post)
                worst, beyond, blown, values = max(worst, e), beyond + b, blown + bl, values + v
            else:  # the NEGATIVE control: every static pair per environment (the default above 1 024 environments until round 5)
                be.step()
                e, b, _, _ = _err_stats(st[:, :, :B].cpu(), post)
                ctl_worst, ctl_beyond, ctl_steps = max(ctl_worst, e), ctl_beyond + b, ctl_steps + (1 if b else 0)
    # free-running: the native Environment.step (ingest + physics + post-step in one launch) from the same start
    genv.set_state(snapshot)
    for k in range(n_ref_steps):
        genv.step(list(acts_dev[k % EPISODE].unbind(0)))
    got = w._packed_state()[:, :, :B].cpu()
    d = (got.double() - final_ref.double()).abs()
    sane = torch.isfinite(d) & (final_ref.abs() < 1e3)
    d = d[sane]
    genv.set_state(snapshot)
    return {
        "teacher_forced_max_abs": worst, "values_beyond_1e-5": beyond, "values_compared": values, "blown_up_values_skipped": blown,
        "teacher_forced_steps": len(samples), "envs": B,
        "broad_phase": "the reference's batch-global rule (World.collides core.py:2788-2803), form %d of vmas_world_exact_form" % be.exact_form(),
        "per_environment_form_control": {"values_beyond_1e-5": ctl_beyond, "teacher_forced_max_abs": ctl_worst,
                                         "steps_with_a_value_beyond": ctl_steps,
                                         "note": "the same samples stepped with every static pair evaluated per environment - the "
                                                 "default of rounds 1-5 above 1 024 environments: what `values_beyond_1e-5` would "
                                                 "read without the rule"},
        "free_running_drift": {"steps": n_ref_steps, "max_abs": float(d.max()) if d.numel() else 0.0,
                               "median_abs": float(d.median()) if d.numel() else 0.0,
                               "frac_beyond_1e-3": float((d > 1e-3).double().mean()) if d.numel() else 0.0},
        "note": "same run, same reference steps as `cpu_baseline`: EVERY reference World.step replayed as one native "
                "World.step from the reference's pre-step state and forces (teacher-forced; tolerance 1e-5 abs + 1e-5 rel, "
                "north_star); free_running = native Environment.step from the same start and actions, |state - reference's| after "
                "`steps` steps (two fp32 implementations of a chaotic contact system: the reference against itself with "
                "another thread count drifts alike, SURVEY.md App. C-2)",
    }


# ------------------------------------------------------------------------------------------------ gather
def time_rollout_gather(dist, shard_cls, packed_cls, rank, world_size, device, per_gpu_envs, t_steps=100,
                        max_chunk_bytes=2 << 30, shrink=1):
    """End-of-rollout all-gather for the shapes of BASELINE configs 2 / 4 / 5 on synthetic buffers: every
    rank's rollout chunk is ONE packed buffer and the exchange ONE all_gather_into_tensor per chunk, straight into the
    global buffer.  Chunks of steps keep the gathered buffer below ``max_chunk_bytes`` (config 5: 46 GB per 100-step
    rollout on 8 GPUs).  (`sharded_rollout` times the same collective on a real rollout.)"""
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


def sharded_rollout_leg(env, shard, acts_dev, snapshot, dist, device, t_steps=EPISODE, max_chunk_bytes=2 << 30):
    """The pipeline SURVEY.md 8e describes, for real: this rank's shard of the batch is rolled out K steps per launch
    (Environment.rollout, pre-computed actions) with the kernel storing straight into the buffer that ONE
    all_gather_into_tensor then sends (shard.NativeRollout) - chunk after chunk until `t_steps` steps are done.  Times, MAX
    over ranks: the rollout launches and the collectives by HIP events on the stream both run on, and the wall clock of
    the whole pipeline."""
    import torch
    from vectorizedmultiagentsimulator_amd.rollout import collect_native
    from vectorizedmultiagentsimulator_amd.shard import NativeRollout, max_over_ranks

    if not (env._one_launch and getattr(env._post, "rollout_ok", True)):
        return {"skipped": "this configuration's Environment.step is not one launch per step on this shard size "
                           "(navigation above one tile per CU: its collision penalties reduce over the batch per step)"}
    ws = shard.world_size
    per_step = sum(math.prod(s) * torch.empty((), dtype=d).element_size() for _, s, d in env.rollout_fields(1))
    tc = max(1, min(t_steps, max_chunk_bytes // max(per_step * ws, 1)))
    while t_steps % tc:
        tc -= 1
    nr = NativeRollout.for_env(shard, env, tc)
    acts = [acts_dev[:tc, i].contiguous() for i in range(acts_dev.shape[1])]

    def pipeline():
        env.set_state(snapshot)
        t_roll = t_gath = 0.0
        pend = []
        for _ in range(t_steps // tc):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            collect_native(env, acts, shard, into=nr)
            ev[1].record()
            g = nr.gather()
            ev[2].record()
            pend.append(ev)
        torch.cuda.synchronize()
        for e in pend:
            t_roll += e[0].elapsed_time(e[1]) * 1e-3
            t_gath += e[1].elapsed_time(e[2]) * 1e-3
        return t_roll, t_gath, g

    pipeline()  # warm-up: communicator, the gathered buffer, one-time costs
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    t_roll, t_gath, g = pipeline()
    wall = time.perf_counter() - t0
    R = next(iter(g.values()))
    ranks_in_result = len(R) if isinstance(R, list) else R.shape[0]
    mine = t_roll / t_steps * 1e6
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor([mine], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(t) for _ in range(ws)]
        dist.all_gather(allr, t)
        per_rank = [float(x.item()) for x in allr]
    t_roll, t_gath, wall = (max_over_ranks(x, device) for x in (t_roll, t_gath, wall))
    gathered = nr.nbytes * ws * (t_steps // tc)  # bytes of the gathered buffers over the whole rollout
    return {
        "steps": t_steps, "steps_per_launch": tc, "envs_per_gpu": env.num_envs, "ranks_in_result": ranks_in_result,
        "rollout_us_per_step": t_roll / t_steps * 1e6, "per_rank_rollout_us_per_step": per_rank,
        "gather_ms_per_100_steps": t_gath * 1e3 * 100 / t_steps, "collectives": t_steps // tc,
        "gathered_GB_per_100_steps": gathered * 100 / t_steps / 1e9,
        "GBps_received_per_gpu": (gathered * (ws - 1) / ws / t_gath / 1e9) if (ws > 1 and t_gath > 0) else None,
        "pipeline_wall_ms_per_100_steps": wall * 1e3 * 100 / t_steps,
        "value": ws * env.num_envs * env.world.substeps * t_steps / wall, "unit": "env-steps/s",
        "note": "Environment.rollout (K Environment.step per launch, pre-computed actions) storing into shard.NativeRollout, "
                "then ONE all_gather_into_tensor of that buffer per chunk; value = all ranks' env-steps / pipeline wall time",
    }


def _sync(device):
    import torch

    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ attached reference
def attached_reference_leg(name, kw, B, device, n=300, brief=False):
    """attach(vmas.make_env(..., device='cuda')) - the reference's own Environment / Scenario / World objects on the GPU.
    `env_step_us`: the reference's ``env.step`` as `attach()` leaves it by default - for the four benchmark scenarios the
    one-launch kernel (ingest prologue + World.step + reward/observation/done/info epilogue, attached_env.py) with the
    reference's action asserts kept (one stream synchronisation per step); `env_step_deferred_validate_us`: the asserts
    deferred by one step (``validate_actions="deferred"``, no synchronisation); `env_step_no_validate_us`: none; `env_step_unfused_us`: ``attach(fused=False)`` - only World.step (and Lidar.measure) rebound,
    the reference's tensor-op ingest / reward / observation around it (what rounds 1-4 measured); `world_step_*`: the
    rebound seam alone."""
    import torch
    from oracle import ref  # (locates the reference package - /root/reference or its byte-compiled build; it is the HOST here)
    from vectorizedmultiagentsimulator_amd.adapter import attach

    sc = CONFIGS[name]["scenario"]
    env = ref.make_env(sc, num_envs=B, device=str(device), seed=0, continuous_actions=True, **kw)
    world = env.world
    # fresh random actions every step (a cycle of 25 pre-drawn sets: one held action drives every body into a wall and
    # measures the dense-contact case only)
    cycle = [[env.get_random_action(a) for a in env.agents] for _ in range(25)]
    acts = cycle[0]
    sub = int(getattr(world, "_substeps", 1))

    def time_env_steps(m, warm=20):
        env.reset(seed=0)  # every leg from the same post-reset distribution (the cost of a step depends on the state it finds)
        for k in range(warm):
            env.step(cycle[k % 25])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for k in range(m):
            env.step(cycle[k % 25])
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / m * 1e6, e0.elapsed_time(e1) / m * 1e3

    out = {"envs": B}
    with torch.no_grad():
        # (1) the default attach: env.step itself is the kernel where a post-step kernel covers the configuration
        h = attach(env, specialize=None)
        out["fused"] = h.fused is not None
        if h.fused is None:
            out["fused_reason"] = h.fused_reason
        else:
            out["launches_per_env_step"] = 1 if h.fused.one_launch else (2 if h.fused.ingest_in_step else 3)
        m = n if h.fused is not None else max(10, n // 10)
        out["env_step_us"], out["env_step_gpu_us"] = time_env_steps(m)
        out["kernel"] = "world-specialised" if h.backend.specialized else ("lane-compacted" if getattr(h.backend, "compact", False)
                                                                          else "schedule interpreter")
        out["exact_broad_phase"] = bool(h.exact_broad_phase)
        if not brief:
            # the rebound seam alone: forces as the last env.step left them
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.2:
                for _ in range(50):
                    world.step()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                world.step()
            t1 = time.perf_counter()
            e1.record()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.update({"world_step_us": (t2 - t0) / n * 1e6, "world_step_host_enqueue_us": (t1 - t0) / n * 1e6,
                        "world_step_gpu_us": e0.elapsed_time(e1) / n * 1e3, "steps": n,
                        "world_step_protocol": "world.step() alone, the agent forces HELD as the last env.step left them: over the "
                                               "300 calls every body is driven into a wall - the dense-contact case (football: "
                                               "2-3x the random-action step, see `environment_step.bound` of the native line); "
                                               "host_enqueue includes the queue's back-pressure once the GPU is the slower side"})
        out["refreshes"] = h.refreshes
        h.detach()
        if out["fused"]:
            # (2) the asserts deferred by one step (the step launch flags a bad action, the next env.step raises: no sync) ...
            h = attach(env, specialize=None, validate_actions="deferred")
            out["env_step_deferred_validate_us"], out["env_step_deferred_validate_gpu_us"] = time_env_steps(m)
            h.detach()
            # ... and without them
            h = attach(env, specialize=None, validate_actions=False)
            out["env_step_no_validate_us"], out["env_step_no_validate_gpu_us"] = time_env_steps(m)
            # K steps of the reference's environment per launch (handle.fused.rollout: vmas_world_rollout_env on its objects)
            if h.fused.one_launch and getattr(h.fused.post, "rollout_ok", True):
                try:
                    K = 50
                    env.reset(seed=0)
                    racts = [torch.stack([cycle[k % 25][i] for k in range(K)]).contiguous() for i in range(len(env.agents))]
                    h.fused.rollout(racts)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        h.fused.rollout(racts)
                    e1.record()
                    torch.cuda.synchronize()
                    out["rollout_us_per_step"] = e0.elapsed_time(e1) / (4 * K) * 1e3
                except Exception as e:  # noqa: BLE001
                    out["rollout_error"] = repr(e)[:200]
            h.detach()
        if not brief:
            # (3) only the seam rebound: the reference's own Environment.step around the native World.step
            h = attach(env, specialize=None, fused=False)
            out["env_step_unfused_us"], _ = time_env_steps(max(10, n // 10), warm=3)
            h.detach()
    out["value"] = B * sub / (out["env_step_us"] * 1e-6)
    out["unit"] = "env-steps/s"
    out["value_is"] = "env-steps/s through the REFERENCE's env.step after attach() (defaults: fused where covered, action asserts kept)"
    if "env_step_no_validate_us" in out:
        out["value_no_validate"] = B * sub / (out["env_step_no_validate_us"] * 1e-6)
        out["value_deferred_validate"] = B * sub / (out["env_step_deferred_validate_us"] * 1e-6)
    return out


def attached_headline(name, kw, B, device, steps, warmup, repeats, clock_warmup, dist, seed=0):
    """THE HEADLINE (north_star; round-5 review item 7): `steps` calls of the REFERENCE's ``Environment.step`` - the object
    ``vmas.make_env(..., device='cuda')`` returns, after ``attach()`` with its defaults: the one-launch kernel where a post-step
    kernel covers the configuration, the reference's action asserts kept, the reference's batch-global broad phase - between
    fences (barrier + synchronize on both sides), `warmup` untimed calls first, `repeats` windows, every rank its own shard.
    Returns the windows as (HIP-event seconds, wall seconds), each MAX over ranks."""
    import torch
    from oracle import ref  # (locates the reference package - /root/reference or its byte-compiled build; it is the HOST here)
    from vectorizedmultiagentsimulator_amd.adapter import attach
    from vectorizedmultiagentsimulator_amd.shard import max_over_ranks

    sc = CONFIGS[name]["scenario"]
    env = ref.make_env(sc, num_envs=B, device=str(device), seed=seed, continuous_actions=True, **kw)
    cycle = [[env.get_random_action(a) for a in env.agents] for _ in range(25)]  # fresh random actions every step
    h = attach(env, specialize=None)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n, start=0):
        for k in range(n):
            env.step(cycle[(start + k) % 25])

    out = {"fused": h.fused is not None, "fused_reason": h.fused_reason, "exact_broad_phase": bool(h.exact_broad_phase),
           "exact_form": h.backend.exact_form(), "substeps": int(getattr(env.world, "_substeps", 1)),
           "kernel": "world-specialised" if h.backend.specialized else ("lane-compacted" if getattr(h.backend, "compact", False)
                                                                       else "schedule interpreter")}
    if h.fused is not None:
        out["launches_per_env_step"] = 1 if h.fused.one_launch else (2 if h.fused.ingest_in_step else 3)
    with torch.no_grad():
        env.reset(seed=seed)
        t_clock = time.perf_counter()
        while time.perf_counter() - t_clock < clock_warmup:
            run(50)
            torch.cuda.synchronize()
        run(warmup)
        windows = []
        for _ in range(max(1, repeats)):
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            run(steps, start=warmup)
            e1.record()
            fence()
            wall = time.perf_counter() - t0
            windows.append((max_over_ranks(e0.elapsed_time(e1) * 1e-3, device), max_over_ranks(wall, device)))
    h.detach()
    out["windows"] = windows
    return out


# ------------------------------------------------------------------------------------------------ HBM traffic (PMC)
def traffic_child(args, device):
    """What the counter passes profile: the headline's own launches (same world, same queues, recorded forces), 200 of them."""
    import torch
    from vectorizedmultiagentsimulator_amd.environment import make_env

    cfg = CONFIGS[args.config]
    B = args.num_envs or cfg["envs"]
    if args.traffic_attached:  # the north-star path: the REFERENCE's env.step after attach() - its one-launch step kernel
        from oracle import ref
        from vectorizedmultiagentsimulator_amd.adapter import attach

        renv = ref.make_env(cfg["scenario"], num_envs=B, device=str(device), seed=0, continuous_actions=True,
                            **config_kwargs(args.config, args.n_agents))
        attach(renv, specialize=None, validate_actions=False)
        cyc = [[renv.get_random_action(a) for a in renv.agents] for _ in range(25)]
        with torch.no_grad():
            for k in range(200):
                renv.step(cyc[k % 25])
        torch.cuda.synchronize()
        return
    env = make_env(cfg["scenario"], num_envs=B, device=device, seed=0, validate_actions=False, **config_kwargs(args.config, args.n_agents))
    be = env.world._get_backend()
    if args.lanes:
        be.set_lanes_per_env(args.lanes)
    be.set_queues(args.queues)
    acts = make_actions(env, 20, 1234).to(device)
    forces = record_episode_forces(env, acts)
    for _ in range(10):
        be.step_n(20, forces, exact=bool(env.world.exact_broad_phase))
    torch.cuda.synchronize()


def measure_traffic(args, n_launches_per_step, timeout_s=90, attached=False):
    """HBM bytes per launch of the dominant kernel by the PMC counters, collected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in separate `rocprofv3 --pmc` passes (they do not fit one), counters only (no trace domain
    beside them), FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes; confirmed in this library's own access
    pattern, profiles/r01_traffic_calibration.txt), both in KiB.  Returns (bytes per launch, details) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--config", args.config, "--queues", str(args.queues)]
    if args.num_envs:
        child += ["--num-envs", str(args.num_envs)]
    if args.n_agents:
        child += ["--n-agents", str(args.n_agents)]
    if args.lanes:
        child += ["--lanes", str(args.lanes)]
    if attached:
        child += ["--traffic-attached"]
    med = {}
    kernel = None
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            try:
                r = subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "p", "--"] + child,
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} timed out"
            vals = {}
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] == counter and ("step_kernel" in name or "vmas_rt_" in name or (
                            attached and ("post_kernel" in name or "collision_kernel" in name))):
                        vals.setdefault(name, []).append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no step-kernel dispatch in the {counter} pass (rc {r.returncode}): {r.stderr[-200:]}"
            name = max(vals, key=lambda n: len(vals[n]))  # the kernel of the timed launches (the recording steps are other forms)
            # (an env.step that is two kernels - football's step + post-step, navigation's step + collision kernel beyond one tile
            #  per CU: the step's traffic is the sum of both, each at its median)
            names = [n for n in vals if len(vals[n]) * 2 > len(vals[name])] if attached else [name]
            med[counter] = 0.0
            for n in names:
                v = sorted(vals[n][len(vals[n]) // 2:])  # (the second half: past the first touches of the buffers)
                med[counter] += v[len(v) // 2]
            kernel = " + ".join(n.split("(")[0][-80:] for n in names)
    traffic = (2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024.0
    return traffic, {"FETCH_SIZE_KiB_median": med["FETCH_SIZE"], "WRITE_SIZE_KiB_median": med["WRITE_SIZE"], "kernel": kernel,
                     "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}


# ------------------------------------------------------------------------------------------------ one configuration
def _library_build_id():
    """Digest of the sources libvmas_hip.so was built from (vmas_build_id, csrc/build.sh): ties a line to a commit's kernels."""
    try:
        from vectorizedmultiagentsimulator_amd import _abi as A
        return (A.load_library().vmas_build_id() or b"").decode()
    except Exception:  # noqa: BLE001 - a line without the id is still a line
        return None


def measure(name, args, device, shard, dist, rank, world_size, brief=False):
    """Everything bench.py measures on one configuration; `brief`: the short form the default run appends for the other
    configurations (physics + Environment.step, fewer steps, no CPU legs)."""
    import torch
    from vectorizedmultiagentsimulator_amd.environment import make_env
    from vectorizedmultiagentsimulator_amd.shard import max_over_ranks

    cfg = CONFIGS[name]
    kw = config_kwargs(name, args.n_agents)
    B = shard.local_envs
    steps = args.steps if not brief else min(args.steps, 300)
    warmup = args.warmup if not brief else min(args.warmup, 50)
    repeats = max(1, args.repeats if not brief else 3)
    env = make_env(cfg["scenario"], num_envs=B, device=device, seed=shard.seed(0), validate_actions=False, **kw)
    w = env.world
    be = w._get_backend()
    if args.lanes:
        be.set_lanes_per_env(args.lanes)
    be.set_queues(args.queues)
    snapshot = env.get_state()
    actions = make_actions(env, EPISODE, 1234 + rank)  # host
    acts_dev = actions.to(device)
    forces = record_episode_forces(env, acts_dev)
    state0 = w._packed_state().clone()
    stream = torch.cuda.current_stream()
    sub = w.substeps

    def run(n_steps, start=0, fused=args.fused):
        done = 0
        while done < n_steps:
            k = (start + done) % EPISODE
            if k == 0:
                w._state.copy_(state0)
            chunk = min(EPISODE - k, n_steps - done)
            (be.rollout if fused else be.step_n)(chunk, forces[k: k + chunk], exact=bool(w.exact_broad_phase))
            done += chunk

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    gate_cycles = 0
    if args.gate_us > 0:  # calibrate torch's spin kernel (its unit differs between builds) to ~gate_us microseconds
        try:
            torch.cuda._sleep(1000)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.cuda._sleep(100000)
            e1.record()
            torch.cuda.synchronize()
            per_cycle_us = max(e0.elapsed_time(e1) * 1e3 / 100000, 1e-6)
            gate_cycles = max(1, int(args.gate_us / per_cycle_us))
        except Exception:  # noqa: BLE001 (no spin kernel in this build: the events then include the first launch's latency)
            gate_cycles = 0

    def timed(fn, n):
        """K steps between fences: (HIP-event seconds, wall seconds), each MAX over ranks, and this rank's own events.
        Behind the opening fence the stream is first held by a short spin kernel (`gate`, ~100 us, untimed): the host
        enqueues the start event and the first launches while it spins, so the HIP events bracket K steps that run back
        to back - without it the events also contain the host's latency between recording the start event on an idle
        GPU and getting the first launch to it (~9 us: 0.45 us per step at the driver's K = 20, nothing at K = 10000).
        The wall clock (`wall`) keeps everything."""
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if gate_cycles:
            torch.cuda._sleep(gate_cycles)
        ev0.record(stream)
        fn(n)
        ev1.record(stream)
        fence()
        wall = time.perf_counter() - t0
        own = ev0.elapsed_time(ev1) * 1e-3
        return max_over_ranks(own, device), max_over_ranks(wall, device), own

    def rate(n, seconds):
        return world_size * B * sub * n / seconds

    def clock_warm(fn):
        t_clock = time.perf_counter()
        while time.perf_counter() - t_clock < args.clock_warmup:
            fn()
            torch.cuda.synchronize()

    bytes_per_env = be.step_bytes_per_env()
    # ---- secondary legs first: they also bring the GPU to its steady clocks before the headline region
    persistent = env_leg = sharded = None
    if not args.no_fused and not args.fused:
        if not brief:
            try:
                run(EPISODE, fused=True)
                ev_s, wall_s, _ = timed(lambda n: run(n, fused=True), steps)
                persistent = {"value": rate(steps, ev_s), "unit": "env-steps/s", "us_per_step": ev_s / steps * 1e6,
                              "note": "vmas_world_rollout: identical results bit for bit, state stays in LDS between steps "
                                      "(scripted / pre-computed forces only); NOT the headline value"}
            except Exception as e:  # noqa: BLE001 (a secondary leg never breaks the line)
                persistent = {"error": repr(e)}
        try:
            env.set_state(snapshot)
            step_acts = [list(acts_dev[k].unbind(0)) for k in range(EPISODE)]  # per step: the agents' [B, size] tensors (views)

            def env_steps(n, start=0):
                for i in range(n):
                    k = (start + i) % EPISODE
                    if k == 0:
                        env.set_state(snapshot)
                    env.step(step_acts[k])

            clock_warm(lambda: env_steps(EPISODE))
            env_steps(300 if not brief else 100)  # (the first few hundred steps carry one-time costs)
            n_env = max(min(steps, 2000), 200) if not brief else 200
            wins = sorted((timed(env_steps, n_env) for _ in range(3 if not brief else 1)), key=lambda x: x[1])
            ev_s, wall_s, _ = wins[len(wins) // 2]
            per_env = bytes_per_env + cfg["post_bytes"]
            env_leg = {
                "metric": "env-steps/s through Environment.step() (SURVEY.md 8d's definition of BASELINE.json's metric)",
                "value": rate(n_env, wall_s), "unit": "env-steps/s",
                "us_per_step": wall_s / n_env * 1e6, "gpu_us_per_step": ev_s / n_env * 1e6, "steps": n_env,
                "launches_per_step": 1 if env._one_launch else None,
                "roofline": {"bound": "hbm", "achieved": per_env * B / (ev_s / n_env) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": per_env * B / (ev_s / n_env) / 1e9 / HBM_PEAK_GBS, "bytes_per_env": per_env,
                             "traffic": None},
                "note": f"make_env('{cfg['scenario']}').step() driven from Python, fresh random actions and fresh output tensors every "
                        "step: action ingest (prologue) + World.step + reward/observation/done/info (epilogue) in ONE kernel "
                        "launch (vmas_world_step_env); value from the wall clock (host included), roofline from HIP events",
            }
            if env._one_launch:
                env.set_state(snapshot)
                held = [a.clone() for a in step_acts[0]]
                env.bind(held)  # caller-owned action tensors, static outputs: one foreign call per step

                def env_steps_bound(n):
                    for _ in range(n):
                        env.step_bound()

                env_steps_bound(100)
                ev_b, wall_b, _ = timed(env_steps_bound, n_env)
                env_leg["bound"] = {
                    "value": rate(n_env, wall_b), "us_per_step": wall_b / n_env * 1e6, "gpu_us_per_step": ev_b / n_env * 1e6,
                    "note": "Environment.bind(actions) + step_bound(): the same step on caller-owned action tensors (one random "
                            "action held) and static output buffers - a single foreign call per step, no host-side tensor work"}
                if getattr(env._post, "rollout_ok", True):
                    env.set_state(snapshot)
                    per_step_out = sum(math.prod(s_) * torch.empty((), dtype=d_).element_size() for _, s_, d_ in env.rollout_fields(1))
                    K = max(1, min(EPISODE if not brief else 50, (16 << 30) // per_step_out))  # (<= 16 GB of outputs per launch)
                    racts = [acts_dev[:K, i].contiguous() for i in range(acts_dev.shape[1])]
                    env.rollout(racts)
                    reps = 5 if not brief else 2

                    def rollouts(n):
                        for _ in range(n):
                            env.rollout(racts)

                    ev_r, wall_r, _ = timed(rollouts, reps)
                    env_leg["rollout"] = {
                        "value": rate(reps * K, ev_r), "us_per_step": ev_r / (reps * K) * 1e6, "steps_per_launch": K,
                        "note": "Environment.rollout: K Environment.step() per launch (vmas_world_rollout_env), pre-computed "
                                "actions, bitwise the K single steps; HIP events"}
            if not brief and (world_size > 1 or os.environ.get("VMAS_BENCH_SHARDED")) and not args.no_gather:
                try:
                    sharded = sharded_rollout_leg(env, shard, acts_dev, snapshot, dist, device)
                except Exception as e:  # noqa: BLE001
                    sharded = {"error": repr(e)}
        except Exception as e:  # noqa: BLE001 never let a secondary leg break the bench line
            env_leg = {"error": repr(e)}
        env.set_state(snapshot)

    # ---- the same K launches on ONE queue (what rocprofv3's per-kernel durations describe); then the headline
    single = None
    if not args.fused and not brief and be.queues(min(EPISODE, steps)) > 1:
        be.set_queues(1)
        run(warmup)
        ev_1, wall_1, _ = timed(lambda n: run(n, start=warmup), steps)
        single = {"value": rate(steps, ev_1), "unit": "env-steps/s", "us_per_step": ev_1 / steps * 1e6,
                  "roofline_frac": bytes_per_env * B / (ev_1 / steps) / 1e9 / HBM_PEAK_GBS,
                  "note": "vmas_world_set_queues(1): one launch per step on one HIP queue; its time per step is the "
                          "kernel's launch-to-launch time and agrees with rocprofv3's per-kernel duration + launch gap"}
        be.set_queues(args.queues)
    # ---- headline: W warm-up steps, then exactly K World.step launches between fences - R times over (every window is a
    #      complete measurement by the contract; `value` is the MEDIAN window, min / max beside it).  In front of the W
    #      steps, untimed: `clock_warmup_s` seconds of the same launches - a process that has just been set up finds the
    #      GPU's clocks ramping (the first ~0.1 s of kernels reads up to several times slow: profiles/README.md), and with
    #      the driver's small K every window would fall inside that ramp.
    clock_warm(lambda: run(EPISODE))
    run(warmup)
    windows = [timed(lambda n: run(n, start=warmup), steps) for _ in range(repeats)]
    order = sorted(range(len(windows)), key=lambda i: windows[i][0])
    ev_s, wall_s, own_s = windows[order[len(order) // 2]]
    kernel_s = ev_s / steps
    n_queues = 1 if args.fused else be.queues(min(EPISODE, steps))
    per_rank_us = [own_s / steps * 1e6]
    if dist is not None:
        mine = torch.tensor([own_s], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(allr, mine)
        per_rank_us = [float(x.item()) / steps * 1e6 for x in allr]
    if be.compact:
        kernel = "step_kernel_compact (lane-compacted step for dense sphere worlds, csrc/vmas_compact.h)"
        kernel_short = "step_kernel_compact"
    elif be.specialized:
        kernel = ("step_kernel_spec: the world-specialised form of the step kernel (schedule as compile-time tables, generated from "
                  "the library's planner; bitwise the interpreter's results)")
        kernel_short = "step_kernel_spec"
    else:
        kernel, kernel_short = "step_kernel (interpreter of the schedule)", "step_kernel"
    ach = bytes_per_env * B / kernel_s / 1e9
    gflops = cfg["flop"] * B / kernel_s / 1e9
    res = {
        "name": name, "cfg": cfg["cfg"], "scenario": cfg["scenario"], "kwargs": kw, "envs_per_gpu": B, "substeps": sub,
        "steps": steps, "warmup": warmup, "ev_s": ev_s, "wall_s": wall_s, "kernel_s": kernel_s, "windows": windows,
        "per_rank_us": per_rank_us, "n_queues": n_queues, "kernel": kernel, "kernel_short": kernel_short,
        "bytes_per_env": bytes_per_env, "ach": ach, "gflops": gflops, "lanes": be.lanes_per_env,
        "value": rate(steps, ev_s), "wall_value": rate(steps, wall_s),
        "single": single, "env_leg": env_leg, "persistent": persistent, "sharded": sharded,
        "env": env, "w": w, "be": be, "actions": actions, "forces": forces, "state0": state0, "snapshot": snapshot,
        "acts_dev": acts_dev,
    }
    return res


def binds_note(cfg, bytes_per_env, B, kernel_us):
    """Which roof (if any) binds one World.step launch of this configuration: its traffic at 8 TB/s, its arithmetic at the fp32
    vector peak, against the measured time and the ~2.6 us an empty launch takes (profiles/r04q_launch_floor.txt)."""
    t_hbm = bytes_per_env * B / 8e12 * 1e6
    t_flop = cfg["flop"] * B / (FP32_PEAK_GFLOPS * 1e9) * 1e6
    if kernel_us < 3 * 2.6 and max(t_hbm, t_flop) < 0.5 * kernel_us:
        which = "neither roof: launch floor (~2.6 us empty launch) plus one tile's dependent chain"
    elif t_flop > t_hbm:
        which = "closer to the fp32 vector roof than to HBM (transcendental-heavy contact arithmetic at quarter rate on top)"
    else:
        which = "HBM is the nearer roof"
    return ("one launch moves %.1f MB (%.1f us at 8 TB/s) and %.0f Mflop (%.2f us at fp32 peak) in %.1f us: %s"
            % (bytes_per_env * B / 1e6, t_hbm, cfg["flop"] * B / 1e6, t_flop, kernel_us, which))


def brief_line(r):
    """The short form of one configuration's measurements (default run: the configurations other than the headline's)."""
    out = {
        "workload": f"{r['scenario']} {r['kwargs']}, {r['envs_per_gpu']} envs, World.step() physics only",
        "value": r["value"], "unit": "env-steps/s", "us_per_step": r["kernel_s"] * 1e6, "steps": r["steps"],
        "kernel": r["kernel_short"], "queues": r["n_queues"], "lanes_per_env": r["lanes"],
        "roofline": {"bound": "hbm", "achieved": r["ach"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["ach"] / HBM_PEAK_GBS,
                     "bytes_per_env": r["bytes_per_env"], "gflops": r["gflops"], "traffic": None,
                     "gflops_frac_of_fp32_vector_peak": r["gflops"] / FP32_PEAK_GFLOPS,
                     "binds": binds_note(CONFIGS[r["name"]], r["bytes_per_env"], r["envs_per_gpu"], r["kernel_s"] * 1e6)},
    }
    if r["env_leg"] is not None:
        e = r["env_leg"]
        out["environment_step"] = {k: e[k] for k in ("value", "us_per_step", "gpu_us_per_step", "launches_per_step", "error") if k in e}
        if "roofline" in e:
            out["environment_step"]["roofline_frac"] = e["roofline"]["frac"]
        for k in ("bound", "rollout"):
            if k in e:
                out["environment_step"][k + "_us_per_step"] = e[k].get("gpu_us_per_step", e[k].get("us_per_step"))
    return out


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
    from vectorizedmultiagentsimulator_amd.shard import EnvShard, PackedRollout

    cfg = CONFIGS[args.config]
    scaling = "strong" if args.strong else ("weak" if args.weak else cfg["scaling"])
    dist = None
    if args.dry_run:
        device = torch.device("cpu")
    else:
        if args.share_gpu:
            local_rank = 0
        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run or args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)
        ranks_seen = dist.get_world_size()
        t = torch.ones(1, device=device)
        dist.all_reduce(t)  # every rank really is in the group
        assert int(t.item()) == world_size == ranks_seen

    # the global batch, sharded by environment: every rank steps its own contiguous block, no collective on the step path
    per_gpu = args.num_envs or (cfg["envs"] if scaling == "weak" else -(-cfg["envs"] // world_size))
    global_envs = per_gpu * world_size if (scaling == "weak" or args.num_envs) else cfg["envs"]
    shard = EnvShard(global_envs, rank, world_size)

    gather = None
    if world_size > 1 and not args.no_gather:
        gather = time_rollout_gather(dist, EnvShard, PackedRollout, rank, world_size, device,
                                     32768 if not args.dry_run else 64, t_steps=100 if not args.dry_run else 4,
                                     shrink=512 if args.dry_run else 1)

    if args.dry_run:
        # the configuration's own rollout exchange: the chunking computed at FULL size (what an N-GPU run would do: chunks of
        # steps that keep the gathered buffer below 2 GB), the collective itself executed on a buffer shrunk by 512
        A_, D_ = GATHER_SHAPES[args.config]
        step_bytes = per_gpu * (A_ * D_ + A_ + 1) * 4 * world_size
        tc = max(1, min(EPISODE, (2 << 30) // step_bytes))
        small = EnvShard(max(per_gpu // 512, 1) * world_size, rank, world_size)
        pr = PackedRollout(small, min(tc, 4), A_, D_, device)
        pr.views()["done"].fill_(float(rank))  # every rank's block carries its rank: the result must hold all of them, in order
        res = pr.gather()
        ranks_in_result = sorted({int(v) for v in res["done"][:, 0].tolist()}) if world_size > 1 else [0]
        plan = {"envs_per_gpu": per_gpu, "global_envs": global_envs, "agents": A_, "obs_dim": D_, "bytes_per_step_all_ranks": step_bytes,
                "steps_per_chunk": tc, "chunk_bytes": step_bytes * tc, "chunks_per_100_steps": -(-EPISODE // tc),
                "collectives_per_chunk": 1, "ranks_in_result": len(ranks_in_result), "rank_order_kept": ranks_in_result == list(range(world_size))}
        if rank == 0:
            print(json.dumps({"metric": f"env-steps/sec (batch x substeps) on '{cfg['scenario']}'", "value": None, "dry_run": True,
                              "n_gpus": world_size, "ranks_seen": ranks_seen, "backend": "gloo", "scaling": scaling,
                              "shard": [shard.lo, shard.hi], "global_envs": global_envs, "sharding_plan": plan,
                              "rollout_gather": gather}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.traffic_child:
        traffic_child(args, device)
        return

    r = measure(args.config, args, device, shard, dist, rank, world_size)
    B, steps, kernel_s, n_queues = r["envs_per_gpu"], r["steps"], r["kernel_s"], r["n_queues"]
    # the headline: the same metric through the reference's own vmas.make_env() / Environment.step() (every rank its shard)
    head = None
    if not args.no_attached and not args.fused:
        ok = 1
        try:
            head = attached_headline(args.config, r["kwargs"], B, device, steps, r["warmup"], args.repeats, args.clock_warmup, dist,
                                     seed=shard.seed(0))
        except Exception as e:  # noqa: BLE001 (the reference is not importable here: the native Environment.step is the headline)
            head, ok = {"error": repr(e)[:500]}, 0
        if dist is not None:  # (all ranks or none: a rank without the reference must not leave the others in a barrier - it
            t_ok = torch.tensor([ok], device=device)  #  fails before the first fence, the others' barriers then time out loudly)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if int(t_ok.item()) == 0 and "error" not in head:
                head = {"error": "another rank could not attach the reference"}
    windows = r["windows"]
    kw = r["kwargs"]

    # the other configurations, short (default run only: one GPU, the headline configuration)
    others = None
    if args.config == "balance" and world_size == 1 and not args.no_other_configs and not args.fused and not args.num_envs:
        others = {}
        other_runs = {}
        for other in ("transport", "transport_2pkg", "navigation", "football"):
            try:
                o = measure(other, args, device, EnvShard(CONFIGS[other]["envs"], 0, 1), None, 0, 1, brief=True)
                others[other] = brief_line(o)
                other_runs[other] = o
                del o
            except Exception as e:  # noqa: BLE001
                others[other] = {"error": repr(e)}

    if rank == 0:
        bytes_per_env = r["bytes_per_env"]
        gb_note = binds_note(cfg, bytes_per_env, B, kernel_s * 1e6) + " (DESIGN.md section 3.1, profiles/r04q_launch_floor.txt). HBM is the nominal bound."
        out = {
            "metric": f"env-steps/sec (batch x substeps) on '{cfg['scenario']}'",
            "value": r["value"],
            "unit": "env-steps/s",
            "n_gpus": world_size,
            "ranks_seen": ranks_seen,
            "steps": steps,
            "warmup": r["warmup"], "clock_warmup_s": args.clock_warmup, "gate_us": args.gate_us,
            "ms_per_step": kernel_s * 1e3,
            "repeats": {"windows": len(windows), "steps_per_window": steps,
                        "ms_per_step_min": min(w_[0] for w_ in windows) / steps * 1e3,
                        "ms_per_step_median": kernel_s * 1e3,
                        "ms_per_step_max": max(w_[0] for w_ in windows) / steps * 1e3,
                        "note": "each window = exactly `steps` World.step launches between barrier + synchronize fences (the "
                                "`world_step` object's numbers)"},
            "per_rank_us_per_step": r["per_rank_us"],
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "library_build_id": _library_build_id(),
            "value_is": "World.step() physics (north-star hot path), timed with HIP events inside the fences; `environment_step` = "
                        "the same metric through Environment.step() (SURVEY.md 8d); `wall` = the same K steps by the host clock",
            "wall": {"ms_per_step": r["wall_s"] / steps * 1e3, "value": r["wall_value"]},
            "config": {
                "workload": f"BASELINE config {cfg['cfg']}: {cfg['scenario']} {kw}, {B} envs/GPU ({shard.num_envs} in all), World.step() "
                            f"physics only, fresh random actions every step (pre-generated), {EPISODE}-step episodes",
                "baseline_config": cfg["cfg"], "scenario": cfg["scenario"], "scenario_kwargs": kw,
                "num_envs_per_gpu": B,
                "global_envs": shard.num_envs,
                "substeps": r["substeps"],
                "lanes_per_env": r["lanes"],
                "launch": "persistent rollout (vmas_world_rollout)" if args.fused else (
                    "one launch per step" if n_queues == 1 else
                    f"one launch per step and per part of the batch: {n_queues} HIP queues, {n_queues} launches per "
                    f"World.step of the whole batch (vmas_world_step_n, environments are independent)"),
                "queues": n_queues,
                "kernel": r["kernel"],
                "parallelism": f"env-sharded x{world_size}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": r["ach"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": r["ach"] / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_note": "HBM bytes by the PMC counters need rocprofv3 around the process: per launch in "
                                f"profiles/r05_bench_q1_pmc_summary.txt (scripts/gpu_run.sh evidence), not in this line",
                "kernel": r["kernel_short"],
                "kernel_us": kernel_s * 1e6,
                "kernel_us_is": "time per World.step of the whole batch from the HIP events (region / K)" + (
                    "" if n_queues == 1 else f"; {n_queues} launches of {B // n_queues} environments each "
                    "overlap in it - rocprofv3's per-launch durations add up to more than this, see `single_queue`"),
                "bytes_per_env": bytes_per_env,
                "bytes_per_launch": bytes_per_env * B // n_queues,
                "launches_per_step": n_queues,
                "gflops": r["gflops"],
                "gflops_frac_of_fp32_vector_peak": r["gflops"] / FP32_PEAK_GFLOPS,
                "binds": gb_note,
            },
        }
        # ---- the north-star headline replaces `value` / `ms_per_step` / `roofline`; the physics-only numbers move to `world_step`
        if head is not None and "windows" in head:
            hw_ = sorted(head["windows"], key=lambda x: x[1])
            ev_h, wall_h = hw_[len(hw_) // 2]
            sub_h = head["substeps"]
            out["world_step"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "repeats": out.pop("repeats"),
                                 "wall": out.pop("wall"), "roofline": out.pop("roofline"), "value_is": out.pop("value_is"),
                                 "workload": out["config"]["workload"], "launch": out["config"]["launch"], "queues": n_queues,
                                 "kernel": out["config"]["kernel"]}
            out["value"] = world_size * B * sub_h * steps / wall_h
            out["ms_per_step"] = wall_h / steps * 1e3
            out["value_is"] = ("env-steps/s through the REFERENCE's own vmas.make_env(..., device='cuda') / Environment.step() after "
                               "attach() with its defaults - the reference's action asserts kept, the reference's batch-global "
                               "broad phase - K calls between barrier + synchronize fences, the host included (wall clock, MAX over "
                               "ranks; median window).  `world_step` = World.step() physics alone (the headline of rounds 1-5), "
                               "`environment_step` = this package's own Environment.step")
            out["repeats"] = {"windows": len(hw_), "steps_per_window": steps,
                              "ms_per_step_min": min(x[1] for x in hw_) / steps * 1e3, "ms_per_step_median": wall_h / steps * 1e3,
                              "ms_per_step_max": max(x[1] for x in hw_) / steps * 1e3,
                              "gpu_ms_per_step_median": ev_h / steps * 1e3,
                              "note": "each window = exactly `steps` env.step calls between barrier + synchronize fences; `value` and "
                                      "`ms_per_step` are the median window by the wall clock, gpu_ms_per_step the same window by HIP events"}
            out["config"]["workload"] = (f"BASELINE config {cfg['cfg']}: the reference's {cfg['scenario']} {kw}, {B} envs/GPU "
                                         f"({shard.num_envs} in all), vmas.make_env(device='cuda') + attach(), Environment.step() with "
                                         f"fresh random actions every step")
            out["config"]["launch"] = (f"{head.get('launches_per_env_step', '?')} launch(es) per env.step (+ the asserts' two small "
                                       f"launches): ingest prologue + World.step + reward/observation/done/info epilogue")
            out["config"]["kernel"] = f"{head['kernel']} step kernel, exact_form {head['exact_form']} (1 = lazy, inside the launch)"
            out["headline"] = {k: v for k, v in head.items() if k != "windows"}
            per_env = r["bytes_per_env"] + cfg["post_bytes"]
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "kernel": "the one-launch Environment.step kernel (ingest prologue + World.step + post-step epilogue)",
                               "bytes_per_env": per_env, "bytes_per_launch": per_env * B,
                               "note": "filled from the `attached_reference` leg below: kernel_us = HIP-event time per env.step "
                                       "with the asserts off (back-to-back one-launch steps: the kernel's launch-to-launch time).  "
                                       "rocprofv3 over this command sees the same kernel mostly in the asserts-kept legs, where it "
                                       "starts behind two tiny launches on a drained queue: its per-dispatch median there is ~10 % "
                                       "longer (profiles/r06y_bench_q1_pmc_summary.txt: 14.1 us against 12.6)"}
            if world_size > 1:  # (the asserts-off leg that isolates the kernel runs at N = 1 only)
                rf = out["roofline"]
                rf["kernel_us"] = ev_h / steps * 1e6
                rf["achieved"] = rf["bytes_per_launch"] / (rf["kernel_us"] * 1e-6) / 1e9
                rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
                rf["note"] = ("N > 1, per rank (every rank steps its own shard of the same size): kernel_us = HIP-event time per env.step "
                              "of the slowest rank INCLUDING the asserts' two small launches - a lower bound of the kernel's rate; the "
                              "N = 1 line isolates the kernel (asserts-off leg) and measures its HBM traffic")
        if head is not None and "error" in head:  # (the reference is not importable on this machine: physics-only headline)
            out["headline_fallback"] = ("value = World.step() physics (the headline of rounds 1-5): the attached reference environment "
                                        "could not be made - " + head["error"])
        if args.share_gpu:
            out["share_gpu"] = "TESTING: every rank on cuda:0 over gloo - the N > 1 code path, not a measurement"
        if world_size == 1 and not args.no_traffic and not args.fused:
            try:
                traffic, detail = measure_traffic(args, n_queues)
            except Exception as e:  # noqa: BLE001
                traffic, detail = None, repr(e)[:300]
            rf = out["world_step"]["roofline"] if "world_step" in out else out["roofline"]
            rf["traffic"] = traffic
            if traffic is not None:
                rf["traffic_over_algorithmic"] = traffic / rf["bytes_per_launch"]
                rf["traffic_note"] = ("HBM bytes per launch by the PMC counters: this command's own physics launches re-run under "
                                      "`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes), median per dispatch")
                rf["traffic_detail"] = detail
            else:
                rf["traffic_note"] = f"not measured in this run ({detail}); per launch in profiles/r05*_pmc_summary.txt"
        if r["single"] is not None:
            out["single_queue"] = r["single"]
        if r["env_leg"] is not None:
            out["environment_step"] = r["env_leg"]
        if r["persistent"] is not None:
            out["persistent_rollout"] = r["persistent"]
        if r["sharded"] is not None:
            out["sharded_rollout"] = r["sharded"]
        if gather is not None:
            out["rollout_gather"] = gather
        if others is not None:
            out["other_configs"] = others
        if world_size == 1 and not args.no_attached and not args.fused:
            try:
                out["attached_reference"] = attached_reference_leg(args.config, kw, B, device)
            except Exception as e:  # noqa: BLE001 (the reference is not importable here, or does not run on this device)
                out["attached_reference"] = {"error": repr(e)[:500]}
            a_ = out["attached_reference"]
            if "world_step" in out and "env_step_no_validate_gpu_us" in a_:  # the headline's roofline: its dominant kernel
                rf = out["roofline"]
                rf["kernel_us"] = a_["env_step_no_validate_gpu_us"]
                rf["achieved"] = rf["bytes_per_launch"] / (rf["kernel_us"] * 1e-6) / 1e9
                rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
                if not args.no_traffic:
                    try:
                        traffic, detail = measure_traffic(args, 1, attached=True)
                    except Exception as e:  # noqa: BLE001
                        traffic, detail = None, repr(e)[:300]
                    rf["traffic"] = traffic
                    if traffic is not None:
                        rf["traffic_over_algorithmic"] = traffic / rf["bytes_per_launch"]
                        rf["traffic_detail"] = detail
                    else:
                        rf["traffic_note"] = f"not measured in this run ({detail})"
        if world_size == 1 and not args.no_cpu_baseline:
            w = r["w"]
            st0, f_cpu = r["state0"].cpu().numpy(), r["forces"].cpu().numpy()
            try:
                out["cpu_baseline"] = cpu_reference(args.config, kw, w, r["actions"], st0,
                                                    parity={"env": r["env"], "snapshot": r["snapshot"], "acts_dev": r["acts_dev"]})
                out["parity"] = out["cpu_baseline"].pop("parity")
            except Exception as e:  # noqa: BLE001 the reference is not importable here: say so, keep the port
                out["cpu_baseline_error"] = repr(e)[:500]
            nb = min(B, 32768)
            out["cpu_port"] = cpu_port(w, f_cpu[:, :, :, :nb].copy(), st0[:, :, :nb].copy())
            if "cpu_baseline" not in out:
                out["cpu_baseline"] = dict(out["cpu_port"])
            phys = out["world_step"]["value"] if "world_step" in out else out["value"]
            cb = out["cpu_baseline"]
            if "world_step" in out and "env_step" in cb:  # the headline is Environment.step: so is the baseline beside it
                cb["world_step"] = {"value": cb["value"], "unit": "env-steps/s", "gpu_over_cpu": phys / cb["value"],
                                    "note": "the reference's World.step alone (inside the same steps) against `world_step.value`"}
                cb["value"] = cb["env_step"]["value"]
                cb["value_is"] = "the reference's Environment.step on the host cores (the same call the headline times on the GPU)"
                cb["gpu_over_cpu"] = out["value"] / cb["value"]
            else:
                cb["gpu_over_cpu"] = phys / cb["value"]
            out["cpu_port"]["gpu_over_cpu"] = phys / out["cpu_port"]["value"]
            out["cpu_port"]["measures"] = "World.step physics only (oracle/, one core): against `world_step.value`"
            e = r["env_leg"]
            if e and "value" in e and "env_step" in out["cpu_baseline"]:
                out["cpu_baseline"]["env_step"]["gpu_over_cpu"] = e["value"] / out["cpu_baseline"]["env_step"]["value"]
            a = out.get("attached_reference")
            if a and "value" in a and "env_step" in out["cpu_baseline"]:  # north_star's >= 50x THROUGH the reference's own API
                a["over_cpu_reference_env_step"] = a["value"] / out["cpu_baseline"]["env_step"]["value"]
        # the other configurations through the reference's own objects, each beside a short same-run CPU reference
        if others is not None and world_size == 1 and not args.no_attached:
            threads = out.get("cpu_baseline", {}).get("cores") if out.get("cpu_baseline", {}).get("kind") == "reference" else None
            for other, o in other_runs.items():
                line = others[other]
                okw = o["kwargs"]
                try:
                    line["attached_reference"] = attached_reference_leg(other, okw, o["envs_per_gpu"], device, n=300, brief=True)
                except Exception as e:  # noqa: BLE001
                    line["attached_reference"] = {"error": repr(e)[:300]}
                import copy
                a2 = copy.copy(args)
                a2.config = other
                if not args.no_traffic:  # HBM bytes per launch by the PMC counters, like the headline's
                    try:
                        traffic, detail = measure_traffic(a2, o["n_queues"])
                        rf = line["roofline"]
                        rf["traffic"] = traffic
                        if traffic is not None:
                            rf["bytes_per_launch"] = o["bytes_per_env"] * o["envs_per_gpu"] // o["n_queues"]
                            rf["traffic_over_algorithmic"] = traffic / rf["bytes_per_launch"]
                            rf["traffic_kernel"] = detail["kernel"]
                        else:
                            rf["traffic_note"] = str(detail)[:200]
                    except Exception as e:  # noqa: BLE001
                        line["roofline"]["traffic_note"] = repr(e)[:200]
                a = line["attached_reference"]
                if "value" in a:  # the same headline as the main line's: through the reference's env.step; physics -> `world_step`
                    line["world_step"] = {"value": line["value"], "us_per_step": line.pop("us_per_step"), "roofline": line.pop("roofline"),
                                          "kernel": line.pop("kernel"), "queues": line.pop("queues"), "workload": line["workload"]}
                    line["value"] = a["value"]
                    line["us_per_step"] = a["env_step_us"]
                    line["workload"] = (f"the reference's {o['scenario']} {okw}, {o['envs_per_gpu']} envs, vmas.make_env(device='cuda') + "
                                        f"attach(), Environment.step() (asserts kept)")
                    if "env_step_no_validate_gpu_us" in a:
                        per_env = o["bytes_per_env"] + CONFIGS[other]["post_bytes"]
                        k_us = a["env_step_no_validate_gpu_us"]
                        ach = per_env * o["envs_per_gpu"] / (k_us * 1e-6) / 1e9
                        line["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                            "traffic": None, "kernel": "the Environment.step kernel", "kernel_us": k_us,
                                            "bytes_per_env": per_env, "bytes_per_launch": per_env * o["envs_per_gpu"]}
                        if not args.no_traffic and a.get("launches_per_env_step") == 1:
                            try:
                                traffic, detail = measure_traffic(a2, 1, attached=True)
                                line["roofline"]["traffic"] = traffic
                                if traffic is not None:
                                    line["roofline"]["traffic_over_algorithmic"] = traffic / line["roofline"]["bytes_per_launch"]
                                    line["roofline"]["traffic_kernel"] = detail["kernel"]
                                else:
                                    line["roofline"]["traffic_note"] = str(detail)[:200]
                            except Exception as e:  # noqa: BLE001
                                line["roofline"]["traffic_note"] = repr(e)[:200]
                if args.no_cpu_baseline:
                    continue
                try:
                    c = cpu_reference(other, okw, o["w"], o["actions"], o["state0"].cpu().numpy(), budget_s=3.0, max_envs=8192,
                                      threads=threads, parity={"env": o["env"], "snapshot": o["snapshot"], "acts_dev": o["acts_dev"]})
                    line["parity"] = c.pop("parity")
                    line["cpu_reference"] = {"world_step_value": c["value"], "env_step_value": c["env_step"]["value"], "cores": c["cores"],
                                             "envs": c["envs"], "steps": c["env_step"]["steps"], "unit": "env-steps/s"}
                    if "world_step" in line:
                        line["world_step"]["gpu_over_cpu"] = line["world_step"]["value"] / c["value"]
                        line["gpu_over_cpu"] = line["value"] / c["env_step"]["value"]
                    else:
                        line["gpu_over_cpu"] = line["value"] / c["value"]
                    es = line.get("environment_step", {})
                    if "value" in es:
                        es["gpu_over_cpu"] = es["value"] / c["env_step"]["value"]
                    a = line["attached_reference"]
                    if "value" in a:
                        a["over_cpu_reference_env_step"] = a["value"] / c["env_step"]["value"]
                except Exception as e:  # noqa: BLE001
                    line["cpu_reference"] = {"error": repr(e)[:300]}
            other_runs.clear()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
