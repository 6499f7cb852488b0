// vmas_env.hip - the fused stages of Environment.step() around World.step() for gfx950
// (C ABI: include/vmas_env_hip.h).
//
// One launch per stage instead of the reference's per-agent Python loops of small torch ops:
//   ingest_kernel           Environment._set_action (environment.py:616-749, continuous branch)
//                           + Holonomic(.WithRotation).process_action (dynamics/holonomic.py:14-15)
//   balance_post_kernel     balance.py:218-267     reward / observation / done / info
//   transport_post_kernel   transport.py:131-191
//   navigation_post_kernel  navigation.py:200-285
//
// Mapping (same idea as the physics step): a block owns a tile of 64 consecutive environments,
// lane = environment, and the block's NW waves split the tile's independent work - the rows to
// stage, the shared geometric queries, one agent's observation each.  At the benchmark sizes
// (32 768 environments = 512 tiles on 1024 SIMDs) run time is the length of one wave's
// dependent instruction chain (~12 cycles per instruction at one wave per SIMD, measured with
// SQ_WAVE_CYCLES / SQ_INSTS_*), not bytes: the first version - one wave doing everything for its
// tile - took 10-14 us for 16 MB of traffic.  Hence:
//   * every HBM read is issued in an opening burst of independent loads into LDS (one latency,
//     not one per row); the rows keep the packed layout with ld = 64, so vmas_device.h's geometry
//     helpers run on the LDS copy unchanged;
//   * the reference's observation layout is row-major [batch, obs_dim] per agent: a wave lays its
//     64 x obs_dim tile out in LDS (odd row stride: conflict-free), and streams it out linearly
//     through a divmod table built once per block - coalesced 256-byte stores, ~4 instructions
//     per element;
//   * block barriers only separate the phases stage -> shared queries -> per-agent work, all before
//     the first observation store (vmcnt counts loads and stores in issue order: a wait for any later load
//     is a wait for every store issued before it).
// All arithmetic restates the reference's operation order with the helpers of the physics step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include "../../include/vmas_env_hip.h"
#include "vmas_env_device.h"

namespace vmas {
int host_fail(const char* msg);  // vmas_hip.hip: sets vmas_last_error(), returns -1
}
using namespace vmas;

namespace {


// ------------------------------------------------------------------------------------ ingest
__global__ __launch_bounds__(256) void ingest_kernel(const VmasIngestArgs args, int batch, const float* __restrict__ state,
                                                     float* __restrict__ agent_ft, long ld, uint32_t* __restrict__ err) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad = 0;
  float u[3];
  if ((int)blockIdx.y < args.n_agents)
    ingest_slot(args.agents[blockIdx.y], args.clamp, env, env < batch, agent_ft, ld, u, bad);
  else
  {
    const VmasAgentScript& S = args.scripts[blockIdx.y - args.n_agents];
    run_script(S, state + (long)S.entity * 6 * ld + (env < batch ? env : 0), ld, env, env < batch, agent_ft, ld, u);
  }
  if (err != nullptr && bad != 0) raise_action_error(err, bad);
}

// One thread, enqueued behind a kernel whose completion the host wants to see without a stream synchronisation: the flags the
// kernels in front of it raised (the device-side gate word) and `seq`, into pinned host memory.  (Tried first: the kernel's own
// blocks counting themselves and the last one writing the word - a system-scope release per block is an L2 write-back each:
// +13 us at 512 blocks, +90 us at 2048.)
__global__ void mark_done_kernel(const uint32_t* gate, uint32_t* host_flags, uint32_t* done, uint32_t seq) {
  const uint32_t f = *(const volatile uint32_t*)gate;
  if (f != 0u) __hip_atomic_fetch_or(host_flags, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------ balance / transport
// stand-alone: stage the tile's rows (row i of the packed state is row i of the tile), then the
// shared tile function.  LDS: rows[nE*6][64] | scratch (vmas_env_device.h)
__global__ __launch_bounds__(512) void balance_post_kernel(const VmasBalanceDesc d, const VmasBalanceBuffers o, int batch,
                                                           const float* __restrict__ state, long ld, int nE) {
  extern __shared__ float lds[];
  const TileCtx C(batch);
  stage_rows(C, lds, nE * 6, [&](int i) { return state[(long)i * ld + C.e]; });
  float prev_shaping = C.live ? o.global_shaping[C.env] : 0.f;
  float steps_in = C.wave == 0 ? load_steps(o.limit, C) : 0.f;
  __syncthreads();
  balance_post_tile(C, d, o, batch, lds, lds + nE * 6 * 64, prev_shaping, steps_in);
}

__global__ __launch_bounds__(512) void transport_post_kernel(const VmasTransportDesc d, const VmasTransportBuffers o,
                                                             int batch, const float* __restrict__ state, long ld, int nE) {
  extern __shared__ float lds[];
  const TileCtx C(batch);
  float* scratch = lds + nE * 6 * 64;
  stage_rows(C, lds, nE * 6, [&](int i) { return state[(long)i * ld + C.e]; });
  stage_rows(C, scratch, d.n_packages, [&](int p) { return C.live ? o.global_shaping[(long)p * batch + C.env] : 0.f; });
  float steps_in = C.wave == 0 ? load_steps(o.limit, C) : 0.f;
  __syncthreads();
  transport_post_tile(C, d, o, batch, lds, scratch, steps_in);
}

// ------------------------------------------------------------------------------------ navigation
// LDS: rows[A*6][64] (agent pos, vel, its goal's pos) | per_agent[A][64] | collide_with[32] | tab[D][64] |
//      tiles[NW][64][D|1].  The LIDAR part of an observation is private to the wave that owns the agent: it
//      goes from HBM into that wave's tile directly, in the opening burst.
__global__ __launch_bounds__(512) void navigation_post_kernel(const VmasNavigationDesc d,
                                                                  const VmasNavigationBuffers o, int batch,
                                                                  const float* __restrict__ state, long ld) {
  extern __shared__ float lds[];
  const TileCtx C(batch);
  const int A = d.n_agents;
  const int n_goal = d.observe_all_goals ? A : 1;
  const int n_rays = d.collisions ? d.n_rays : 0;
  const int D = 4 + 2 * n_goal + n_rays;
  float* rows = lds;
  float* per_agent = rows + A * 6 * 64;  // agent.pos_shaping in
  uint32_t* collide_with = (uint32_t*)(per_agent + A * 64);  // [A] bit j: World.collides(agent a, agent j)
  int* tab = (int*)(collide_with + VMAS_ENV_MAX_AGENTS);
  const ObsTile T = obs_tile(C, (float*)(tab + D * 64), tab, D);

  stage_rows(C, rows, A * 6, [&](int i) {
    const int a = i / 6, k = i - a * 6;
    const long row = k < 4 ? (long)(d.agent0 + a) * 6 + k : (long)d.goal_of[a] * 6 + (k - 4);
    return state[row * ld + C.e];
  });
  if (C.wave < A)  // first agent of this wave: sensor a, ray r = row a * n_rays + r of the cast_rays output
    burst<16>(n_rays, [&](int r) { return o.lidar[((long)C.wave * n_rays + r) * ld + C.e]; },
              [&](int r, float v) { T.put(4 + 2 * n_goal + r, d.lidar_range - v); });
  stage_rows(C, per_agent, A, [&](int a) { return C.live ? o.pos_shaping[(long)a * batch + C.env] : 0.f; });
  build_flush_table(C, tab, D, D | 1);
  const float steps_in = C.wave == 0 ? load_steps(o.limit, C) : 0.f;
  if (d.collisions && C.wave == C.nw - 1 && C.lane < A) {  // the batch-global reduction was made by vmas_world_pair_mask
    uint32_t m = 0;
    for (int j = 0; j < A; ++j) {
      const int pi = o.pair_index[C.lane * A + j];
      if (j != C.lane && pi >= 0 && ((o.pair_any[pi >> 5] >> (pi & 31)) & 1u)) m |= 1u << j;
    }
    collide_with[C.lane] = m;
  }
  __syncthreads();
  auto pos = [&](int a) { return V(rows[(a * 6 + 0) * 64 + C.lane], rows[(a * 6 + 1) * 64 + C.lane]); };
  auto vel = [&](int a) { return V(rows[(a * 6 + 2) * 64 + C.lane], rows[(a * 6 + 3) * 64 + C.lane]); };
  auto goal = [&](int a) { return V(rows[(a * 6 + 4) * 64 + C.lane], rows[(a * 6 + 5) * 64 + C.lane]); };

  auto rays = [&](int a, int s, int, int) {
    if (s > 0)  // (the wave's first agent had its rays put in the opening burst)
      burst<16>(n_rays, [&](int r) { return o.lidar[((long)a * n_rays + r) * ld + C.e]; },
                [&](int r, float v) { T.put(4 + 2 * n_goal + r, d.lidar_range - v); });
  };
  navigation_post_body<false>(C, d, o, batch, per_agent, collide_with, T, steps_in, pos, vel, goal, rays);
}

// The collision penalties of a one-launch navigation step (vmas_world_step_env with VMAS_POST_NAVIGATION): the step
// kernel's epilogue stored every agent's reward without them and ORed World.collides' per-tile pair bits into `mask`;
// this kernel - behind it on the same stream, so the reduction over the whole batch is complete - adds
// navigation.py:218-229.  It leaves the mask as it is: the launches alternate between two masks and the step kernel zeroes
// the one this kernel is done with (a "last block clears" counter - every block bumping one word - was 7 of this kernel's
// 9.5 us at 65 536 environments).
__global__ __launch_bounds__(256) void navigation_collision_kernel(const VmasNavigationDesc d, float* __restrict__ rew,
                                                                  float* __restrict__ collision_rew,
                                                                  const int32_t* __restrict__ pair_index, int batch,
                                                                  const float* __restrict__ state, long ld,
                                                                  uint32_t* __restrict__ mask, int mask_words,
                                                                  const uint32_t* gate) {
  // (behind a GATED step launch: the validation in front of it refused an action - nothing of this step happens)
  if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)*(const volatile uint32_t*)gate) != 0) return;
  __shared__ uint32_t collide_with[VMAS_ENV_MAX_AGENTS];
  const int A = d.n_agents;
  if ((int)threadIdx.x < VMAS_ENV_MAX_AGENTS) collide_with[threadIdx.x] = 0u;
  __syncthreads();
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long e = env < batch ? env : (long)batch - 1;
  // Two round trips instead of one per use (a thread per agent walking its row of pair_index chained 2 * A dependent
  // loads: most of this kernel's time): the pair indices with every position and reward of this environment, then the
  // mask words.
  extern __shared__ float sh[];  // xy[2 * A][256] | rw[A][256]
  float* xy = sh + threadIdx.x;
  float* rw = xy + 2 * A * 256;
  const int first = (int)threadIdx.x < A * A ? pair_index[threadIdx.x] : -1;
  // (positions and rewards in ONE burst: rows 0 .. 2A - 1 = xy, 2A .. 3A - 1 = rw, contiguous in `sh`)
  burst<32>(3 * A,
            [&](int i) {
              return i < 2 * A ? state[((long)(d.agent0 + (i >> 1)) * 6 + (i & 1)) * ld + e] : rew[(long)(i - 2 * A) * batch + e];
            },
            [&](int i, float v) { xy[i * 256] = v; });
  // bit j of collide_with[a]: World.collides(agent a, agent j), a thread per (a, j)
  auto look = [&](int i, int pi) {
    const int a = i / A, j = i - a * A;
    if (j != a && pi >= 0 && ((__builtin_nontemporal_load(&mask[pi >> 5]) >> (pi & 31)) & 1u)) atomicOr(&collide_with[a], 1u << j);
  };
  const uint32_t word = first >= 0 ? __builtin_nontemporal_load(&mask[first >> 5]) : 0u;
  if ((int)threadIdx.x < A * A) {
    const int a = threadIdx.x / A, j = threadIdx.x - a * A;
    if (j != a && ((word >> (first & 31)) & 1u)) atomicOr(&collide_with[a], 1u << j);
  }
  for (int i = threadIdx.x + blockDim.x; i < A * A; i += blockDim.x) look(i, pair_index[i]);  // (more than 16 agents)
  __syncthreads();
  if (env >= batch) return;
  auto pos = [&](int a) { return V(xy[2 * a * 256], xy[(2 * a + 1) * 256]); };
  for (int a = 0; a < A; ++a) {
    float col = 0.f;
    uint32_t m = collide_with[a];
    if (m) {
      const v2 p = pos(a);
      while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1;
        const float distance = (vnorm(p - pos(j)) - d.agent_radius) - d.agent_radius;
        if (distance <= d.min_collision_distance) col += d.agent_collision_penalty;
      }
    }
    collision_rew[(long)a * batch + env] = col;
    rew[(long)a * batch + env] = rw[a * 256] + col;
  }
}

// ------------------------------------------------------------------------------------ football
// stand-alone form of football_post_tile (vmas_env_device.h).  LDS: rows[(n + 1) * 6][64] (pos, vel, force of every agent
// and of the ball) | the tile's observation array [64][D + 2] (one pass per agent: its 64 * D floats leave as one run)
__global__ __launch_bounds__(512) void football_post_kernel(const VmasFootballDesc d, const VmasFootballBuffers o, int batch,
                                                            const float* __restrict__ state, long ld, int stp,
                                                            const uint32_t* gate) {
  if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)*(const volatile uint32_t*)gate) != 0) return;  // (see above)
  extern __shared__ float lds[];
  const TileCtx C(batch);
  const int n = d.n_blue + d.n_red;
  float* rows = lds;  // rows[(slot * 6 + k) * 64 + lane], k: px py vx vy fx fy
  stage_rows(C, rows, (n + 1) * 6, [&](int i) {
    const int slot = i / 6, k = i - slot * 6;
    return k < 4 ? state[((long)(d.agent0 + slot) * 6 + k) * ld + C.e] : o.agent_ft[((long)slot * 3 + (k - 4)) * ld + C.e];
  });
  float prev[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) prev[k] = C.live ? o.pos_shaping[(long)k * batch + C.env] : 0.f;  // (every wave: all need the team rewards)
  float steps_in = C.wave == 0 ? load_steps(o.limit, C) : 0.f;
  __syncthreads();
  football_post_tile<true>(C, d, o, batch, [&](int slot, int k) { return rows[(slot * 6 + k) * 64 + C.lane]; },
                     (n + 1) * 6 * 64, -64, prev, steps_in, stp);  // (stp: the step's slab of every per-step output)
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", what, hipGetErrorString(e));
  return host_fail(msg);
}

template <class K>
int ensure_lds(K kernel, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return 0;
  if (bytes > 160 * 1024) return host_fail("observation too wide for one LDS tile");
  // per (device, kernel): the opt-in is sticky on the device it was made on, ask once per size
  static std::map<std::pair<int, const void*>, size_t> set_for;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return host_fail(what);
  size_t& have = set_for[{dev, (const void*)kernel}];
  if (have >= bytes) return 0;
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return host_fail(what);
  have = bytes;
  return 0;
}

// waves per 64-environment tile: one per agent up to 8 (more waves = shorter dependent chain per wave)
int waves_for(int n_agents) { return n_agents > 4 ? 8 : 4; }

#define LAUNCH_POST(kernel, nw, lds, what, ...)                                                                        \
  do {                                                                                                                 \
    if (ensure_lds(kernel, lds, what ": hipFuncSetAttribute failed")) return -1;                                       \
    hipLaunchKernelGGL(kernel, dim3((batch + 63) / 64), dim3(64 * (nw)), lds, (hipStream_t)stream, __VA_ARGS__);       \
    return check_launch(what);                                                                                         \
  } while (0)

}  // namespace

// ---- argument validation, shared with vmas_world_step_env (vmas_hip.hip) ----
namespace vmas {

// fused != 0: as the epilogue of the step kernel (LIDAR cast and World.collides' reduction made in the launch itself)
int check_navigation_args(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch, const float* state,
                          int64_t ld, int fused) {
  if (!d || !o || !state) return host_fail("vmas_navigation_post_step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_navigation_post_step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS)
    return host_fail("vmas_navigation_post_step: n_agents out of range");
  if (!o->pos_shaping || !o->obs || !o->rew || !o->agent_pos_rew || !o->pos_rew || !o->final_rew || !o->collision_rew ||
      !o->done)
    return host_fail("vmas_navigation_post_step: null buffer");
  if (d->collisions && (!o->pair_index || d->n_rays < 0))
    return host_fail("vmas_navigation_post_step: collisions need pair_index and n_rays >= 0");
  if (!fused) {
    if (d->collisions && (!o->lidar || !o->pair_any))
      return host_fail("vmas_navigation_post_step: collisions need lidar, pair_any and pair_index");
    if (d->collisions && o->lidar_max_rays != d->n_rays)
      return host_fail("vmas_navigation_post_step: every registered sensor must have n_rays rays (lidar_max_rays != n_rays)");
  }
  return 0;
}

// football's post-step (vmas_football_post_step; vmas_hip.hip: step `stp` of a rollout whose steps are two launches each)
int launch_football_post(const VmasFootballDesc* d, const VmasFootballBuffers* o, int32_t batch, const float* state, int64_t ld,
                         int stp, void* stream, const uint32_t* gate) {
  if (!d || !o || !state) return host_fail("vmas_football_post_step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_football_post_step: bad batch / ld");
  if (d->n_blue < 1 || d->n_red < 1 || d->n_blue + d->n_red + 1 > VMAS_ENV_MAX_AGENTS || d->agent0 < 0)
    return host_fail("vmas_football_post_step: team sizes out of range");
  if (!o->pos_shaping || !o->obs || !o->rew || !o->terms || !o->touching || !o->done || !o->agent_ft)
    return host_fail("vmas_football_post_step: null buffer");
  const int nw = 8;
  const int n_others = (d->observe_adversaries ? std::max(d->n_red, d->n_blue) : 0) + (d->observe_teammates ? std::max(d->n_blue, d->n_red) - 1 : 0);
  const size_t lds = ((size_t)(d->n_blue + d->n_red + 1) * 6 * 64 + football_shared_slab_floats(64, 16 + 8 * n_others)) * sizeof(float);
  LAUNCH_POST(football_post_kernel, nw, lds, "vmas_football_post_step", *d, *o, batch, state, (long)ld, stp, gate);
}

// behind a step kernel with the navigation epilogue, same stream (navigation_collision_kernel)
int launch_navigation_collisions(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch,
                                 const float* state, int64_t ld, uint32_t* mask, int mask_words, void* stream,
                                 const uint32_t* gate) {
  const size_t lds = (size_t)3 * d->n_agents * 256 * sizeof(float);
  if (ensure_lds(navigation_collision_kernel, lds, "vmas_world_step_env: hipFuncSetAttribute failed")) return -1;
  hipLaunchKernelGGL(navigation_collision_kernel, dim3((batch + 255) / 256), dim3(256), lds, (hipStream_t)stream, *d, o->rew,
                     o->collision_rew, o->pair_index, batch, state, (long)ld, mask, mask_words, gate);
  return check_launch("vmas_world_step_env: navigation collision penalties");
}

int check_ingest_args(const VmasIngestArgs* args, int32_t batch, const float* agent_ft, int64_t ld) {
  if (!args || !agent_ft) return host_fail("action ingest: null argument");
  if (batch <= 0 || ld < batch) return host_fail("action ingest: bad batch / ld");
  if (args->n_agents < 0 || args->n_agents > VMAS_ENV_MAX_AGENTS) return host_fail("action ingest: n_agents out of range");
  for (int a = 0; a < args->n_agents; ++a) {
    const VmasActionSlot& s = args->agents[a];
    if ((!s.action && !s.action_index) || s.action_size < 2 || s.action_size > 3 || s.agent_index < 0 ||
        s.agent_index >= VMAS_ENV_MAX_AGENTS)
      return host_fail("action ingest: malformed action slot");
    for (int k = 0; s.action_index && k < s.action_size; ++k)
      if (s.nvec[k] < 2) return host_fail("action ingest: discrete_action_nvec entries must be >= 2");
  }
  if (args->n_scripts < 0 || args->n_scripts > VMAS_ENV_MAX_SCRIPTS) return host_fail("action ingest: n_scripts out of range");
  for (int i = 0; i < args->n_scripts; ++i) {
    const VmasAgentScript& s = args->scripts[i];
    if (s.kind != VMAS_SCRIPT_FOOTBALL_BALL || s.agent_index < 0 || s.agent_index >= VMAS_ENV_MAX_AGENTS || s.entity < 0)
      return host_fail("action ingest: malformed agent script");
  }
  return 0;
}

static bool bad_entity(int e, int n_entities) { return e < 0 || (n_entities >= 0 && e >= n_entities); }

int check_balance_args(const VmasBalanceDesc* d, const VmasBalanceBuffers* o, int32_t batch, const float* state, int64_t ld,
                       int n_entities) {
  if (!d || !o || !state) return host_fail("balance post-step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("balance post-step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS) return host_fail("balance post-step: n_agents out of range");
  if (bad_entity(d->goal, n_entities) || bad_entity(d->package, n_entities) || bad_entity(d->line, n_entities) ||
      bad_entity(d->floor, n_entities) || bad_entity(d->agent0, n_entities) ||
      bad_entity(d->agent0 + d->n_agents - 1, n_entities))
    return host_fail("balance post-step: entity index out of range");
  if (!o->global_shaping || !o->obs || !o->rew || !o->pos_rew || !o->ground_rew || !o->on_the_ground || !o->done)
    return host_fail("balance post-step: null buffer");
  return 0;
}

int check_transport_args(const VmasTransportDesc* d, const VmasTransportBuffers* o, int32_t batch, const float* state,
                         int64_t ld, int n_entities) {
  if (!d || !o || !state) return host_fail("transport post-step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("transport post-step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS) return host_fail("transport post-step: n_agents out of range");
  if (d->n_packages < 1 || d->n_packages > VMAS_ENV_MAX_PACKAGES)
    return host_fail("transport post-step: n_packages out of range");
  if (bad_entity(d->goal, n_entities) || bad_entity(d->package0, n_entities) ||
      bad_entity(d->package0 + d->n_packages - 1, n_entities) || bad_entity(d->agent0, n_entities) ||
      bad_entity(d->agent0 + d->n_agents - 1, n_entities))
    return host_fail("transport post-step: entity index out of range");
  if (!o->global_shaping || !o->on_goal || !o->obs || !o->rew || !o->done)
    return host_fail("transport post-step: null buffer");
  return 0;
}

}  // namespace vmas

extern "C" {

int vmas_env_ingest_actions(const VmasIngestArgs* args, int32_t batch, const float* state, float* agent_ft, int64_t ld,
                            uint32_t* err_flags, void* stream) {
  if (check_ingest_args(args, batch, agent_ft, ld)) return -1;
  if (args->n_scripts > 0 && !state) return host_fail("vmas_env_ingest_actions: agent scripts need the world state");
  if (args->n_agents + args->n_scripts == 0) return 0;
  hipLaunchKernelGGL(ingest_kernel, dim3((batch + 255) / 256, args->n_agents + args->n_scripts), dim3(256), 0,
                     (hipStream_t)stream, *args, batch, state, agent_ft, (long)ld, err_flags);
  return check_launch("vmas_env_ingest_actions");
}

// The flag block (pinned, mapped, coherent host memory, 64 bytes): word 0 the VMAS_ACTION_ERR_* flags, word 1 the sequence
// number of the last validation that has completed (written by mark_done_kernel), word 2 the host's launch counter, bytes
// 16..23 the address of the GATE word - one uint32 in DEVICE memory that the validation's ingest kernel ORs its flags into and
// a gated step launch (vmas_world_step_env_gated) reads when it starts.
static uint32_t* gate_of(const uint32_t* host) {
  uint32_t* g = nullptr;
  memcpy(&g, host + 4, sizeof(g));
  return g;
}

int vmas_host_word_create(int32_t device_id, uint32_t** host, uint32_t** dev) {
  if (!host || !dev) return host_fail("vmas_host_word_create: null argument");
  if (hipSetDevice(device_id) != hipSuccess) return host_fail("vmas_host_word_create: hipSetDevice failed");
  uint32_t* h = nullptr;
  if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    return host_fail("vmas_host_word_create: hipHostMalloc failed");
  }
  memset(h, 0, 64);
  uint32_t* d = nullptr;
  uint32_t* gate = nullptr;
  if (hipHostGetDevicePointer((void**)&d, h, 0) != hipSuccess || d == nullptr || hipMalloc((void**)&gate, 64) != hipSuccess ||
      hipMemset(gate, 0, 64) != hipSuccess) {
    (void)hipGetLastError();
    if (gate) (void)hipFree(gate);
    (void)hipHostFree(h);
    return host_fail("vmas_host_word_create: mapping the word / allocating the gate failed");
  }
  memcpy(h + 4, &gate, sizeof(gate));
  *host = h;
  *dev = d;
  return 0;
}

void vmas_host_word_destroy(uint32_t* host) {
  if (!host) return;
  uint32_t* gate = gate_of(host);
  if (gate) (void)hipFree(gate);
  (void)hipHostFree(host);
}

uint32_t* vmas_host_word_gate(uint32_t* host) { return host ? gate_of(host) : nullptr; }

int vmas_env_validate_begin(const VmasIngestArgs* args, int32_t batch, const float* state, float* agent_ft, int64_t ld,
                            uint32_t* err_host, uint32_t* err_dev, void* stream) {
  if (!err_host || !err_dev) return host_fail("vmas_env_validate_begin: needs the flag word of vmas_host_word_create");
  if (check_ingest_args(args, batch, agent_ft, ld)) return -1;
  if (args->n_scripts > 0 && !state) return host_fail("vmas_env_validate_begin: agent scripts need the world state");
  uint32_t* gate = gate_of(err_host);
  const uint32_t seq = (++err_host[2]) & 0x3fffffffu;
  if (args->n_agents + args->n_scripts > 0)
    hipLaunchKernelGGL(ingest_kernel, dim3((batch + 255) / 256, args->n_agents + args->n_scripts), dim3(256), 0,
                       (hipStream_t)stream, *args, batch, state, agent_ft, (long)ld, gate);
  hipLaunchKernelGGL(mark_done_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gate, err_dev, err_dev + 1, seq);
  if (check_launch("vmas_env_validate_begin")) return -1;
  return (int)seq;
}

int vmas_env_validate_end(uint32_t* err_host, int32_t seq, void* stream) {
  if (!err_host) return host_fail("vmas_env_validate_end: null flag word");
  // The wait: a one-thread kernel behind the ingest writes `seq` into host memory - polled here, not waited for through the
  // stream (hipStreamSynchronize costs ~15 us on this runtime).  Bounded: after a few milliseconds of polling (a queue backed up
  // behind earlier work) the stream is synchronised the ordinary way.
  bool seen = false;
  for (int spin = 0; spin < 400000; ++spin) {
    if (__atomic_load_n(err_host + 1, __ATOMIC_ACQUIRE) == (uint32_t)seq) { seen = true; break; }
    __builtin_ia32_pause();
  }
  if (!seen) {
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
      (void)hipGetLastError();
      return host_fail("vmas_env_validate_end: hipStreamSynchronize failed");
    }
  }
  const uint32_t flags = __atomic_exchange_n(err_host, 0u, __ATOMIC_ACQ_REL);
  if (flags != 0u) {  // re-open the gate behind whatever was launched against it (stream order: a gated step in front stays shut)
    if (hipMemsetAsync(gate_of(err_host), 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
      (void)hipGetLastError();
      return host_fail("vmas_env_validate_end: clearing the gate failed");
    }
  }
  return (int)(flags & 0x7fffffffu);
}

int vmas_env_validate_actions(const VmasIngestArgs* args, int32_t batch, const float* state, float* agent_ft, int64_t ld,
                              uint32_t* err_host, uint32_t* err_dev, void* stream) {
  const int seq = vmas_env_validate_begin(args, batch, state, agent_ft, ld, err_host, err_dev, stream);
  if (seq < 0) return -1;
  return vmas_env_validate_end(err_host, seq, stream);
}

// ------------------------------------------------------------------------------------ masked reset
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so a reset needs no shared stream state.
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// torch's uniform_ law: lo + (hi - lo) * u, u = 24 random bits / 2^24 in [0, 1)
__device__ __forceinline__ float uniform_in(uint32_t bits, float lo, float hi) {
  return lo + (hi - lo) * ((float)(bits >> 8) * (1.0f / 16777216.0f));
}

__global__ __launch_bounds__(256) void reset_kernel(const VmasResetArgs A, int batch, int nE, int nA,
                                                    const uint8_t* __restrict__ mask, float* __restrict__ state,
                                                    float* __restrict__ agent_ft, long ld) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= batch || !mask[env]) return;  // unmasked environments are not touched
  const uint32_t episode = A.episode[env] + 1u;
  A.episode[env] = episode;
  for (int i = 0; i < nE * 6; ++i) state[(long)i * ld + env] = 0.f;  // World.reset core.py:1184-1192
  for (int i = 0; i < nA * 3; ++i) agent_ft[(long)i * ld + env] = 0.f;
  auto px = [&](int e) -> float& { return state[((long)e * 6 + 0) * ld + env]; };
  auto py = [&](int e) -> float& { return state[((long)e * 6 + 1) * ld + env]; };
  const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
  for (int i = 0; i < A.n_ops; ++i) {
    const VmasSpawnOp& op = A.ops[i];
    float x = op.x_lo, y = op.y_lo;
    if (op.kind == VMAS_SPAWN_UNIFORM) {
      for (uint32_t tries = 0;; ++tries) {  // find_random_pos_for_entity utils.py:276-319, this environment's own loop
        uint32_t c[4] = {(uint32_t)env, episode, ((uint32_t)i << 20) | tries, (uint32_t)((unsigned long)env >> 32)};
        philox4x32(c, k0, k1);
        x = uniform_in(c[0], op.x_lo, op.x_hi);
        y = uniform_in(c[1], op.y_lo, op.y_hi);
        bool overlaps = false;
        for (int j = op.avoid_from; j < i; ++j) {
          const int o = A.ops[j].entity;
          overlaps = overlaps || norm2(px(o) - x, py(o) - y) < op.min_dist;  // torch.cdist(...) < min_dist
        }
        if (!overlaps) break;
        if (tries >= (uint32_t)VMAS_SPAWN_TRIES) {  // an infeasible placement ends instead of hanging - and is counted
          if (A.gave_up != nullptr) atomicAdd(A.gave_up, 1u);
          break;
        }
      }
    } else if (op.kind == VMAS_SPAWN_OFFSET) {
      float dx = op.x_lo;
      if (op.x_hi != op.x_lo) {
        uint32_t c[4] = {(uint32_t)env, episode, (uint32_t)i << 20, (uint32_t)((unsigned long)env >> 32)};
        philox4x32(c, k0, k1);
        dx = uniform_in(c[0], op.x_lo, op.x_hi);
      }
      x = px(op.base) + dx;
      y = py(op.base) + op.y_lo;
    }
    px(op.entity) = x;
    py(op.entity) = y;
    if (op.has_rot) state[((long)op.entity * 6 + 4) * ld + env] = op.rot;
  }
  for (int t = 0; t < A.n_terms; ++t) {
    const VmasResetTerm& T = A.terms[t];
    float v;
    if (T.kind == VMAS_TERM_DIST_POINT) {
      v = norm2(px(T.a) - T.px, py(T.a) - T.py) * T.factor;
    } else if (T.kind == VMAS_TERM_MIN_DIST) {  // torch.cdist(...).min(): football.py:586-600
      float m = kInf;
      for (int e = T.a; e < T.a + T.n; ++e) m = min_t(m, norm2(px(e) - px(T.b), py(e) - py(T.b)));
      v = m * T.factor;
    } else {
      v = T.a < 0 ? T.factor : norm2(px(T.a) - px(T.b), py(T.a) - py(T.b)) * T.factor;
    }
    T.out[env] = v;
  }
  for (int f = 0; f < A.n_flags; ++f) A.flags[f][env] = 0;
  if (A.steps != nullptr) A.steps[env] = 0.f;
}

int vmas_env_reset_where(const VmasResetArgs* a, int32_t batch, int32_t n_entities, int32_t n_agents, const uint8_t* mask,
                         float* state, float* agent_ft, int64_t ld, void* stream) {
  if (!a || !mask || !state || !a->episode) return host_fail("vmas_env_reset_where: null argument");
  if (n_agents > 0 && !agent_ft) return host_fail("vmas_env_reset_where: world has agents but agent_ft is null");
  if (batch <= 0 || ld < batch || n_entities <= 0) return host_fail("vmas_env_reset_where: bad batch / ld / n_entities");
  if (a->n_ops < 0 || a->n_ops > VMAS_RESET_MAX_OPS || a->n_terms < 0 || a->n_terms > VMAS_RESET_MAX_TERMS ||
      a->n_flags < 0 || a->n_flags > 8)
    return host_fail("vmas_env_reset_where: too many operations / terms / flags");
  for (int i = 0; i < a->n_ops; ++i) {
    const VmasSpawnOp& op = a->ops[i];
    if (op.kind < VMAS_SPAWN_UNIFORM || op.kind > VMAS_SPAWN_FIXED || op.entity < 0 || op.entity >= n_entities)
      return host_fail("vmas_env_reset_where: malformed spawn operation");
    if (op.kind == VMAS_SPAWN_OFFSET && (op.base < 0 || op.base >= n_entities))
      return host_fail("vmas_env_reset_where: OFFSET operation without a base entity");
    if (op.kind == VMAS_SPAWN_UNIFORM && (op.avoid_from < 0 || op.avoid_from > i))
      return host_fail("vmas_env_reset_where: avoid_from must name an earlier operation");
  }
  for (int t = 0; t < a->n_terms; ++t) {
    const VmasResetTerm& T = a->terms[t];
    bool bad = !T.out || T.a >= n_entities || T.kind < VMAS_TERM_DIST || T.kind > VMAS_TERM_MIN_DIST;
    if (T.kind == VMAS_TERM_DIST) bad = bad || (T.a >= 0 && (T.b < 0 || T.b >= n_entities));
    if (T.kind == VMAS_TERM_DIST_POINT) bad = bad || T.a < 0;
    if (T.kind == VMAS_TERM_MIN_DIST) bad = bad || T.a < 0 || T.n < 1 || T.a + T.n > n_entities || T.b < 0 || T.b >= n_entities;
    if (bad) return host_fail("vmas_env_reset_where: malformed shaping term");
  }
  for (int f = 0; f < a->n_flags; ++f)
    if (!a->flags[f]) return host_fail("vmas_env_reset_where: null flag tensor");
  hipLaunchKernelGGL(reset_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a, batch, n_entities,
                     n_agents, mask, state, agent_ft, (long)ld);
  return check_launch("vmas_env_reset_where");
}

int vmas_balance_post_step(const VmasBalanceDesc* d, const VmasBalanceBuffers* o, int32_t batch, const float* state,
                           int64_t ld, void* stream) {
  if (check_balance_args(d, o, batch, state, ld, -1)) return -1;
  const int nE = std::max({d->goal, d->package, d->line, d->floor, d->agent0 + d->n_agents - 1}) + 1;
  const int nw = waves_for(d->n_agents);
  const size_t lds = ((size_t)nE * 6 * 64 + balance_scratch_floats(nw)) * sizeof(float);
  LAUNCH_POST(balance_post_kernel, nw, lds, "vmas_balance_post_step", *d, *o, batch, state, (long)ld, nE);
}

int vmas_transport_post_step(const VmasTransportDesc* d, const VmasTransportBuffers* o, int32_t batch, const float* state,
                             int64_t ld, void* stream) {
  if (check_transport_args(d, o, batch, state, ld, -1)) return -1;
  const int nE = std::max({d->goal, d->package0 + d->n_packages - 1, d->agent0 + d->n_agents - 1}) + 1;
  const int nw = waves_for(d->n_agents);
  const size_t lds = ((size_t)nE * 6 * 64 + transport_scratch_floats(nw, d->n_packages)) * sizeof(float);
  LAUNCH_POST(transport_post_kernel, nw, lds, "vmas_transport_post_step", *d, *o, batch, state, (long)ld, nE);
}

int vmas_navigation_post_step(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch,
                              const float* state, int64_t ld, void* stream) {
  if (check_navigation_args(d, o, batch, state, ld, 0)) return -1;
  const int D = 4 + 2 * (d->observe_all_goals ? d->n_agents : 1) + (d->collisions ? d->n_rays : 0);
  // waves per tile: one agent per wave (8) only if four tiles still fit a CU's LDS together - all tiles of
  // a 65536-environment batch resident at once beat shorter chains in two rounds (33 -> 24 us measured)
  auto lds_for = [&](int nw) {
    return ((size_t)d->n_agents * (6 + 1) * 64 + VMAS_ENV_MAX_AGENTS + D * 64 + (size_t)nw * 64 * (D | 1)) * sizeof(float);
  };
  const int nw = d->n_agents > 4 && lds_for(8) <= 40 * 1024 ? 8 : 4;
  const size_t lds = lds_for(nw);
  LAUNCH_POST(navigation_post_kernel, nw, lds, "vmas_navigation_post_step", *d, *o, batch, state, (long)ld);
}

int vmas_football_post_step(const VmasFootballDesc* d, const VmasFootballBuffers* o, int32_t batch, const float* state,
                            int64_t ld, void* stream) {
  return vmas::launch_football_post(d, o, batch, state, ld, 0, stream, nullptr);
}

}  // extern "C"
