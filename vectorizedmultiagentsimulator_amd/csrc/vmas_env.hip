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
//     the first observation store (s_barrier drains vmcnt, i.e. would wait for stores in flight).
// All arithmetic restates the reference's operation order with the helpers of the physics step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/vmas_env_hip.h"
#include "vmas_device.h"

namespace vmas {
int host_fail(const char* msg);  // vmas_hip.hip: sets vmas_last_error(), returns -1
}
using namespace vmas;

namespace {

// A wave's LDS instructions execute in program order: exchanging data between its own lanes through
// LDS needs no s_barrier (and above all no s_waitcnt vmcnt(0), which would drain the global stores
// of the previous tile) - only the compiler must not reorder the accesses.
VD void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr float kPi = 3.14159265358979323846f;  // torch.pi -> fp32

// torch.remainder(x, pi): fmod, then shifted into [0, pi) (sign of the divisor)
VD float remainder_pi(float x) {
  float r = fmodf(x, kPi);
  if (r != 0.f && r < 0.f) r += kPi;
  return r;
}

// `CH` independent loads in flight, then their stores: a plain copy loop waits for every load
// before it issues the next (load -> s_waitcnt -> ds_write per iteration) = one HBM latency per row.
template <int CH, class Src, class Dst>
VD void burst(int n, Src src, Dst dst) {
  for (int i0 = 0; i0 < n; i0 += CH) {
    float t[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) t[k] = src(i0 + k < n ? i0 + k : n - 1);
#pragma unroll
    for (int k = 0; k < CH; ++k)
      if (i0 + k < n) dst(i0 + k, t[k]);
  }
}

// rows wave, wave + NW, ... of an n-row [n][64] LDS array, loaded by `src(row)`
template <int NW, class Src>
VD void stage_rows(float* col /* array + lane */, int n, int wave, Src src) {
  const int mine = n > wave ? (n - wave + NW - 1) / NW : 0;
  burst<16>(mine, [&](int j) { return src(wave + NW * j); }, [&](int j, float v) { col[(wave + NW * j) * 64] = v; });
}

template <int NW>
struct TileCtx {
  int lane, wave, n_rows;
  long b0, env, e;  // e = env clamped into the batch (loads of the tail lanes stay in bounds)
  bool live;
  VD TileCtx(int batch) {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    b0 = (long)blockIdx.x * 64;
    env = b0 + lane;
    live = env < batch;
    e = live ? env : (long)batch - 1;
    n_rows = (int)(batch - b0 < 64 ? batch - b0 : 64);
  }
};

// Row-major observation tiles: obs element i = row * dim + col of the 64 x dim tile lives at
// slab[row * stride + col]; tab[i] holds that offset (built once per block, one divmod each).
template <int NW>
VD void build_flush_table(int* tab, int dim, int stride, int lane, int wave) {
  for (int k = wave; k < dim; k += NW) {
    const int i = k * 64 + lane, r = i / dim;
    tab[i] = r * stride + (i - r * dim);
  }
}
struct ObsTile {
  float* row;      // slab + lane * stride
  const float* slab;
  const int* tab;  // + lane
  int dim, lane;
  VD void put(int d, float v) const { row[d] = v; }
  VD void put(int d, v2 v) const { row[d] = v.x; row[d + 1] = v.y; }
  // out = first element of this tile in the agent's [batch, dim] matrix; rows >= n_rows dropped
  VD void flush(float* __restrict__ out, int n_rows) const {
    wave_lds_fence();
    const int total = n_rows * dim;
    float* dst = out + lane;
    for (int k0 = 0; k0 < dim; k0 += 4) {
      int idx[4];
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) idx[k] = tab[(k0 + k < dim ? k0 + k : dim - 1) * 64];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = slab[idx[k]];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k0 + k < dim && (k0 + k) * 64 + lane < total) dst[(k0 + k) * 64] = v[k];
    }
    wave_lds_fence();
  }
};

// Distance from point p to the filled box (0 inside): the early-out test of the overlap queries.
VD float box_outside_distance(v2 c, float cs, float sn, float length, float width, v2 p) {
  const v2 q = p - c;
  const float lx = fabsf(q.x * cs + q.y * sn) - length / 2.f, ly = fabsf(q.y * cs - q.x * sn) - width / 2.f;
  return norm2(lx > 0.f ? lx : (lx != lx ? lx : 0.f), ly > 0.f ? ly : (ly != ly ? ly : 0.f));
}

// ------------------------------------------------------------------------------------ ingest
__global__ __launch_bounds__(256) void ingest_kernel(const VmasIngestArgs args, int batch, float* __restrict__ agent_ft,
                                                     long ld, uint32_t* __restrict__ err) {
  const VmasActionSlot& S = args.agents[blockIdx.y];
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad = 0;
  if (env < batch) {
    for (int k = 0; k < S.action_size; ++k) {
      float u = S.action[env * S.action_size + k];
      if (u != u) bad |= VMAS_ACTION_ERR_NAN;
      if (args.clamp) {
        u = max_t(min_t(u, S.u_range[k]), -S.u_range[k]);  // torch.maximum(torch.minimum(u, r), -r)
      } else if (fabsf(u) > S.u_range[k]) {
        bad |= VMAS_ACTION_ERR_OUT_OF_RANGE;
      }
      u = u * S.u_multiplier[k];
      agent_ft[((long)S.agent_index * 3 + k) * ld + env] = u;
      if (S.u_out != nullptr) S.u_out[env * S.action_size + k] = u;
    }
  }
  if (err != nullptr && bad != 0) atomicOr(err, bad);
}

// ------------------------------------------------------------------------------------ balance
// LDS: rows[nE*6][64] | flags[2][64] (line-floor, package-floor) | tab[16][64] | tiles[NW][64][17]
template <int NW>
__global__ __launch_bounds__(NW * 64) void balance_post_kernel(const VmasBalanceDesc d, const VmasBalanceBuffers o,
                                                               int batch, const float* __restrict__ state, long ld,
                                                               int nE) {
  extern __shared__ float lds[];
  const TileCtx<NW> C(batch);
  constexpr int D = 16, STRIDE = D | 1;
  float* rows = lds;
  float* flags = rows + nE * 6 * 64;
  int* tab = (int*)(flags + 2 * 64);
  float* slab = (float*)(tab + D * 64) + C.wave * 64 * STRIDE;
  const ObsTile T = {slab + C.lane * STRIDE, slab, tab + C.lane, D, C.lane};
  auto R = [&](int ent, int f) { return rows[(ent * 6 + f) * 64 + C.lane]; };
  auto P2 = [&](int ent, int f) { return V(R(ent, f), R(ent, f + 1)); };

  // phase 0: the tile's state rows (row i of the packed state is row i here), the flush table
  stage_rows<NW>(rows + C.lane, nE * 6, C.wave, [&](int i) { return state[(long)i * ld + C.e]; });
  build_flush_table<NW>(tab, D, STRIDE, C.lane, C.wave);
  const float prev_shaping = C.live ? o.global_shaping[C.env] : 0.f;
  const float steps_in = (C.wave == 0 && o.limit.steps != nullptr && C.live) ? o.limit.steps[C.env] : 0.f;
  __syncthreads();

  // phase 1: compute_on_the_ground balance.py:218-221, one query per wave:
  //   wave 0: is_overlapping(line, floor) = World.get_distance(box, line) < 0 (core.py:1880-1893)
  //   wave 1: is_overlapping(package, floor), the box-sphere rule (core.py:1932-1961)
  // each skipped when no lane's body can reach the floor box (outside distance of its centre to the
  // box > its reach: conservative, fp slack included, NaN counts as near).
  if (C.wave < 2) {
    const v2 floor = P2(d.floor, 0);
    const float floor_rot = R(d.floor, 4);
    float fs, fc;
    sincosf(floor_rot, &fs, &fc);
    const bool is_line = C.wave == 0;
    const v2 body = P2(is_line ? d.line : d.package, 0);
    const float reach = (is_line ? d.line_length / 2.f : d.package_radius) + kLineMinDist + 1e-3f;
    const bool near = !(box_outside_distance(floor, fc, fs, d.floor_length, d.floor_width, body) > reach);
    int hit = 0;
    if (__any(near)) {
      float fs2, fc2;
      sincosf(floor_rot + kHalfPi, &fs2, &fc2);
      seg_t be[4];
      box_edges(floor, fc, fs, fc2, fs2, d.floor_length, d.floor_width, be);
      if (is_line) {
        float ls, lc;
        sincosf(R(d.line, 4), &ls, &lc);
        const seg_t l = {body, lc, ls, d.line_length / 2.f};
        v2 qb, ql;
        closest_seg_box(be, l, qb, ql);
        hit = (vnorm(qb - ql) - kLineMinDist) < 0.f;
      } else {
        const v2 cp = closest_point_box(be, body);
        const float d_sphere_cp = vnorm(body - cp), d_sphere_box = vnorm(body - floor), d_box_cp = vnorm(floor - cp);
        hit = (d_sphere_box < d_box_cp) || (d_sphere_cp < d.package_radius + kLineMinDist);
      }
    }
    flags[C.wave * 64 + C.lane] = hit ? 1.f : 0.f;
  }
  __syncthreads();

  // phase 2: reward balance.py:223-241 (every wave, it is a handful of operations; wave 0 stores it)
  const v2 pkg = P2(d.package, 0), goal = P2(d.goal, 0), line = P2(d.line, 0);
  const bool on_ground = flags[C.lane] != 0.f || flags[64 + C.lane] != 0.f;
  const float package_dist = vnorm(pkg - goal);
  const float ground_rew = on_ground ? d.fall_reward : 0.f;
  const float shaping = package_dist * d.shaping_factor;
  const float pos_rew = prev_shaping - shaping;
  const float rew = ground_rew + pos_rew;
  if (C.wave == 0) {
    const bool pkg_goal = ((package_dist - d.package_radius) - d.goal_radius) < 0.f;  // core.py:1822-1829
    bool done = on_ground || pkg_goal;                                                // balance.py:260-263
    if (o.limit.steps != nullptr && o.limit.max_steps >= 0.f) done = done || (steps_in + 1.f >= o.limit.max_steps);
    if (C.live) {
      o.global_shaping[C.env] = shaping;
      o.pos_rew[C.env] = pos_rew;
      o.ground_rew[C.env] = ground_rew;
      o.on_the_ground[C.env] = on_ground ? 1 : 0;
      o.done[C.env] = done ? 1 : 0;
      if (o.limit.steps != nullptr) o.limit.steps[C.env] = steps_in + 1.f;  // self.steps += 1 (environment.py:399)
    }
  }

  // observation balance.py:243-258, agents wave, wave + NW, ...
  const v2 pkg_vel = P2(d.package, 2), line_vel = P2(d.line, 2), pkg_goal_rel = pkg - goal;
  const float line_av = R(d.line, 5), rot_mod = remainder_pi(R(d.line, 4));
  for (int a = C.wave; a < d.n_agents; a += NW) {
    const v2 p = P2(d.agent0 + a, 0), v = P2(d.agent0 + a, 2);
    T.put(0, p); T.put(2, v); T.put(4, p - pkg); T.put(6, p - line); T.put(8, pkg_goal_rel);
    T.put(10, pkg_vel); T.put(12, line_vel); T.put(14, line_av); T.put(15, rot_mod);
    T.flush(o.obs + ((long)a * batch + C.b0) * D, C.n_rows);
    if (C.live) o.rew[(long)a * batch + C.env] = rew;
  }
}

// ------------------------------------------------------------------------------------ transport
// LDS: rows[nE*6][64] | pk[3][P][64] (prev shaping -> reward term, on_goal, -) | tab[D][64] | tiles[NW][64][D|1]
template <int NW>
__global__ __launch_bounds__(NW * 64) void transport_post_kernel(const VmasTransportDesc d, const VmasTransportBuffers o,
                                                                 int batch, const float* __restrict__ state, long ld,
                                                                 int nE) {
  extern __shared__ float lds[];
  const TileCtx<NW> C(batch);
  const int D = 4 + 7 * d.n_packages, STRIDE = D | 1, P = d.n_packages;
  float* rows = lds;
  float* term = rows + nE * 6 * 64;  // [P][64]: package.global_shaping in, its reward term out
  float* on_goal_f = term + P * 64;  // [P][64]
  int* tab = (int*)(on_goal_f + P * 64);
  float* slab = (float*)(tab + D * 64) + C.wave * 64 * STRIDE;
  const ObsTile T = {slab + C.lane * STRIDE, slab, tab + C.lane, D, C.lane};
  auto R = [&](int ent, int f) { return rows[(ent * 6 + f) * 64 + C.lane]; };
  auto P2 = [&](int ent, int f) { return V(R(ent, f), R(ent, f + 1)); };

  stage_rows<NW>(rows + C.lane, nE * 6, C.wave, [&](int i) { return state[(long)i * ld + C.e]; });
  stage_rows<NW>(term + C.lane, P, C.wave, [&](int p) { return C.live ? o.global_shaping[(long)p * batch + C.env] : 0.f; });
  build_flush_table<NW>(tab, D, STRIDE, C.lane, C.wave);
  const float steps_in = (C.wave == 0 && o.limit.steps != nullptr && C.live) ? o.limit.steps[C.env] : 0.f;
  __syncthreads();

  // phase 1, packages wave, wave + NW, ...: reward term transport.py:141-161
  const v2 goal = P2(d.goal, 0);
  const float reach = norm2(d.package_length / 2.f, d.package_width / 2.f) + kLineMinDist + 1e-3f;
  for (int p = C.wave; p < P; p += NW) {
    const int ent = d.package0 + p;
    const float dist = vnorm(P2(ent, 0) - goal);
    int on_goal = 0;  // is_overlapping(package, goal): skipped when no lane's goal is within reach of its package
    if (__any(!(dist > d.goal_radius + reach))) {
      DevQuery q = {0, ent, d.goal, kBox, kSphere, d.package_length, d.package_width, 0.f, 0.f, 0.f, d.goal_radius};
      (void)pair_distance(q, rows, 64, C.lane, on_goal);
    }
    const float shaping = dist * d.shaping_factor;
    const float prev = term[p * 64 + C.lane];
    term[p * 64 + C.lane] = on_goal ? 0.f : prev - shaping;
    on_goal_f[p * 64 + C.lane] = on_goal ? 1.f : 0.f;
    if (C.live) {
      o.global_shaping[(long)p * batch + C.env] = shaping;
      o.on_goal[(long)p * batch + C.env] = on_goal ? 1 : 0;
    }
  }
  __syncthreads();

  // phase 2: the shared reward (terms summed in package order), done transport.py:184-191
  float rew = 0.f;
  bool all_on_goal = true;
  for (int p = 0; p < P; ++p) {
    rew = rew + term[p * 64 + C.lane];
    all_on_goal = all_on_goal && on_goal_f[p * 64 + C.lane] != 0.f;
  }
  if (C.wave == 0) {
    bool done = all_on_goal;
    if (o.limit.steps != nullptr && o.limit.max_steps >= 0.f) done = done || (steps_in + 1.f >= o.limit.max_steps);
    if (C.live) {
      o.done[C.env] = done ? 1 : 0;
      if (o.limit.steps != nullptr) o.limit.steps[C.env] = steps_in + 1.f;
    }
  }
  // observation transport.py:165-182
  for (int a = C.wave; a < d.n_agents; a += NW) {
    const v2 ap = P2(d.agent0 + a, 0), av = P2(d.agent0 + a, 2);
    T.put(0, ap); T.put(2, av);
    for (int p = 0; p < P; ++p) {
      const int ent = d.package0 + p;
      const v2 pp = P2(ent, 0);
      T.put(4 + 7 * p, pp - goal); T.put(6 + 7 * p, pp - ap); T.put(8 + 7 * p, P2(ent, 2));
      T.put(10 + 7 * p, on_goal_f[p * 64 + C.lane]);
    }
    T.flush(o.obs + ((long)a * batch + C.b0) * D, C.n_rows);
    if (C.live) o.rew[(long)a * batch + C.env] = rew;
  }
}

// ------------------------------------------------------------------------------------ navigation
// LDS: rows[A*6][64] (agent pos, vel, its goal's pos) | per_agent[A][64] | collide_with[32] | tab[D][64] |
//      tiles[NW][64][D|1].  The LIDAR part of an observation is private to the wave that owns the agent: it
//      goes from HBM into that wave's tile directly, in the opening burst.
template <int NW>
__global__ __launch_bounds__(NW * 64) void navigation_post_kernel(const VmasNavigationDesc d,
                                                                  const VmasNavigationBuffers o, int batch,
                                                                  const float* __restrict__ state, long ld) {
  extern __shared__ float lds[];
  const TileCtx<NW> C(batch);
  const int A = d.n_agents;
  const int n_goal = d.observe_all_goals ? A : 1;
  const int n_rays = d.collisions ? d.n_rays : 0;
  const int D = 4 + 2 * n_goal + n_rays, STRIDE = D | 1;
  float* rows = lds;
  float* per_agent = rows + A * 6 * 64;  // agent.pos_shaping in
  uint32_t* collide_with = (uint32_t*)(per_agent + A * 64);  // [A] bit j: World.collides(agent a, agent j)
  int* tab = (int*)(collide_with + VMAS_ENV_MAX_AGENTS);
  float* slab = (float*)(tab + D * 64) + C.wave * 64 * STRIDE;
  const ObsTile T = {slab + C.lane * STRIDE, slab, tab + C.lane, D, C.lane};

  stage_rows<NW>(rows + C.lane, A * 6, C.wave, [&](int i) {
    const int a = i / 6, k = i - a * 6;
    const long row = k < 4 ? (long)(d.agent0 + a) * 6 + k : (long)d.goal_of[a] * 6 + (k - 4);
    return state[row * ld + C.e];
  });
  if (C.wave < A)  // first agent of this wave: sensor a, ray r = row a * n_rays + r of the cast_rays output
    burst<16>(n_rays, [&](int r) { return o.lidar[((long)C.wave * n_rays + r) * ld + C.e]; },
              [&](int r, float v) { T.put(4 + 2 * n_goal + r, d.lidar_range - v); });
  stage_rows<NW>(per_agent + C.lane, A, C.wave, [&](int a) { return C.live ? o.pos_shaping[(long)a * batch + C.env] : 0.f; });
  build_flush_table<NW>(tab, D, STRIDE, C.lane, C.wave);
  const float steps_in = (C.wave == 0 && o.limit.steps != nullptr && C.live) ? o.limit.steps[C.env] : 0.f;
  if (d.collisions && C.wave == NW - 1 && C.lane < A) {  // the batch-global reduction was made by vmas_world_pair_mask
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

  // agent_reward of every agent (navigation.py:232-242, 206-216), recomputed by every wave: the shared
  // terms need all of them; the wave that owns agent a stores a's terms
  float pos_rew = 0.f, my_pos_rew[(VMAS_ENV_MAX_AGENTS + NW - 1) / NW];
  bool all_reached = true, all_done = true;
  for (int a = 0; a < A; ++a) {
    const float dist = vnorm(pos(a) - goal(a));
    all_reached = all_reached && (dist < d.goal_radius);
    all_done = all_done && (dist < d.agent_radius);  // done(): compared with the AGENT's radius
    const float shaping = dist * d.pos_shaping_factor;
    const float r = per_agent[a * 64 + C.lane] - shaping;
    if (a % NW == C.wave) {
      my_pos_rew[a / NW] = r;
      if (C.live) {
        o.pos_shaping[(long)a * batch + C.env] = shaping;
        o.agent_pos_rew[(long)a * batch + C.env] = r;
      }
    }
    pos_rew = pos_rew + r;
  }
  const float final_rew = all_reached ? d.final_reward : 0.f;
  if (C.wave == 0) {
    bool done = all_done;
    if (o.limit.steps != nullptr && o.limit.max_steps >= 0.f) done = done || (steps_in + 1.f >= o.limit.max_steps);
    if (C.live) {
      o.pos_rew[C.env] = pos_rew;
      o.final_rew[C.env] = final_rew;
      o.done[C.env] = done ? 1 : 0;
      if (o.limit.steps != nullptr) o.limit.steps[C.env] = steps_in + 1.f;
    }
  }

#pragma unroll
  for (int s = 0; s < (VMAS_ENV_MAX_AGENTS + NW - 1) / NW; ++s) {
    const int a = C.wave + s * NW;
    if (a >= A) break;
    const v2 p = pos(a);
    // pairwise penalties navigation.py:218-229: a pair counts only if World.collides(a, b) holds
    float col = 0.f;
    if (d.collisions) {
      uint32_t m = __builtin_amdgcn_readfirstlane(collide_with[a]);
      while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1;
        const float distance = (vnorm(p - pos(j)) - d.agent_radius) - d.agent_radius;
        if (distance <= d.min_collision_distance) col += d.agent_collision_penalty;
      }
    }
    if (C.live) {
      o.collision_rew[(long)a * batch + C.env] = col;
      o.rew[(long)a * batch + C.env] = ((d.shared_rew ? pos_rew : my_pos_rew[s]) + final_rew) + col;
    }
    // observation navigation.py:244-263
    T.put(0, p); T.put(2, vel(a));
    if (d.observe_all_goals) {
      for (int g = 0; g < A; ++g) T.put(4 + 2 * g, p - goal(g));
    } else {
      T.put(4, p - goal(a));
    }
    if (s > 0)  // (the wave's first agent had its rays put in the opening burst)
      burst<16>(n_rays, [&](int r) { return o.lidar[((long)a * n_rays + r) * ld + C.e]; },
                [&](int r, float v) { T.put(4 + 2 * n_goal + r, d.lidar_range - v); });
    T.flush(o.obs + ((long)a * batch + C.b0) * D, C.n_rows);
  }
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
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return host_fail(what);
  return 0;
}

// waves per 64-environment tile: one per agent up to 8 (more waves = shorter dependent chain per wave)
int waves_for(int n_agents) { return n_agents > 4 ? 8 : 4; }

#define LAUNCH_POST(kernel, nw, lds, what, ...)                                                                     \
  do {                                                                                                              \
    if ((nw) == 8) {                                                                                                \
      if (ensure_lds(kernel<8>, lds, what ": hipFuncSetAttribute failed")) return -1;                               \
      hipLaunchKernelGGL(kernel<8>, dim3((batch + 63) / 64), dim3(512), lds, (hipStream_t)stream, __VA_ARGS__);     \
    } else {                                                                                                        \
      if (ensure_lds(kernel<4>, lds, what ": hipFuncSetAttribute failed")) return -1;                               \
      hipLaunchKernelGGL(kernel<4>, dim3((batch + 63) / 64), dim3(256), lds, (hipStream_t)stream, __VA_ARGS__);     \
    }                                                                                                               \
    return check_launch(what);                                                                                      \
  } while (0)

}  // namespace

extern "C" {

int vmas_env_ingest_actions(const VmasIngestArgs* args, int32_t batch, float* agent_ft, int64_t ld, uint32_t* err_flags,
                            void* stream) {
  if (!args || !agent_ft) return host_fail("vmas_env_ingest_actions: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_env_ingest_actions: bad batch / ld");
  if (args->n_agents < 0 || args->n_agents > VMAS_ENV_MAX_AGENTS)
    return host_fail("vmas_env_ingest_actions: n_agents out of range");
  for (int a = 0; a < args->n_agents; ++a) {
    const VmasActionSlot& s = args->agents[a];
    if (!s.action || s.action_size < 2 || s.action_size > 3 || s.agent_index < 0)
      return host_fail("vmas_env_ingest_actions: malformed action slot");
  }
  if (args->n_agents == 0) return 0;
  hipLaunchKernelGGL(ingest_kernel, dim3((batch + 255) / 256, args->n_agents), dim3(256), 0, (hipStream_t)stream, *args,
                     batch, agent_ft, (long)ld, err_flags);
  return check_launch("vmas_env_ingest_actions");
}

int vmas_balance_post_step(const VmasBalanceDesc* d, const VmasBalanceBuffers* o, int32_t batch, const float* state,
                           int64_t ld, void* stream) {
  if (!d || !o || !state) return host_fail("vmas_balance_post_step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_balance_post_step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS) return host_fail("vmas_balance_post_step: n_agents out of range");
  if (!o->global_shaping || !o->obs || !o->rew || !o->pos_rew || !o->ground_rew || !o->on_the_ground || !o->done)
    return host_fail("vmas_balance_post_step: null buffer");
  const int nE = std::max({d->goal, d->package, d->line, d->floor, d->agent0 + d->n_agents - 1}) + 1;
  const int nw = waves_for(d->n_agents);
  const size_t lds = ((size_t)nE * 6 * 64 + 2 * 64 + 16 * 64 + (size_t)nw * 64 * (16 | 1)) * sizeof(float);
  LAUNCH_POST(balance_post_kernel, nw, lds, "vmas_balance_post_step", *d, *o, batch, state, (long)ld, nE);
}

int vmas_transport_post_step(const VmasTransportDesc* d, const VmasTransportBuffers* o, int32_t batch, const float* state,
                             int64_t ld, void* stream) {
  if (!d || !o || !state) return host_fail("vmas_transport_post_step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_transport_post_step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS) return host_fail("vmas_transport_post_step: n_agents out of range");
  if (d->n_packages < 1 || d->n_packages > VMAS_ENV_MAX_PACKAGES)
    return host_fail("vmas_transport_post_step: n_packages out of range");
  if (!o->global_shaping || !o->on_goal || !o->obs || !o->rew || !o->done)
    return host_fail("vmas_transport_post_step: null buffer");
  const int D = 4 + 7 * d->n_packages;
  const int nE = std::max({d->goal, d->package0 + d->n_packages - 1, d->agent0 + d->n_agents - 1}) + 1;
  const int nw = waves_for(d->n_agents);
  const size_t lds = ((size_t)nE * 6 * 64 + 2 * 64 * d->n_packages + D * 64 + (size_t)nw * 64 * (D | 1)) * sizeof(float);
  LAUNCH_POST(transport_post_kernel, nw, lds, "vmas_transport_post_step", *d, *o, batch, state, (long)ld, nE);
}

int vmas_navigation_post_step(const VmasNavigationDesc* d, const VmasNavigationBuffers* o, int32_t batch,
                              const float* state, int64_t ld, void* stream) {
  if (!d || !o || !state) return host_fail("vmas_navigation_post_step: null argument");
  if (batch <= 0 || ld < batch) return host_fail("vmas_navigation_post_step: bad batch / ld");
  if (d->n_agents < 1 || d->n_agents > VMAS_ENV_MAX_AGENTS)
    return host_fail("vmas_navigation_post_step: n_agents out of range");
  if (!o->pos_shaping || !o->obs || !o->rew || !o->agent_pos_rew || !o->pos_rew || !o->final_rew || !o->collision_rew ||
      !o->done)
    return host_fail("vmas_navigation_post_step: null buffer");
  if (d->collisions && (!o->lidar || !o->pair_any || !o->pair_index || d->n_rays < 0))
    return host_fail("vmas_navigation_post_step: collisions need lidar, pair_any and pair_index");
  if (d->collisions && o->lidar_max_rays != d->n_rays)
    return host_fail("vmas_navigation_post_step: every registered sensor must have n_rays rays (lidar_max_rays != n_rays)");
  const int D = 4 + 2 * (d->observe_all_goals ? d->n_agents : 1) + (d->collisions ? d->n_rays : 0);
  const int nw = waves_for(d->n_agents);
  const size_t lds = ((size_t)d->n_agents * (6 + 1) * 64 + VMAS_ENV_MAX_AGENTS + D * 64 + (size_t)nw * 64 * (D | 1)) *
                     sizeof(float);
  LAUNCH_POST(navigation_post_kernel, nw, lds, "vmas_navigation_post_step", *d, *o, batch, state, (long)ld);
}

}  // extern "C"
