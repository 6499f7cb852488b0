// vmas_env_device.h - tile-level device code of the fused Environment.step() stages
// (include/vmas_env_hip.h), shared by the stand-alone post-step kernels (vmas_env.hip) and by the
// epilogue of the physics step kernel (vmas_hip.hip, vmas_world_step_env).
//
// A block owns a tile of 64 consecutive environments, lane = environment, and its `nw` waves
// split the tile's independent work (rows to stage, shared geometric queries, one agent's
// observation each).  `rows` is the tile's state in LDS in the packed layout itself with ld = 64
// (row = entity * 6 + field), which is also the layout the physics step keeps its tile in - so a
// post-step function runs unchanged on a freshly staged copy or on the step kernel's own tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vmas_env_hip.h"
#include "vmas_device.h"

namespace vmas {

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

struct TileCtx {
  int lane, wave, nw, n_rows;
  long b0, env, e;  // e = env clamped into the batch (loads of the tail lanes stay in bounds)
  bool live;
  VD TileCtx(int batch) {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nw = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
    b0 = (long)blockIdx.x * 64;
    env = b0 + lane;
    live = env < batch;
    e = live ? env : (long)batch - 1;
    n_rows = (int)(batch - b0 < 64 ? batch - b0 : 64);
  }
};

// rows wave, wave + nw, ... of an n-row [n][64] LDS array, loaded by `src(row)`
template <class Src>
VD void stage_rows(const TileCtx& C, float* array, int n, Src src) {
  float* col = array + C.lane;
  const int mine = n > C.wave ? (n - C.wave + C.nw - 1) / C.nw : 0;
  burst<16>(mine, [&](int j) { return src(C.wave + C.nw * j); },
            [&](int j, float v) { col[(C.wave + C.nw * j) * 64] = v; });
}

// Row-major observation tiles: obs element i = row * dim + col of the 64 x dim tile lives at
// slab[row * stride + col]; tab[i] holds that offset (built once per block, one divmod each).
VD void build_flush_table(const TileCtx& C, int* tab, int dim, int stride) {
  for (int k = C.wave; k < dim; k += C.nw) {
    const int i = k * 64 + C.lane, r = i / dim;
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
VD ObsTile obs_tile(const TileCtx& C, float* tiles, const int* tab, int dim) {
  float* slab = tiles + C.wave * 64 * (dim | 1);
  return ObsTile{slab + C.lane * (dim | 1), slab, tab + C.lane, dim, C.lane};
}

// Distance from point p to the filled box (0 inside): the early-out test of the overlap queries.
VD float box_outside_distance(v2 c, float cs, float sn, float length, float width, v2 p) {
  const v2 q = p - c;
  const float lx = fabsf(q.x * cs + q.y * sn) - length / 2.f, ly = fabsf(q.y * cs - q.x * sn) - width / 2.f;
  return norm2(lx > 0.f ? lx : (lx != lx ? lx : 0.f), ly > 0.f ? ly : (ly != ly ? ly : 0.f));
}

// What a post-step function needs from HBM besides the state rows; loaded by the caller as early
// as it can (the step kernel issues these loads before its own physics).
struct PostIn {
  float steps;  // Environment.steps of this lane's environment (0 when there is no counter)
};
VD float load_steps(const VmasStepLimit& lim, const TileCtx& C) {
  return (lim.steps != nullptr && C.live) ? lim.steps[C.env] : 0.f;
}
// self.steps += 1 (environment.py:399) and done |= steps >= max_steps (environment.py:407-412)
VD bool apply_step_limit(const VmasStepLimit& lim, const TileCtx& C, float steps_in, bool done) {
  if (lim.steps != nullptr) {
    if (lim.max_steps >= 0.f) done = done || (steps_in + 1.f >= lim.max_steps);
    if (C.live) lim.steps[C.env] = steps_in + 1.f;
  }
  return done;
}

// ------------------------------------------------------------------------------------ balance
// balance.py:218-267.  scratch: flags[2][64] (line-floor, package-floor) | tab[16][64] | tiles[nw][64][17].
// Preconditions: `rows` complete and the flush table built (balance_build_table), both visible to the block;
// the caller loaded prev_shaping (Scenario.global_shaping) and steps_in for this lane.  One block barrier: the
// two overlap queries (waves 0 and 1) run beside the observations of the other waves, the reward follows it.
constexpr int kBalanceObsDim = 16;
VD void balance_build_table(const TileCtx& C, float* scratch) {
  build_flush_table(C, (int*)(scratch + 2 * 64), kBalanceObsDim, kBalanceObsDim | 1);
}
__host__ __device__ inline size_t balance_scratch_floats(int nw) { return 2 * 64 + kBalanceObsDim * 64 + (size_t)nw * 64 * (kBalanceObsDim | 1); }

VD void balance_post_tile(const TileCtx& C, const VmasBalanceDesc& d, const VmasBalanceBuffers& o, int batch,
                          const float* rows, float* scratch, float& prev_shaping /* in: before, out: after this step */,
                          float& steps_in /* wave 0: in/out Environment.steps */, int ablate = 0,
                          const float* floor_trig = nullptr /* this lane's cos, sin, cos2, sin2 rows (stride 64) */) {
  constexpr int D = kBalanceObsDim;
  float* flags = scratch;
  int* tab = (int*)(flags + 2 * 64);
  const ObsTile T = obs_tile(C, (float*)(tab + D * 64), tab, D);
  auto R = [&](int ent, int f) { return rows[(ent * 6 + f) * 64 + C.lane]; };
  auto P2 = [&](int ent, int f) { return V(R(ent, f), R(ent, f + 1)); };

  // phase 1: compute_on_the_ground balance.py:218-221, one query per wave:
  //   wave 0: is_overlapping(line, floor) = World.get_distance(box, line) < 0 (core.py:1880-1893)
  //   wave 1: is_overlapping(package, floor), the box-sphere rule (core.py:1932-1961)
  // each skipped when no lane's body can reach the floor box (outside distance of its centre to the
  // box > its reach: conservative, fp slack included, NaN counts as near).
  if (C.wave < 2 && (ablate & 1)) flags[C.wave * 64 + C.lane] = 0.f;  // profiling: queries off
  if (C.wave < 2 && !(ablate & 1)) {
    const v2 floor = P2(d.floor, 0);
    const float floor_rot = R(d.floor, 4);
    float fs, fc;
    if (floor_trig != nullptr) {  // the physics kernel keeps cos/sin of every box in its tile
      fc = floor_trig[0];
      fs = floor_trig[64];
    } else {
      sincosf(floor_rot, &fs, &fc);
    }
    const bool is_line = C.wave == 0;
    const v2 body = P2(is_line ? d.line : d.package, 0);
    const float reach = (is_line ? d.line_length / 2.f : d.package_radius) + kLineMinDist + 1e-3f;
    const bool near = !(box_outside_distance(floor, fc, fs, d.floor_length, d.floor_width, body) > reach);
    int hit = 0;
    if (__any(near)) {
      float fs2, fc2;
      if (floor_trig != nullptr) {
        fc2 = floor_trig[128];
        fs2 = floor_trig[192];
      } else {
        sincosf(floor_rot + kHalfPi, &fs2, &fc2);
      }
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

  // observation balance.py:243-258 (needs no query result): agent a on wave (a + 2) mod nw, so that with more
  // waves than agents + 2 the observations are written while waves 0 and 1 are still in their queries
  const v2 pkg = P2(d.package, 0), goal = P2(d.goal, 0), line = P2(d.line, 0);
  const v2 pkg_vel = P2(d.package, 2), line_vel = P2(d.line, 2), pkg_goal_rel = pkg - goal;
  const float line_av = R(d.line, 5), rot_mod = remainder_pi(R(d.line, 4));
  const int first = (C.wave + C.nw - (2 % C.nw)) % C.nw;
  for (int a = first; a < d.n_agents && !(ablate & 2); a += C.nw) {  // (2: profiling, observations off)
    const v2 p = P2(d.agent0 + a, 0), v = P2(d.agent0 + a, 2);
    T.put(0, p); T.put(2, v); T.put(4, p - pkg); T.put(6, p - line); T.put(8, pkg_goal_rel);
    T.put(10, pkg_vel); T.put(12, line_vel); T.put(14, line_av); T.put(15, rot_mod);
    T.flush(o.obs + ((long)a * batch + C.b0) * D, C.n_rows);
  }
  __syncthreads();

  // phase 2: reward balance.py:223-241 (every wave: a handful of operations; wave 0 stores the shared terms)
  const bool on_ground = flags[C.lane] != 0.f || flags[64 + C.lane] != 0.f;
  const float package_dist = vnorm(pkg - goal);
  const float ground_rew = on_ground ? d.fall_reward : 0.f;
  const float shaping = package_dist * d.shaping_factor;
  const float pos_rew = prev_shaping - shaping;
  const float rew = ground_rew + pos_rew;
  if (C.wave == 0) {
    const bool pkg_goal = ((package_dist - d.package_radius) - d.goal_radius) < 0.f;  // core.py:1822-1829
    const bool done = apply_step_limit(o.limit, C, steps_in, on_ground || pkg_goal);   // balance.py:260-263
    if (C.live) {
      o.global_shaping[C.env] = shaping;
      o.pos_rew[C.env] = pos_rew;
      o.ground_rew[C.env] = ground_rew;
      o.on_the_ground[C.env] = on_ground ? 1 : 0;
      o.done[C.env] = done ? 1 : 0;
    }
  }
  if (C.live)
    for (int a = first; a < d.n_agents; a += C.nw) o.rew[(long)a * batch + C.env] = rew;
  prev_shaping = shaping;  // (every wave computed it: the next step of a multi-step rollout starts from it)
  if (o.limit.steps != nullptr) steps_in = steps_in + 1.f;
}

// ------------------------------------------------------------------------------------ transport
// transport.py:131-191.  scratch: term[P][64] (package.global_shaping in, reward term out) | on_goal[P][64] |
// tab[D][64] | tiles[nw][64][D|1].  Preconditions: `rows` complete; term[p] holds package p's previous
// shaping (block-visible); steps_in loaded.  Contains two block barriers.
__host__ __device__ inline int transport_obs_dim(int n_packages) { return 4 + 7 * n_packages; }
__host__ __device__ inline size_t transport_scratch_floats(int nw, int n_packages) {
  const int D = transport_obs_dim(n_packages);
  return 2 * 64 * (size_t)n_packages + (size_t)D * 64 + (size_t)nw * 64 * (D | 1);
}

VD void transport_post_tile(const TileCtx& C, const VmasTransportDesc& d, const VmasTransportBuffers& o, int batch,
                            const float* rows, float* scratch, float& steps_in /* wave 0: in/out Environment.steps */) {
  const int P = d.n_packages, D = transport_obs_dim(P);
  float* term = scratch;
  float* on_goal_f = term + P * 64;
  int* tab = (int*)(on_goal_f + P * 64);
  const ObsTile T = obs_tile(C, (float*)(tab + D * 64), tab, D);
  auto R = [&](int ent, int f) { return rows[(ent * 6 + f) * 64 + C.lane]; };
  auto P2 = [&](int ent, int f) { return V(R(ent, f), R(ent, f + 1)); };
  build_flush_table(C, tab, D, D | 1);

  // phase 1, packages wave, wave + nw, ...: reward term transport.py:141-161
  const v2 goal = P2(d.goal, 0);
  const float reach = norm2(d.package_length / 2.f, d.package_width / 2.f) + kLineMinDist + 1e-3f;
  for (int p = C.wave; p < P; p += C.nw) {
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
    const bool done = apply_step_limit(o.limit, C, steps_in, all_on_goal);
    if (C.live) o.done[C.env] = done ? 1 : 0;
  }
  // observation transport.py:165-182
  for (int a = C.wave; a < d.n_agents; a += C.nw) {
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
  if (o.limit.steps != nullptr) steps_in = steps_in + 1.f;
}

// ------------------------------------------------------------------------------------ action ingest
// Environment._set_action (environment.py:616-749, continuous branch) + Holonomic(.WithRotation)
// .process_action for ONE agent slot and this lane's environment: returns the (up to 3) scaled
// action components, stores agent_ft / u_out, and ORs VMAS_ACTION_ERR_* into `bad`.
// `row0`: first row of this step's actions in the caller's [n_steps * batch, action_size] tensor (multi-step rollouts).
VD void ingest_slot(const VmasActionSlot& S, int clamp, long env, bool live, float* __restrict__ agent_ft, long ld,
                    float u_out[3], uint32_t& bad, long row0 = 0) {
  u_out[0] = u_out[1] = u_out[2] = 0.f;
  if (!live) return;
  long flat = 0;
  if (S.action_index != nullptr) {  // flat index -> per-dimension index -> [-u_range, u_range] (environment.py:657-705)
    flat = S.action_index[row0 + env];
    long total = 1;
    for (int k = 0; k < S.action_size; ++k) total *= S.nvec[k];
    if (flat < 0 || flat >= total) {
      bad |= VMAS_ACTION_ERR_OUT_OF_RANGE;
      flat = 0;
    }
  }
  for (int k = 0; k < S.action_size; ++k) {
    float u;
    if (S.action_index != nullptr) {
      long m = 1;
      for (int j = k + 1; j < S.action_size; ++j) m *= S.nvec[j];
      const int n = S.nvec[k];
      int a = (int)(flat / m);
      flat = flat % m;
      if (n & 1) a = a == 0 ? n / 2 : (a <= n / 2 ? a - 1 : a);  // odd count: index 0 is "stay"
      u = ((float)a / (float)(n - 1)) * (2.f * S.u_range[k]) - S.u_range[k];
    } else {
      u = S.action[(row0 + env) * S.action_size + k];
    }
    if (u != u) bad |= VMAS_ACTION_ERR_NAN;
    if (S.action_index != nullptr) {
    } else if (clamp) {
      u = max_t(min_t(u, S.u_range[k]), -S.u_range[k]);  // torch.maximum(torch.minimum(u, r), -r)
    } else if (fabsf(u) > S.u_range[k]) {
      bad |= VMAS_ACTION_ERR_OUT_OF_RANGE;
    }
    u = u * S.u_multiplier[k];
    u_out[k] = u;
    agent_ft[((long)S.agent_index * 3 + k) * ld + env] = u;
    if (S.u_out != nullptr) S.u_out[env * S.action_size + k] = u;
  }
}

// A scripted agent whose script the library knows (VmasAgentScript): same contract as ingest_slot.
VD void run_script(const VmasAgentScript& S, const float* __restrict__ state, long env, bool live,
                   float* __restrict__ agent_ft, long ld, float u_out[3]) {
  u_out[0] = u_out[1] = u_out[2] = 0.f;
  if (!live) return;
  if (S.kind == VMAS_SCRIPT_FOOTBALL_BALL) {  // ball_action_script football.py:1620-1680
    const float* E = state + (long)S.entity * 6 * ld + env;
    const float x = E[0], y = E[ld], vy = E[3 * ld];
    const float thr = S.params[0], half_w = S.params[1], half_l = S.params[2], half_goal = S.params[3];
    auto near = [&](float d) { return 1.f - min_t(d, thr) / thr; };  // 1 at the border, 0 from `thr` away
    const float upper = near(half_w - y), lower = near(half_w + y), right = near(half_l - x), left = near(half_l + x);
    const float slow = 1.f - min_t(fabsf(vy), 0.3f) / 0.3f;  // (the reference damps BOTH components by |vel.y|)
    float ax = ((left - right) * slow) * 0.05f;
    const float ay = ((lower - upper) * slow) * 0.05f;
    if (y < half_goal && y > -half_goal) ax = 0.f;  // no push along x in front of a goal mouth
    u_out[0] = ax;
    u_out[1] = ay;
  }
  for (int k = 0; k < 2; ++k) {
    agent_ft[((long)S.agent_index * 3 + k) * ld + env] = u_out[k];
    if (S.u_out != nullptr) S.u_out[env * 2 + k] = u_out[k];
  }
}

}  // namespace vmas
