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

#include <type_traits>

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

// A block barrier for data exchanged through LDS only: __syncthreads() also waits for every outstanding GLOBAL access of the
// wave (s_waitcnt vmcnt(0)) - which is exactly what a load requested early, to be looked at much later, must not be made to do.
VD void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
  // nw_known: the waves per tile where the caller knows them at compile time (the world-specialised kernels) - blockDim.x is
  // a vector load from the implicit kernel arguments with a wait behind it, at the top of the kernel
  VD TileCtx(int batch, int nw_known = 0) {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nw = nw_known > 0 ? nw_known : __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
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
// balance.py:218-267.  scratch: flags[2][64] (line-floor, package-floor).
// Preconditions: `rows` complete and visible to the block; the caller loaded prev_shaping (Scenario.global_shaping) and
// steps_in for this lane.  One block barrier: the two overlap queries (waves 0 and 1) run beside the observations of the
// other waves, the reward follows it.  Observations: a row is 16 floats = 64 bytes, every lane stores its environment's
// row itself with four 16-byte stores (the four stores of a wave cover whole 128-byte lines back to back and merge in
// L2) - round 2 staged the 64 x 16 tile through LDS and streamed it out through a divmod table: 35 KB of LDS per tile
// and two wave-level fences on the step's dependent chain.
constexpr int kBalanceObsDim = 16;
__host__ __device__ inline size_t balance_scratch_floats(int) { return 2 * 64; }

VD void balance_post_tile(const TileCtx& C, const VmasBalanceDesc& d, const VmasBalanceBuffers& o, int batch,
                          const float* rows, float* scratch, float& prev_shaping /* in: before, out: after this step */,
                          float& steps_in /* wave 0: in/out Environment.steps */, int ablate = 0,
                          const float* floor_trig = nullptr /* this lane's cos, sin, cos2, sin2 rows (stride 64) */) {
  constexpr int D = kBalanceObsDim;
  float* flags = scratch;
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
    // Early out, exact: a sphere whose centre is farther than r + LINE_MIN_DIST outside the box cannot overlap it; a
    // segment whose separating-axis gap to the box (a lower bound of their distance) exceeds LINE_MIN_DIST has a positive
    // World.get_distance (core.py:1880-1893).  The centre-to-box distance the line used to be tested with never fired: the
    // line rides 6 cm above the floor, well within its half length - the four-edge solve ran on every step.
    float ls = 0.f, lc = 1.f;
    bool near;
    if (is_line) {
      sincosf(R(d.line, 4), &ls, &lc);
      near = !(seg_obb_gap(body, lc, ls, d.line_length / 2.f, floor, fc, fs, d.floor_length / 2.f, d.floor_width / 2.f) >
               kLineMinDist + 1e-3f);
    } else {
      near = !(box_outside_distance(floor, fc, fs, d.floor_length, d.floor_width, body) > d.package_radius + kLineMinDist + 1e-3f);
    }
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
    if (C.live) {
      float4* row = (float4*)(o.obs + ((long)a * batch + C.env) * D);
      const v2 dp = p - pkg, dl = p - line;
      row[0] = make_float4(p.x, p.y, v.x, v.y);
      row[1] = make_float4(dp.x, dp.y, dl.x, dl.y);
      row[2] = make_float4(pkg_goal_rel.x, pkg_goal_rel.y, pkg_vel.x, pkg_vel.y);
      row[3] = make_float4(line_vel.x, line_vel.y, line_av, rot_mod);
    }
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

// ------------------------------------------------------------------------------------ LIDAR
// World.cast_rays (core.py:1662-1786) for RAY_CHUNK rays r0 .. of one sensor in one environment.  `col` is that
// environment's column of the packed state - entity e, field f at col[(e * 6 + f) * ld] - in HBM (lidar_kernel: ld = the
// planes' leading dimension) or in the step kernel's LDS tile (the navigation epilogue: ld = 64); `o`, `arot` the
// sensor's own position and rotation.  Wave-level: all lanes of the wave call it together (__any).
struct DevMaskPair { int32_t a, b; float bound_sum; };
struct DevLidar {
  int32_t entity, n_rays, n_targets, target_off, angle_off;
  float max_range, half_range;
};
struct DevTarget { int32_t entity, shape; float length, width, radius; };

// `angles_cs`: cos, sin of (angle + 0.f) for every registered ray, made on the device by lidar_table_kernel with the very
// sincosf below - so when no lane of the wave has its sensor rotated (arot == 0: every agent of `navigation`, any
// non-rotatable carrier) the directions are two uniform loads instead of a ~100-instruction sincosf per ray, same bits.
// `target(ti)`: descriptor of the sensor's ti-th target (from the world's registered list, or implied by the scenario).
template <int RAY_CHUNK, class TargetFn>
VD void lidar_cast_chunk(const DevLidar& L, TargetFn target, const float* __restrict__ angles,
                         const float2* __restrict__ angles_cs, const float* col, long ld, v2 o, float arot, int r0,
                         float (&best)[RAY_CHUNK]) {
  const float R = L.max_range;
  float c[RAY_CHUNK], s[RAY_CHUNK];
  if (angles_cs != nullptr && __all(arot == 0.f)) {
#pragma unroll
    for (int i = 0; i < RAY_CHUNK; ++i) {
      const int r = r0 + i < L.n_rays ? r0 + i : L.n_rays - 1;
      const float2 cs = angles_cs[L.angle_off + r];
      c[i] = cs.x; s[i] = cs.y;
      best[i] = R;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RAY_CHUNK; ++i) {
      const int r = r0 + i < L.n_rays ? r0 + i : L.n_rays - 1;
      const float th = angles[L.angle_off + r] + arot;  // sensors.py:118
      sincosf(th, &s[i], &c[i]);  // one range reduction for both
      best[i] = R;  // core.py:1672-1674
    }
  }
  // ---- sphere targets: only the ones a ray of this environment can reach.  A sphere whose
  //      centre is farther than max_range + r can only produce distances > max_range, which
  //      never lower the running minimum (core.py:1672-1674, 1785), so it is skipped exactly.
  //      Each lane keeps a bit mask of ITS near spheres and the wave walks the masks together:
  //      the loop runs max-popcount times (~3 of 7 in `navigation`) instead of n_targets times.
  unsigned long long near = 0ull;
  const int n_mask = L.n_targets < 64 ? L.n_targets : 64;
  for (int ti = 0; ti < n_mask; ++ti) {
    const DevTarget Tg = target(ti);
    if (Tg.shape != VMAS_SHAPE_SPHERE) continue;
    const float* tp = col + (long)Tg.entity * 6 * ld;
    const float dx = tp[0] - o.x, dy = tp[ld] - o.y;
    const float lim = R + Tg.radius + 1e-4f;
    if (!(dx * dx + dy * dy > lim * lim)) near |= 1ull << ti;  // NaN counts as near
  }
  while (__any(near != 0ull)) {
    if (near != 0ull) {
      const int ti = __ffsll((long long)near) - 1;
      near &= near - 1ull;
      const DevTarget Tg = target(ti);  // per-lane target
      const float* tp = col + (long)Tg.entity * 6 * ld;
      const v2 tpos = V(tp[0], tp[ld]);
      const v2 u = tpos - o;  // _cast_rays_to_sphere core.py:1414-1490
#pragma unroll
      for (int i = 0; i < RAY_CHUNK; ++i) {
        const v2 dir = V(c[i], s[i]);
        const v2 lp = V(o.x + dir.x * L.half_range, o.y + dir.y * L.half_range);
        const v2 cp = closest_point_line<false>(lp, c[i], s[i], 0.f, tpos);
        const float dn = vnorm(tpos - cp);
        const bool ok = (dn < Tg.radius) && (vdot(u, dir) > 0.f);
        const float a = Tg.radius * Tg.radius - dn * dn;
        const float m = sqrt_n(a > 0.f ? a : 1e-8f);
        float dist = vnorm(cp - o) - m;
        dist = ok ? dist : R;
        best[i] = min_t(best[i], dist);
      }
    }
  }
  for (int ti = 0; ti < L.n_targets; ++ti) {
    const DevTarget Tg = target(ti);
    if (Tg.shape == VMAS_SHAPE_SPHERE && ti < 64) continue;  // handled above
    const float* tp = col + (long)Tg.entity * 6 * ld;
    const v2 tpos = V(tp[0], tp[ld]);
    if (Tg.shape == VMAS_SHAPE_SPHERE) {  // (more than 64 targets: plain loop)
      const v2 u = tpos - o;
#pragma unroll
      for (int i = 0; i < RAY_CHUNK; ++i) {
        const v2 dir = V(c[i], s[i]);
        const v2 lp = V(o.x + dir.x * L.half_range, o.y + dir.y * L.half_range);
        const v2 cp = closest_point_line<false>(lp, c[i], s[i], 0.f, tpos);
        const float dn = vnorm(tpos - cp);
        const bool ok = (dn < Tg.radius) && (vdot(u, dir) > 0.f);
        const float a = Tg.radius * Tg.radius - dn * dn;
        const float m = sqrt_n(a > 0.f ? a : 1e-8f);
        float dist = vnorm(cp - o) - m;
        dist = ok ? dist : R;
        best[i] = min_t(best[i], dist);
      }
    } else if (Tg.shape == VMAS_SHAPE_BOX) {  // _cast_rays_to_box core.py:1281-1372
      const float trot = tp[4 * ld];
      const float cn = cosf(-trot), sn = sinf(-trot), cp_ = cosf(trot), sp_ = sinf(trot);
      const v2 p = rotate(o - tpos, cn, sn);
#pragma unroll
      for (int i = 0; i < RAY_CHUNK; ++i) {
        const v2 q = rotate(V(c[i], s[i]), cn, sn);
        const float tx1 = (-Tg.length / 2.f - p.x) / q.x, tx2 = (Tg.length / 2.f - p.x) / q.x;
        const float ty1 = (-Tg.width / 2.f - p.y) / q.y, ty2 = (Tg.width / 2.f - p.y) / q.y;
        const float t0 = max_t(min_t(tx1, tx2), min_t(ty1, ty2));
        const float t1 = min_t(max_t(tx1, tx2), max_t(ty1, ty2));
        const v2 ia = V(t0 * q.x + p.x, t0 * q.y + p.y);
        const v2 iw = rotate(ia, cp_, sp_) + tpos;
        float dist = vnorm(o - iw);
        dist = ((t1 >= t0) && (t0 > 0.f)) ? dist : R;
        best[i] = min_t(best[i], dist);
      }
    } else {  // _cast_rays_to_line core.py:1544-1626
      const float trot = tp[4 * ld];
      const v2 rr = V(cosf(trot) * Tg.length, sinf(trot) * Tg.length);
      const v2 qo = o - tpos;
#pragma unroll
      for (int i = 0; i < RAY_CHUNK; ++i) {
        const v2 dir = V(c[i], s[i]);
        const float rxs = vcross(rr, dir);
        const float tt = vcross(qo, V(dir.x / rxs, dir.y / rxs));
        const float uu = vcross(qo, V(rr.x / rxs, rr.y / rxs));
        float dist = norm2(uu * dir.x, uu * dir.y);
        dist = (rxs == 0.f || tt > 0.5f || tt < -0.5f || uu < 0.f) ? R : dist;
        best[i] = min_t(best[i], dist);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ navigation
// navigation.py:200-285.  One body for the stand-alone kernel (vmas_env.hip: rows staged from HBM, LIDAR read from a
// vmas_world_cast_rays output, World.collides' batch-global reduction read from a vmas_world_pair_mask output) and for
// the epilogue of the physics step kernel (FUSED: rows = the kernel's own tile, LIDAR cast in place on the tile; the
// pairwise collision penalties need the reduction over ALL tiles of the post-step state, so the epilogue only
// contributes this tile's pair bits and stores the reward without them - navigation_collision_kernel, launched behind
// the step kernel, adds them).
constexpr int kNavMaxOwn = VMAS_ENV_MAX_AGENTS / 4;  // agents one wave can own: ceil(n_agents / nw) (host-checked)
__host__ __device__ inline int navigation_obs_dim(const VmasNavigationDesc& d) {
  return 4 + 2 * (d.observe_all_goals ? d.n_agents : 1) + (d.collisions ? d.n_rays : 0);
}
// scratch of the fused epilogue: per_agent[A][64] (agent.pos_shaping in) | misc[2 * MAX_AGENTS] (LIDAR queue length, this
// tile's pair bits | collide_with[MAX_AGENTS]) | staged descriptors: cos, sin[R] | angles[R] |
// pairs[n_pairs] | pair_index[A * A] | the LIDAR queue: 16 bits per (environment, sensor, target) within reach |
// measured[R][65]: lidar_range - measurement of every ray, as the bits of a non-negative float (R = A * n_rays).
// Round 2 staged every wave's observation tile AND the rays' rows ([R][64] floats) here - 19 + 24 KB at eight agents, one
// tile per CU; writing every lane's own row of the observation matrix straight to HBM instead was measured too: a
// quarter of the kernel went into 8-byte stores 72 bytes apart (one L2 request per lane and store - the L2's request
// rate, not its bandwidth, is the bound).  Now a wave writes the 64 x D block of its agent as ONE contiguous run, each
// lane gathering its two consecutive elements from where they already are: the state tile and `measured`.
constexpr int kNavMeasuredStride = 65;  // (lanes of one environment read consecutive rays: consecutive banks)
__host__ __device__ inline size_t navigation_fixed_floats(int n_agents, int n_rays_total, int n_pairs) {
  return (size_t)n_agents * 64 + 2 * VMAS_ENV_MAX_AGENTS + (size_t)n_rays_total * 3 + (size_t)n_pairs * 3 +
         (n_rays_total > 0 ? (size_t)n_agents * n_agents + (size_t)n_agents * (n_agents - 1) * 32 +
                                 (size_t)n_rays_total * kNavMeasuredStride
                           : 0) + 2;
}
// ... | per wave: own[D - n_rays][65], the other columns of the observation of the agent the wave is writing
__host__ __device__ inline size_t navigation_scratch_floats(int nw, int n_agents, int D, int n_rays_total = 0, int n_pairs = 0) {
  const int n_rays = n_agents > 0 ? n_rays_total / n_agents : 0;
  return navigation_fixed_floats(n_agents, n_rays_total, n_pairs) + (size_t)nw * (D - n_rays) * kNavMeasuredStride;
}

// The observation writer of the fused epilogue: nothing is staged (put is a no-op), the caller's `rays(a, slot)` writes
// agent a's block (navigation_post_tile).  The stand-alone kernel keeps ObsTile.
struct ObsGather {
  float* own;  // this wave's [D - n_rays][65] columns, + lane
  int dim;
  VD void put(int k, float v) const { own[k * kNavMeasuredStride] = v; }
  VD void put(int k, v2 v) const { own[k * kNavMeasuredStride] = v.x; own[(k + 1) * kNavMeasuredStride] = v.y; }
};

struct NavNoMid { VD void operator()() const {} };
template <bool FUSED, class Tile, class Pos, class Vel, class Goal, class Rays, class Mid = NavNoMid>
VD void navigation_post_body(const TileCtx& C, const VmasNavigationDesc& d, const VmasNavigationBuffers& o, int batch,
                             const float* per_agent, const uint32_t* collide_with, Tile T, float steps_in,
                             Pos pos, Vel vel, Goal goal, Rays rays /* (agent, slot): its LIDAR part into T */,
                             float* new_shaping = nullptr /* [kNavMaxOwn] out: the shaping of this wave's agents */,
                             Mid mid = Mid() /* runs between the observations and the collision penalties: the fused
                                                epilogue collects the batch's World.collides bits there (collide_with) */) {
  const int A = d.n_agents;
  // agent_reward of every agent (navigation.py:232-242, 206-216), recomputed by every wave: the shared
  // terms need all of them; the wave that owns agent a stores a's terms
  float pos_rew = 0.f, my_pos_rew[kNavMaxOwn];
  bool all_reached = true, all_done = true;
  for (int a = 0; a < A; ++a) {
    const float dist = vnorm(pos(a) - goal(a));
    all_reached = all_reached && (dist < d.goal_radius);
    all_done = all_done && (dist < d.agent_radius);  // done(): compared with the AGENT's radius
    const float shaping = dist * d.pos_shaping_factor;
    const float r = per_agent[a * 64 + C.lane] - shaping;
    if (a % C.nw == C.wave) {
      my_pos_rew[a / C.nw] = r;
      if (new_shaping != nullptr) new_shaping[a / C.nw] = shaping;
      if (C.live) {
        o.pos_shaping[(long)a * batch + C.env] = shaping;
        o.agent_pos_rew[(long)a * batch + C.env] = r;
      }
    }
    pos_rew = pos_rew + r;
  }
  const float final_rew = all_reached ? d.final_reward : 0.f;
  if (C.wave == 0) {
    const bool done = apply_step_limit(o.limit, C, steps_in, all_done);
    if (C.live) {
      o.pos_rew[C.env] = pos_rew;
      o.final_rew[C.env] = final_rew;
      o.done[C.env] = done ? 1 : 0;
    }
  }

  // With at least twice as many waves as agents (16 waves per tile at the shard sizes where a tile has a CU to itself) the
  // waves beyond the first A would idle through the observation writer - the longest single piece of the epilogue: wave w
  // then serves agent w % A and writes every `nparts`-th 512-byte piece of its block (the few columns that are not rays are
  // recomputed by every part into its own scratch: cheaper than a hand-over between waves); part 0 stores the rewards.
  const int nparts = (std::is_same<Tile, ObsGather>::value && C.nw >= 2 * A) ? C.nw / A : 1;
  auto mine = [&](int s, int& a, int& part) {  // this wave's s-th agent (and which part of its block): false = no more
    a = C.wave + s * C.nw;
    part = 0;
    if (nparts > 1) {
      if (s > 0) return false;
      a = C.wave % A;
      part = C.wave / A;
      if (part >= nparts) return false;
    }
    return a < A;
  };
  // ---- observations navigation.py:244-263 first: nothing in them depends on the other tiles
#pragma unroll
  for (int s = 0; s < kNavMaxOwn; ++s) {
    int a, part;
    if (!mine(s, a, part)) break;
    const v2 p = pos(a);
    T.put(0, p); T.put(2, vel(a));
    if (d.observe_all_goals) {
      for (int g = 0; g < A; ++g) T.put(4 + 2 * g, p - goal(g));
    } else {
      T.put(4, p - goal(a));
    }
    rays(a, s, part, nparts);
    if constexpr (std::is_same<Tile, ObsTile>::value) T.flush(o.obs + ((long)a * batch + C.b0) * T.dim, C.n_rows);
  }
  mid();
  // ---- pairwise penalties navigation.py:218-229 (a pair counts only if World.collides(a, b) holds) and the rewards
#pragma unroll
  for (int s = 0; s < kNavMaxOwn; ++s) {
    int a, part;
    if (!mine(s, a, part)) break;
    if (part != 0) break;
    const v2 p = pos(a);
    float col = 0.f;
    if (!FUSED && d.collisions) {
      uint32_t m = __builtin_amdgcn_readfirstlane(collide_with[a]);
      while (m) {
        const int j = __builtin_ctz(m);
        m &= m - 1;
        const float distance = (vnorm(p - pos(j)) - d.agent_radius) - d.agent_radius;
        if (distance <= d.min_collision_distance) col += d.agent_collision_penalty;
      }
    }
    if (C.live) {
      const float partial = (d.shared_rew ? pos_rew : my_pos_rew[s]) + final_rew;
      if (!FUSED || !d.collisions) {
        o.collision_rew[(long)a * batch + C.env] = col;
        o.rew[(long)a * batch + C.env] = partial + col;
      } else {
        o.rew[(long)a * batch + C.env] = partial;  // + its collision penalties: navigation_collision_kernel
      }
    }
  }
}

// ---- grid-wide exchange of pair bits (World.collides' batch-global `.any()`, core.py:2797-2801) inside one launch whose
// tiles are all resident.  `slot`: this barrier's [tile groups][words] 64-bit words, zero when the first tile gets here; a
// word's low half holds one arrival bit per tile of its group of 32, its high half the OR of those tiles' pair bits.
//   publish (threads < words, behind a block barrier that completed `bits`): ONE agent-scope atomic OR per word carries the
//           tile's arrival bit AND its pair bits - whoever sees the arrival sees the bits, and nobody waits for a reply;
//   collect (wave 0, any time later): lane w polls the `groups` words of pair word w - independent loads, one round trip per
//           poll - until every tile's arrival bit is there, and leaves the batch's pair word w in bits[w] (LDS).
// An agent-scope atomic is a round trip to the memory side of ~2.6 us (this library's own stamps): the earlier protocol -
// OR the bits, wait for the reply, bump an arrival counter, poll it, then read the words back with one atomic load per use
// - put three to nine of them in a row on every tile's path.  `spin_limit` bounds the wait: a grid that is not co-resident
// (it should be) is flagged in `flag` / `gave_up` instead of hanging.  Returns false if the wait gave up.
VD void grid_bits_publish(unsigned long long* slot, int words, const uint32_t* bits) {
  if ((int)threadIdx.x < words) {
    const int g = (int)blockIdx.x >> 5;
    const unsigned long long v = ((unsigned long long)bits[threadIdx.x] << 32) | (1ull << (blockIdx.x & 31));
    __hip_atomic_fetch_or(slot + (size_t)g * words + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
VD bool grid_bits_collect(const unsigned long long* slot, unsigned long long* clear_slot /* block 0 zeroes it (may be NULL) */,
                          int words, uint32_t* bits /* LDS out */, uint32_t* flag, uint32_t* gave_up) {
  const int lane = threadIdx.x & 63;
  const int groups = ((int)gridDim.x + 31) >> 5;
  bool ok = true;
  for (int w0 = 0; w0 < words; w0 += 64) {  // (64 pair words = 2048 pairs per pass)
    const int w = w0 + lane;
    uint32_t mask = 0u;
    int spins = 0;
    for (;;) {
      bool arrived = true;
      mask = 0u;
      if (w < words)
        for (int g = 0; g < groups; ++g) {
          const unsigned long long v = __hip_atomic_load(slot + (size_t)g * words + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int in_group = (int)gridDim.x - 32 * g;
          const uint32_t full = in_group >= 32 ? 0xffffffffu : ((1u << in_group) - 1u);
          arrived = arrived && (((uint32_t)v & full) == full);
          mask |= (uint32_t)(v >> 32);
        }
      if (__all(arrived)) break;
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1 << 18)) {
        if (lane == 0) {
          if (flag) __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (gave_up) __hip_atomic_fetch_or(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ok = false;
        break;
      }
    }
    if (w < words) bits[w] = mask;
    if (blockIdx.x == 0 && clear_slot != nullptr && w < words)
      for (int g = 0; g < groups; ++g)
        __hip_atomic_store(clear_slot + (size_t)g * words + w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return ok;
}
__host__ __device__ inline size_t grid_bits_slot_words64(int words, int max_tiles) { return (size_t)words * ((max_tiles + 31) / 32); }

// ---- the LAZY form of the same rule (round 6): nobody waits unless it has to.
// World.collides skips a pair for the WHOLE batch iff no environment's bounding circles overlap (core.py:2797-2801).
// Evaluating the pair anyway differs from that only in an environment that is "in the band": circles apart, narrow-phase
// force non-zero (a sphere just past the tip of a line, off the corner of a box; any pair with a non-finite pose) - and
// only if NO environment of the batch overlaps.  So every tile steps optimistically with every pair on, and
//   * publishes, per pass (step x substep), the pairs some environment of it overlaps: ONE plain 64-bit store per pair word
//     into a slot of ITS OWN - (launch tag << 32) | pair bits - fire and forget.  No atomics: measured, they are what a
//     launch pays for (profiles/r06a_lazy_cost.jsonl, r06g_lazy_stats.txt: bits ORed into words shared by 32 or 64 tiles
//     cost balance at 32 768 environments 7.0 -> 9.4 and 7.6 -> 11.8 us - a kernel does not end before its atomics have
//     made their round trip through the memory side, ~12 ns apiece on one address; the plain stores cost nothing that
//     could be measured, r06e_publish_variants.txt);
//   * notes the pairs for which one of ITS environments is in the band and no environment of the tile overlaps (LDS);
//   * a tile with such a pair - tests/test_broad_phase_lazy_gpu.py counts them - sweeps the other tiles' words of that pair
//     word, the first 64 tiles first (one load per lane of one wave: 4 096 environments - enough for every pair that is
//     commonly in contact), then all of them: some tile has the bit (monotone: final) -> nothing to do; every tile's word
//     carries this launch's tag and the bit is still clear -> the pass is made again for this tile with the pair off
//     (interpreter / specialised kernels) or the pair's contacts are left out of the owners' sums (lane-compacted kernel).
// Only that tile waits, and only for that; what a sweep costs is the price of a hand-off on this part (0.8-1.5 us for a
// granule, 2.4-4 us for all tiles' - MI355X_MICROARCH.md's price list), which the lane-compacted kernel hides by requesting
// the words while it still has its narrow phase to do.  Nothing is ever cleared: the tag - a per-world launch counter kept by
// the host - tells this launch's words from older ones (a refused gated launch writes nothing and the next one has a new
// tag; under graph capture a replay would repeat its tag: captured launches take the launch-per-substep form).  A waiting
// tile keeps its CU slot: grids beyond what is resident make progress as long as not EVERY resident tile waits; the spin is
// bounded and flags `gave_up` (as the barrier form does) instead of hanging.
struct LazyArgs {
  unsigned long long* slots;  // [passes][words][tiles rounded up to 64]; NULL = lazy form off
  uint32_t* flag;             // device word: a wait gave up (vmas_world_exact_status); + 1 .. 3: statistics
  uint32_t* gave_up;          // host-mapped word: the same, read by the next call without a synchronisation
  uint32_t tag;               // this launch's tag (never 0)
  int32_t words;              // pair words = ceil(n_pairs / 32)
  int32_t tiles_pad;          // tiles of the launch rounded up to 64
  int32_t pad;
  unsigned long long band_words;  // bit w: pair word w is exchanged
};
// (pointers out of a by-value argument struct are generic to the compiler: said to be global here, so that the words are read
//  and written with GLOBAL instructions - FLAT ones count on lgkmcnt too - and no run-time address-space test is made of them)
#define VMAS_GLOBAL __attribute__((address_space(1)))
template <class T> VD const VMAS_GLOBAL T* as_global(const T* p) { return (const VMAS_GLOBAL T*)p; }
template <class T> VD VMAS_GLOBAL T* as_global(T* p) { return (VMAS_GLOBAL T*)p; }
VD VMAS_GLOBAL unsigned long long* lazy_slot(const LazyArgs& Z, int pass, int word) {
  return as_global(Z.slots) + ((size_t)pass * Z.words + word) * Z.tiles_pad;
}
// threads < words, behind a block barrier that completed the tile's words (LDS, at word offset bits_off from the launch's
// dynamic LDS base): nobody waits for the stores
VD void lazy_publish(const LazyArgs& Z, int pass, int bits_off) {
  extern __shared__ uint32_t lazy_lds_words[];
  const uint32_t* bits = lazy_lds_words + bits_off;
  const int t = (int)threadIdx.x;
  if (t < Z.words && ((Z.band_words >> t) & 1ull))
    __hip_atomic_store(lazy_slot(Z, pass, t) + blockIdx.x, ((unsigned long long)Z.tag << 32) | bits[t], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// EVERY thread of the block (block-uniform call).  need (LDS, [words], written before a barrier the caller has passed): the
// pairs this tile must know the batch's bit of; acc (LDS, [words]) and misc (LDS, [2]) are scratch.  Returns 0 once every
// needed bit is set somewhere (the optimistic pass stands; acc is partial), 1 if every tile's word carries this launch's
// tag and some needed bit is clear (acc[] = the batch's words) - or the bounded wait gives up (flags, returns 0).
// All waves take part so that a sweep is ONE memory round trip with one load in flight per lane (the kernel's registers are
// decided by its hot phases, not by this one: with eight loads in flight in one wave the lane-compacted kernel went from 80
// to 95 registers - a resident tile per CU less).  need / acc / misc are WORD OFFSETS from the launch's dynamic LDS base:
// through generic pointers the compiler could not always tell their address space, and its run-time cast trips a back-end
// bug of ROCm 7.2 ("Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base") in the lane-compacted kernel.
VD uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d);
  return v;
}
VD int lazy_collect(const LazyArgs& Z, int pass, int need_off, int acc_off, int misc_off) {
  extern __shared__ uint32_t lazy_lds_words[];
  const uint32_t* need = lazy_lds_words + need_off;
  uint32_t* acc = lazy_lds_words + acc_off;
  uint32_t* misc = lazy_lds_words + misc_off;  // [0] some tile of the sweep has not arrived  [1] some needed bit is still clear
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x, lane = tid & 63;
  const int words = Z.words, tiles = (int)gridDim.x;
  for (int w = tid; w < words; w += nt) acc[w] = 0u;
  if (tid == 0) {
    misc[0] = misc[1] = 0u;
    if (Z.flag) atomicAdd(Z.flag + 1, 1u);  // (statistics: tiles that asked - vmas_debug_lazy_stats)
  }
  __syncthreads();
  // `limit` tiles from the asking tile on (wrapping): 64 first - its neighbours in dispatch order, which started when it
  // did - then the whole grid
  int limit = tiles < 64 ? tiles : 64;
  for (int spins = 0;; ++spins) {
    for (int w = 0; w < words; ++w) {
      if (!((Z.band_words >> w) & 1ull) || need[w] == 0u) continue;  // (only the words somebody asks about; block-uniform)
      const VMAS_GLOBAL unsigned long long* slot = lazy_slot(Z, pass, w);
      uint32_t got = 0u;
      bool arrived = true;
      for (int i = tid; i < limit; i += nt) {
        int t = (int)blockIdx.x + i;
        t = t >= tiles ? t - tiles : t;
        const unsigned long long v = __hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool here = (uint32_t)(v >> 32) == Z.tag;
        arrived = arrived && here;
        got |= here ? (uint32_t)v : 0u;
      }
      got = wave_or(got);
      if (lane == 0 && got != 0u) atomicOr(&acc[w], got);  // (LDS; bits only grow from sweep to sweep)
      if (!__all(arrived) && lane == 0) misc[0] = 1u;
    }
    __syncthreads();
    for (int w = tid; w < words; w += nt)
      if ((need[w] & ~acc[w]) != 0u) misc[1] = 1u;
    __syncthreads();
    const bool open = misc[1] != 0u, missing = misc[0] != 0u;
    __syncthreads();
    if (tid == 0) misc[0] = misc[1] = 0u;
    if (!open) return 0;
    if (limit < tiles) {  // the neighbours do not have it: everybody
      limit = tiles;
      continue;
    }
    if (!missing) {
      if (tid == 0 && Z.flag) atomicAdd(Z.flag + 2, 1u);  // (statistics: ... and found a pair off for the whole batch)
      return 1;
    }
    if (tid == 0 && Z.flag) atomicAdd(Z.flag + 3, 1u);  // (statistics: sweeps that had to be repeated)
    __builtin_amdgcn_s_sleep(8);
    if (spins > (1 << 17)) {
      if (tid == 0) {
        if (Z.flag) __hip_atomic_fetch_or(Z.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (Z.gave_up) __hip_atomic_fetch_or(Z.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return 0;
    }
  }
}
// ONE wave asks for ONE pair (the lane-compacted kernel: the owner of an entity with a band contact of the pair, while the
// other waves of the tile go on): 1 = some environment of the batch overlaps the pair (or the bounded wait gave up: flagged),
// 0 = every tile has arrived and none does.  The 64 tiles from the asking one on first, then the whole grid, four loads in
// flight per lane.
VD int lazy_ask_wave(const LazyArgs& Z, int pass, int pair) {
  const int lane = threadIdx.x & 63, tiles = (int)gridDim.x;
  const VMAS_GLOBAL unsigned long long* slot = lazy_slot(Z, pass, pair >> 5);
  const uint32_t bit = 1u << (pair & 31);
  if (lane == 0 && Z.flag) atomicAdd(Z.flag + 1, 1u);  // (statistics: asks - vmas_debug_lazy_stats)
  {
    int t = (int)blockIdx.x + lane;
    t = t >= tiles ? t % tiles : t;
    const unsigned long long v = __hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__any((uint32_t)(v >> 32) == Z.tag && ((uint32_t)v & bit) != 0u)) return 1;
  }
  for (int spins = 0;; ++spins) {
    bool arrived = true, found = false;
    for (int t0 = 0; t0 < tiles; t0 += 256) {
      unsigned long long v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + 64 * k + lane;
        v[k] = t < tiles ? __hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)Z.tag << 32);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool here = (uint32_t)(v[k] >> 32) == Z.tag;
        arrived = arrived && here;
        found = found || (here && ((uint32_t)v[k] & bit) != 0u);
      }
    }
    if (__any(found)) return 1;
    if (__all(arrived)) {
      if (lane == 0 && Z.flag) atomicAdd(Z.flag + 2, 1u);  // (statistics: ... that found the pair off for the whole batch)
      return 0;
    }
    if (lane == 0 && Z.flag) atomicAdd(Z.flag + 3, 1u);  // (statistics: sweeps that had to be repeated)
    __builtin_amdgcn_s_sleep(8);
    if (spins > (1 << 17)) {
      if (lane == 0) {
        if (Z.flag) __hip_atomic_fetch_or(Z.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (Z.gave_up) __hip_atomic_fetch_or(Z.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return 1;
    }
  }
}

// The reference's test `vector_norm(pos_a - pos_b) <= R_a + R_b` on the radicand: torch's norm of a 2-vector is the correctly
// rounded root of fma(dy, dy, dx * dx) (vmas_device.h), and a correctly rounded root is monotone - so the host computes, once
// per pair, the largest fp32 radicand whose root is still <= the bound (`thr`, overlap_threshold in vmas_hip.hip) and the
// test is one fma and one compare, no root.  A NaN / inf distance does not overlap, as in the reference.
VD bool circles_overlap(float dx, float dy, float thr) { return __fmaf_rn(dy, dy, dx * dx) <= thr; }

// What the fused epilogue needs from the world besides the tile: its registered sensors (sensor a = agent a) and its
// static pair list with the mask words the tiles OR their bits into.
struct NavWorld {
  const float* angles;         // [n_agents * n_rays] sensor a = agent a's, its targets = the other agents in order (host-checked)
  const float2* angles_cs;
  const DevMaskPair* pairs;
  uint32_t* mask;  // [(n_pairs + 31) / 32] words, zero when the launch starts
  uint32_t* sync;  // grid-barrier form (NULL: off): unused | timeout flag | ring of four slots of [pair words][tile groups] 64-bit
                   // words: arrival bits of the group's tiles (low half) | the OR of their pair bits (high half)
  uint32_t seq;    // this launch's barrier number
  int32_t n_pairs;
  uint32_t* gave_up;  // host-mapped word set when the barrier gives up waiting (read by the host at the world's next call)
  int32_t ablate;     // profiling builds only (VMAS_ENV_ABLATE): 1 LIDAR off, 2 observation / reward off, 4 the whole epilogue off
  uint32_t* mask_clear;  // NULL, or the OTHER mask of the two the launches alternate between: the one the previous launch's
                         // collision kernel read - zeroed here by tile 0 for the launch after this one
};

// before the physics (the loads fly behind it): agent.pos_shaping of this lane and the observation flush table
VD void navigation_prologue_tile(const TileCtx& C, const VmasNavigationDesc& d, const VmasNavigationBuffers& o,
                                 const NavWorld& nav, int batch, float* scratch) {
  const int A = d.n_agents;
  float* per_agent = scratch;
  int* misc = (int*)(per_agent + A * 64);  // [0] the LIDAR queue's length, [1..] this tile's pair bits
  for (int a = C.wave; a < A; a += C.nw)
    per_agent[a * 64 + C.lane] = C.live ? o.pos_shaping[(long)a * batch + C.env] : 0.f;
  for (int i = threadIdx.x; i < 2 * VMAS_ENV_MAX_AGENTS; i += C.nw * 64) misc[i] = 0;  // (a tile may be a single wave)
  if (d.collisions) {  // the epilogue's descriptors: a dependent global load per use would chain microseconds behind the physics
    const int R = A * d.n_rays;
    float* stage = (float*)(misc + 2 * VMAS_ENV_MAX_AGENTS) + 2;  // (8-byte aligned: cos / sin pairs first)
    for (int i = threadIdx.x; i < R; i += C.nw * 64) {
      const float2 cs = nav.angles_cs[i];  // (cos, sin first: 8-byte aligned whatever R is)
      stage[2 * i] = cs.x; stage[2 * i + 1] = cs.y;
      stage[2 * R + i] = nav.angles[i];
    }
    const float* pw = (const float*)nav.pairs;
    for (int i = threadIdx.x; i < nav.n_pairs * 3; i += C.nw * 64) stage[3 * R + i] = pw[i];
    int* pidx = (int*)(stage + 3 * R + 3 * nav.n_pairs);
    for (int i = threadIdx.x; i < A * A; i += C.nw * 64) pidx[i] = o.pair_index[i];
    uint32_t* measured = (uint32_t*)(pidx + A * A) + A * (A - 1) * 32;  // (behind the queue)
    for (int i = threadIdx.x; i < R * kNavMeasuredStride; i += C.nw * 64) measured[i] = 0u;  // = lidar_range - max_range
  }
}

// after the last substep: `rows` = the tile with the new state, every wave past the barrier behind the integration
VD void navigation_post_tile(const TileCtx& C, const VmasNavigationDesc& d, const VmasNavigationBuffers& o_in,
                             const NavWorld& nav, int batch, const float* rows, float* scratch, float& steps_in,
                             unsigned long long* tr = nullptr /* profiling build: s_memtime stamps of this wave */,
                             int stp = 0 /* step of a multi-step rollout: barrier number nav.seq + stp */,
                             bool last = true) {
  VmasNavigationBuffers o = o_in;
  if (stp > 0) {  // step k of a rollout writes the k-th slab of every per-step output
    const long nb = (long)d.n_agents * batch;
    o.obs += (long)stp * nb * navigation_obs_dim(d); o.rew += (long)stp * nb; o.agent_pos_rew += (long)stp * nb;
    o.collision_rew += (long)stp * nb; o.pos_rew += (long)stp * batch; o.final_rew += (long)stp * batch;
    o.done += (long)stp * batch;
  }
  auto stamp = [&](int k) { if (tr != nullptr && (threadIdx.x & 63) == 0) tr[k] = __builtin_amdgcn_s_memtime(); };
  stamp(6);
#ifdef VMAS_PROFILE
  const int ablate = nav.ablate;
#else
  constexpr int ablate = 0;
#endif
  if (ablate & 4) return;
  const int A = d.n_agents, D = navigation_obs_dim(d), n_goal = d.observe_all_goals ? A : 1;
  const float* per_agent = scratch;
  int* misc = (int*)(scratch + A * 64);
  uint32_t* collide_with = (uint32_t*)(misc + VMAS_ENV_MAX_AGENTS);
  const float* col = rows + C.lane;
  const int words = (nav.n_pairs + 31) >> 5;
  // World.collides' reduction over the batch, two ways.  nav.sync != NULL (every tile of the grid resident at once, at
  // most one per CU): the tiles OR their bits into this launch's mask slot, meet at a grid-wide barrier - its latency
  // behind the LIDAR units - and apply the collision penalties themselves.  Otherwise the bits go to nav.mask, the reward
  // is stored without the penalties and navigation_collision_kernel, launched behind this kernel, adds them.
  const bool grid_sync = nav.sync != nullptr;
  const uint32_t seq = nav.seq + (uint32_t)stp;  // ring of four mask slots: slot seq + 2 is cleared behind barrier seq
  const int groups = ((int)gridDim.x + 31) >> 5;  // tiles in groups of 32: one arrival bit per tile in a word's low half
  const int R = d.collisions ? A * d.n_rays : 0;
  const float* stage = (const float*)(misc + 2 * VMAS_ENV_MAX_AGENTS) + 2;  // staged by navigation_prologue_tile
  const float2* st_cs = (const float2*)stage;
  const float* st_angles = stage + 2 * R;
  const DevMaskPair* st_pairs = (const DevMaskPair*)(stage + 3 * R);
  const int* st_pair_index = (const int*)(stage + 3 * R + 3 * nav.n_pairs);
  const int ray0 = 4 + 2 * n_goal;  // (= D - n_rays: the columns that are not rays)
  float* own_cols = scratch + navigation_fixed_floats(A, R, nav.n_pairs) + (size_t)C.wave * ray0 * kNavMeasuredStride;
  const ObsGather T{own_cols + C.lane, D};
  if (d.collisions) {  // World.collides' reduction over the batch (core.py:2797-2801), this tile's share: into LDS words
                       // now, into the world's mask by one lane of the tile behind the LIDAR barrier
    for (int k = C.wave; k < nav.n_pairs; k += C.nw) {
      const DevMaskPair P = st_pairs[k];
      const float* sa = col + P.a * 6 * 64;
      const float* sb = col + P.b * 6 * 64;
      const bool hit = C.live && norm2(sa[0] - sb[0], sa[64] - sb[64]) <= P.bound_sum;
      if (__any(hit) && C.lane == 0) atomicOr((uint32_t*)&misc[1 + (k >> 5)], 1u << (k & 31));
    }
    if (grid_sync) {  // publish and arrive now (one atomic per word, no reply awaited), collect behind the LIDAR units
      __syncthreads();
      grid_bits_publish((unsigned long long*)(nav.sync + 2) + (size_t)(seq & 3u) * words * groups, words, (const uint32_t*)&misc[1]);
    }
  }
  stamp(7);
  auto pos = [&](int a) { const float* e = col + (d.agent0 + a) * 6 * 64; return V(e[0], e[64]); };
  auto vel = [&](int a) { const float* e = col + (d.agent0 + a) * 6 * 64; return V(e[2 * 64], e[3 * 64]); };
  auto goal = [&](int a) { const float* e = col + d.goal_of[a] * 6 * 64; return V(e[0], e[64]); };
  // LIDAR, lane-compacted.  A lane per environment walking its sensor's targets spends most of its time on nothing: of
  // the 7 other agents ~1 is within reach of a sensor, but a wave loops as often as its unluckiest lane (~3 times), and of
  // the 12 rays ~2 can touch a sphere that is.  Instead:
  //   1. every (sensor, target) pair of every environment of the tile is tested for reach (the very test of
  //      lidar_cast_chunk) and the near ones are queued in LDS - ballot + one LDS atomic per wave and pair;
  //   2. a lane per queued (environment, sensor, target): a conservative filter over its rays (distance of the target's
  //      centre from the ray's line against radius + a margin far above the rounding of the exact expression; the sign
  //      test is the exact one), then the reference's arithmetic (core.py:1414-1490, expression for expression that of
  //      lidar_cast_chunk) for the rays that pass.  A ray that is filtered out measures max_range against this target
  //      in the reference too, which never lowers the minimum (core.py:1672-1674, 1785);
  //   3. the observation holds lidar_range - min(distances): `measured` starts at zero (= lidar_range - max_range) and
  //      every hit nearer than max_range is folded in with an LDS atomic max on the float's bits - x -> lidar_range - x
  //      is monotone under rounding, so max(lidar_range - d_i) IS lidar_range - min(d_i), bit for bit; the values are
  //      positive floats, which order like their bit patterns.
  if (d.collisions && d.n_rays > 0 && !(ablate & 1)) {
    stamp(11);
    uint16_t* queue = (uint16_t*)const_cast<int*>(st_pair_index + A * A);  // lane | sensor << 6 | target << 11
    uint32_t* measured = (uint32_t*)(st_pair_index + A * A) + A * (A - 1) * 32;
    const float lim = d.lidar_range + d.agent_radius + 1e-4f;
    // (eight pairs at a time: their tests first, then ONE LDS atomic for the wave's slots - an atomic and its round trip
    // per pair was a third of this phase)
    const int n_unordered = A * (A - 1) / 2;
    for (int k0 = C.wave; k0 < n_unordered; k0 += 8 * C.nw) {
      unsigned long long bal[8];
      int sensor[8], target[8], total = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j * C.nw;
        bal[j] = 0ull; sensor[j] = target[j] = 0;
        if (k >= n_unordered) continue;
        int a = 0, rem = k;
        while (rem >= A - 1 - a) { rem -= A - 1 - a; ++a; }
        const int t = a + 1 + rem;
        const float* pa = col + (d.agent0 + a) * 6 * 64;
        const float* pt = col + (d.agent0 + t) * 6 * 64;
        const float dx = pt[0] - pa[0], dy = pt[64] - pa[64];
        const bool near = C.live && !(dx * dx + dy * dy > lim * lim);  // NaN counts as near
        bal[j] = __ballot(near);
        sensor[j] = a; target[j] = t;
        total += 2 * __popcll(bal[j]);
      }
      if (total == 0) continue;
      int base = 0;
      if (C.lane == 0) base = atomicAdd(&misc[0], total);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (bal[j] == 0ull) continue;
        if ((bal[j] >> C.lane) & 1ull) {
          const int slot = base + 2 * (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[j], 0u));
          queue[slot] = (uint16_t)(C.lane | sensor[j] << 6 | target[j] << 11);      // sensor a on target t ...
          queue[slot + 1] = (uint16_t)(C.lane | target[j] << 6 | sensor[j] << 11);  // ... and sensor t on target a
        }
        base += 2 * __popcll(bal[j]);
      }
    }
    stamp(12);
    __syncthreads();
    stamp(14);
    const int n_items = misc[0];
    const float radius = d.agent_radius, range = d.lidar_range, half_range = d.lidar_range * 0.5f;
    for (int i0 = C.wave * 64; i0 < n_items; i0 += C.nw * 64) {
      const bool on = i0 + C.lane < n_items;
      const uint32_t item = on ? queue[i0 + C.lane] : 0u;
      const int e = item & 63, sa = (item >> 6) & 31, ta = item >> 11;
      const float* ps = rows + e + (d.agent0 + sa) * 6 * 64;
      const float* pg = rows + e + (d.agent0 + ta) * 6 * 64;
      const v2 op = V(ps[0], ps[64]), tpos = V(pg[0], pg[64]);
      const float arot = ps[4 * 64];
      const v2 u = tpos - op;
      const float margin = radius + 1e-4f + 1e-5f * (fabsf(op.x) + fabsf(op.y) + fabsf(tpos.x) + fabsf(tpos.y));
      const bool table = __all(!on || arot == 0.f);  // (see lidar_cast_chunk: the same sincosf made the table)
      auto direction = [&](int r, float& c, float& sn) {
        if (table) { const float2 cs = st_cs[sa * d.n_rays + r]; c = cs.x; sn = cs.y; }
        else sincosf(st_angles[sa * d.n_rays + r] + arot, &sn, &c);  // sensors.py:118
      };
      unsigned long long cand = 0ull;
      for (int r = 0; r < d.n_rays; ++r) {
        float c, sn;
        direction(r, c, sn);
        const bool maybe = !(fabsf(u.x * sn - u.y * c) > margin) && vdot(u, V(c, sn)) > 0.f;
        if (on && maybe) cand |= 1ull << r;
      }
      uint32_t* mrow = measured + sa * d.n_rays * kNavMeasuredStride + e;
      while (__any(cand != 0ull)) {
        if (cand != 0ull) {
          const int r = __ffsll((long long)cand) - 1;
          cand &= cand - 1ull;
          float c, sn;
          direction(r, c, sn);
          const v2 dir = V(c, sn);  // _cast_rays_to_sphere core.py:1414-1490
          const v2 lp = V(op.x + dir.x * half_range, op.y + dir.y * half_range);
          const v2 cp = closest_point_line<false>(lp, c, sn, 0.f, tpos);
          const float dn = vnorm(tpos - cp);
          const bool ok = (dn < radius) && (vdot(u, dir) > 0.f);
          const float a2 = radius * radius - dn * dn;
          const float m = sqrt_n(a2 > 0.f ? a2 : 1e-8f);
          const float dist = vnorm(cp - op) - m;
          if (ok && dist < range) atomicMax(mrow + r * kNavMeasuredStride, __float_as_uint(range - dist));
        }
      }
    }
    stamp(8);
    __syncthreads();  // every item's atomic max has landed: the observation writer reads `measured`
    if (!grid_sync) {
      if (blockIdx.x == 0 && nav.mask_clear != nullptr && (int)threadIdx.x < words) nav.mask_clear[threadIdx.x] = 0u;
      // (the mask only grows during a step: a stale read is harmless - and a thousand tiles hammering one word is not)
      if ((int)threadIdx.x < words) {
        const uint32_t b = (uint32_t)misc[1 + threadIdx.x];
        if (b != 0u && (b & ~__builtin_nontemporal_load(&nav.mask[threadIdx.x])) != 0u) atomicOr(&nav.mask[threadIdx.x], b);
      }
    }
  }
  // grid-barrier form: the batch's World.collides bits are collected BEHIND the observation writer (navigation_post_body
  // calls this between the observations and the collision penalties): the tile arrived in front of the LIDAR units, the
  // round trips of the barrier - an atomic to the memory side and a poll back, ~2.6 us each - run beside the LIDAR units
  // and the observation blocks, and only the rewards wait for them
  auto collect = [&]() {
    if (!(grid_sync && d.collisions && d.n_rays > 0 && !(ablate & 1))) return;
    uint32_t* gmask = (uint32_t*)&misc[1];  // (LDS) the batch's pair words, once every tile has arrived
    if (C.wave == 0) {
      // every tile that has arrived at barrier seq is done with the slots up to seq - 1; the one of seq - 2 = seq + 2
      // (mod 4) is cleared by block 0 here - no tile writes it before it has passed barrier seq + 1, which this wave has
      // yet to join
      unsigned long long* base = (unsigned long long*)(nav.sync + 2);
      grid_bits_collect(base + (size_t)(seq & 3u) * words * groups, base + (size_t)((seq + 2u) & 3u) * words * groups, words,
                        gmask, nav.sync + 1, nav.gave_up);
    }
    __syncthreads();
    if ((int)threadIdx.x < A) {  // bit j of collide_with[a]: World.collides(agent a, agent j)
      uint32_t m = 0;
      for (int j = 0; j < A; ++j) {
        const int pi = st_pair_index[threadIdx.x * A + j];
        if (j != (int)threadIdx.x && pi >= 0 && ((gmask[pi >> 5] >> (pi & 31)) & 1u)) m |= 1u << j;
      }
      collide_with[threadIdx.x] = m;
    }
    __syncthreads();
  };
  stamp(9);
  // agent a's 64 x D block of the observation matrix (navigation.py:244-263) as one contiguous run: every lane gathers two
  // consecutive elements (one if D is odd) - the columns the body just put into `own`, the rays from `measured`
  auto rays = [&](int a, int, int part, int nparts) {
    wave_lds_fence();
    float* out = o.obs + ((long)a * batch + C.b0) * D;
    const int total = C.n_rows * D;
    const float* mr = (const float*)((const uint32_t*)(st_pair_index + A * A) + A * (A - 1) * 32) + a * d.n_rays * kNavMeasuredStride;
    const float inv_d = 1.f / (float)D;
    auto elem = [&](int env, int k) {  // (one LDS read whichever array holds the column)
      const float* src = k < ray0 ? own_cols + k * kNavMeasuredStride : mr + (k - ray0) * kNavMeasuredStride;
      return src[env];
    };
    if ((D & 1) == 0) {  // (every block starts 8-byte aligned: 64 * D * 4 bytes per tile, batch * D * 4 per agent)
      for (int i = 2 * C.lane + 128 * part; i < total; i += 128 * nparts) {
        const int env = (int)(((float)i + 0.5f) * inv_d), k = i - env * D;
        *(float2*)(out + i) = make_float2(elem(env, k), elem(env, k + 1));
      }
    } else {
      for (int i = C.lane + 64 * part; i < total; i += 64 * nparts) {
        const int env = (int)(((float)i + 0.5f) * inv_d);
        out[i] = elem(env, i - env * D);
      }
    }
    wave_lds_fence();
  };
  float shaping_out[kNavMaxOwn] = {};
  if (ablate & 2) {}
  else if (grid_sync && d.collisions)
    navigation_post_body<false>(C, d, o, batch, per_agent, collide_with, T, steps_in, pos, vel, goal, rays, shaping_out, collect);
  else
    navigation_post_body<true>(C, d, o, batch, per_agent, nullptr, T, steps_in, pos, vel, goal, rays, shaping_out);
  stamp(10);
  if (o.limit.steps != nullptr) steps_in = steps_in + 1.f;
  if (!last) {  // the next step of the rollout: its previous shaping = this one's (from the registers of the wave that owns
                // the agent), the work counter and the pair words re-armed
    __syncthreads();
    float* pa = scratch;
#pragma unroll
    for (int sl = 0; sl < kNavMaxOwn; ++sl) {
      const int a = C.wave + sl * C.nw;
      if (a < A) pa[a * 64 + C.lane] = C.live ? shaping_out[sl] : 0.f;
    }
    if (threadIdx.x < VMAS_ENV_MAX_AGENTS) misc[threadIdx.x] = 0;
    if (d.collisions) {
      uint32_t* measured = (uint32_t*)(st_pair_index + A * A) + A * (A - 1) * 32;
      for (int i = threadIdx.x; i < R * kNavMeasuredStride; i += C.nw * 64) measured[i] = 0u;
    }
  }
}

// VMAS_ACTION_ERR_* into the caller's flag word: system scope - the word may live in pinned host memory (vmas_host_word_create)
VD void raise_action_error(uint32_t* err, uint32_t bad) {
  __hip_atomic_fetch_or(err, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------ action ingest
// Environment._set_action (environment.py:616-749, continuous branch) + Holonomic(.WithRotation)
// .process_action for ONE agent slot and this lane's environment: returns the (up to 3) scaled
// action components, stores agent_ft / u_out, and ORs VMAS_ACTION_ERR_* into `bad`.
// `row0`: first row of this step's actions in the caller's [n_steps * batch, action_size] tensor (multi-step rollouts).
// Two halves, so that a wave with several agents has all their action loads in flight before it touches the first:
// ingest_fetch issues the loads of one slot (nothing else), ingest_apply is the arithmetic and the stores.  A load per
// agent behind the previous agent's stores chained one HBM round trip per agent along the step kernel's critical path.
struct IngestRaw { float u[3]; long flat; };
// Pointers out of a slot that went through registers (load_action_slot) have lost the "loaded from the kernel arguments: global
// memory" inference and would be dereferenced with FLAT instructions - which count on lgkmcnt too, so that the next wait for a
// scalar load or an LDS read would also wait for the action load.  These casts say what they are.
// (VMAS_GLOBAL / as_global: defined in front of the lazy exact broad phase's helpers above)
// An agent's action slot out of the kernel-argument block, whole and at once: read field by field where it is used, every
// field is its own scalar load with its own wait in front of the branch that needs it - some thirty dependent round trips
// to the scalar cache on the prologue's chain.  18 words in one batch of wide loads, pinned in scalar registers.
VD VmasActionSlot load_action_slot(const VmasActionSlot* p) {
  static_assert(sizeof(VmasActionSlot) == 72, "load_action_slot reads the slot as 18 words");
  const uint32_t* w = (const uint32_t*)p;
  uint32_t r[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) r[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[i]);  // (uniform by construction; said so)
  asm volatile("" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]), "+s"(r[5]), "+s"(r[6]), "+s"(r[7]), "+s"(r[8]),
               "+s"(r[9]), "+s"(r[10]), "+s"(r[11]), "+s"(r[12]), "+s"(r[13]), "+s"(r[14]), "+s"(r[15]), "+s"(r[16]), "+s"(r[17]));
  VmasActionSlot s;
  __builtin_memcpy(&s, r, sizeof(s));
  return s;
}
VD void ingest_fetch(const VmasActionSlot& S, long env, bool live, long row0, IngestRaw& r) {
  r.u[0] = r.u[1] = r.u[2] = 0.f;
  r.flat = 0;
  if (!live) return;
  if (S.action_index != nullptr) {
    r.flat = as_global(S.action_index)[row0 + env];
  } else if (S.action_size == 2 && ((uintptr_t)S.action & 7) == 0) {  // (rows of two floats: one 8-byte load)
    typedef float vf2 __attribute__((ext_vector_type(2)));  // (float2 is a class: no copy out of a qualified address space)
    const vf2 v = *as_global((const vf2*)(S.action + (row0 + env) * 2));
    r.u[0] = v.x; r.u[1] = v.y;
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k)  // (constant bounds: a run-time trip count would index r.u dynamically - a stack slot)
      if (k < S.action_size) r.u[k] = as_global(S.action)[(row0 + env) * S.action_size + k];
  }
}
VD void ingest_apply(const VmasActionSlot& S, int clamp, long env, bool live, float* __restrict__ agent_ft, long ld,
                     float u_out[3], uint32_t& bad, const IngestRaw& r) {
  u_out[0] = u_out[1] = u_out[2] = 0.f;
  if (!live) return;
  long flat = r.flat;
  if (S.action_index != nullptr) {  // flat index -> per-dimension index -> [-u_range, u_range] (environment.py:657-705)
    long total = 1;
#pragma unroll
    for (int k = 0; k < 3; ++k)  // (constant bounds, here and below: a run-time trip count would index the slot dynamically -
      if (k < S.action_size) total *= S.nvec[k];  //  a slot held in registers, load_action_slot, would move to the stack)
    if (flat < 0 || flat >= total) {
      bad |= VMAS_ACTION_ERR_OUT_OF_RANGE;
      flat = 0;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k >= S.action_size) break;
    float u;
    if (S.action_index != nullptr) {
      long m = 1;
#pragma unroll
      for (int j = 1; j < 3; ++j)
        if (j > k && j < S.action_size) m *= S.nvec[j];
      const int n = S.nvec[k];
      int a = (int)(flat / m);
      flat = flat % m;
      if (n & 1) a = a == 0 ? n / 2 : (a <= n / 2 ? a - 1 : a);  // odd count: index 0 is "stay"
      u = ((float)a / (float)(n - 1)) * (2.f * S.u_range[k]) - S.u_range[k];
    } else {
      u = r.u[k];
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
    if (S.u_out != nullptr) as_global(S.u_out)[env * S.action_size + k] = u;
  }
}

VD void ingest_slot(const VmasActionSlot& S, int clamp, long env, bool live, float* __restrict__ agent_ft, long ld,
                    float u_out[3], uint32_t& bad, long row0 = 0) {
  IngestRaw r;
  ingest_fetch(S, env, live, row0, r);
  ingest_apply(S, clamp, env, live, agent_ft, ld, u_out, bad, r);
}

// A scripted agent whose script the library knows (VmasAgentScript): same contract as ingest_slot.  `E` = this
// environment's column of the script's entity - field f at E[f * stride] - in HBM (stride = ld) or in a step kernel's LDS
// tile (stride = 64: the state about to be stepped of a multi-step launch never left LDS).
VD void run_script(const VmasAgentScript& S, const float* E, long stride, long env, bool live,
                   float* __restrict__ agent_ft, long ld, float u_out[3]) {
  u_out[0] = u_out[1] = u_out[2] = 0.f;
  if (!live) return;
  if (S.kind == VMAS_SCRIPT_FOOTBALL_BALL) {  // ball_action_script football.py:1620-1680
    const float x = E[0], y = E[stride], vy = E[3 * stride];
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

// ------------------------------------------------------------------------------------ football
// football.py:1121-1515 (learning-vs-learning game) on a tile whose agent rows are reachable through `G(slot, k)`
// (slot = blue agents, red agents, ball; k = px py vx vy fx fy): the stand-alone kernel (vmas_env.hip) stages them from
// HBM, the compact step kernel (vmas_compact.h) reads its own tile - the post-step as the physics kernel's epilogue.
// An observation is 16 + 8 * (observed others) floats - 88 for 5 v 5, 3.5 KB per environment and step over the ten
// agents: a streaming writer, and HOW a store instruction's lanes cover the tensor decides the rate HBM takes it at
// (scripts/micro/store_pattern.hip, 461 MB at 131 072 environments: every lane its own row 16 bytes at a time 3.6 TB/s,
// 64- or 128-byte pieces of the rows 3.1-3.2 TB/s, the rows as they lie - 1 024 contiguous bytes per instruction - 5.85 TB/s).
// For one agent the rows of a tile are ONE contiguous run of 64 * D floats, so:
//   SHARED (chunk < 0; the stand-alone kernel): `slab` is the TILE's [R = -chunk][D + 2] array.  All waves take the agents
//     one after the other: each wave computes 8-column groups (the own/ball columns are two, every observed other is one)
//     for the R rows of the pass - lanes outside the pass do not write -, block barrier, the block's threads store the
//     pass's run of R * D floats as it lies (R a multiple of 4: whole 128-byte lines), block barrier.  EVERY wave of the
//     tile must call.  As the step kernel's epilogue this form was measured and dropped (profiles/r04h_football_
//     observation_staging_ab.jsonl): a tile in a K-step launch is bound by its own chain, and 60 barriers per step lengthen it
//     (131 072 environments: 210 us per step against 183; one step per launch 260 against 287 - but see vmas_hip.hip: two
//     launches per step take 172).
//   chunk > 0: `slab` is THIS wave's [64][chunk + 1] tile (chunk = 16 or 32 columns) - the wave transposes its agent's
//     observation `chunk` columns at a time; the lanes of a store cover 64- or 128-byte pieces of the rows.  No block
//     barrier: the latency regime's form (16 waves per tile, one agent per wave, nothing to wait for).
//   slab_off < 0: no staging, every lane stores its own row 16 bytes at a time.
// `prev`: the four shaping terms of this lane
// (in: before, out: after this step), `steps_in`: wave 0's Environment.steps (in/out), `stp`: step of a multi-step
// launch (every per-step output is offset by stp slabs).
constexpr int kChunk = 32;               // (chunk > 0 forms: the widest chunk)
constexpr int kFootballStageChunk = 16;  // the step kernel's per-wave form: 64-byte pieces, a quarter of the LDS
__host__ __device__ inline size_t football_shared_slab_floats(int rows, int D) { return (size_t)rows * (D + 2); }

template <bool SHARED = false, class Get>  // SHARED: the chunk < 0 form is compiled in (the stand-alone kernel; not the step kernel's epilogue)
VD void football_post_tile(const TileCtx& C, const VmasFootballDesc& d, const VmasFootballBuffers& o_in, int batch, Get G,
                           int slab_off, int chunk, float (&prev)[4], float& steps_in, int stp) {
  // (slab_off: the staging array's float offset from the launch's dynamic LDS base, < 0 = none.  As a nullable POINTER into LDS
  //  its null test went through the generic address space, which trips a back-end bug of ROCm 7.2 in some instantiations -
  //  "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
  extern __shared__ float football_lds_base[];
  const bool staged = slab_off >= 0;
  float* slab = football_lds_base + (staged ? slab_off : 0);
  const int n = d.n_blue + d.n_red, ball = n;  // slot of the ball
  VmasFootballBuffers o = o_in;
  const int n_adv_b = d.observe_adversaries ? d.n_red : 0, n_adv_r = d.observe_adversaries ? d.n_blue : 0;
  const int D0 = 16 + 8 * (n_adv_b + (d.observe_teammates ? d.n_blue - 1 : 0));  // (host-checked: the same for both teams)
  if (stp > 0) {
    o.obs += (long)stp * n * batch * D0; o.rew += (long)stp * n * batch; o.done += (long)stp * batch;
    o.terms += (long)stp * 9 * batch; o.touching += (long)stp * 2 * batch;
  }
  float* my_row = slab + C.lane * (chunk + 1);  // (used only if staged)
  auto P2 = [&](int slot, int k) { return V(G(slot, k), G(slot, k + 1)); };
  const v2 bpos = P2(ball, 0), bvel = P2(ball, 2), bforce = P2(ball, 4);

  // ---- reward football.py:1121-1219 (wave 0 only stores; every wave needs the two team rewards)
  const bool over_right = bpos.x > d.goal_x, over_left = bpos.x < -d.goal_x;
  const bool in_mouth = bpos.y <= d.goal_half && bpos.y >= -d.goal_half;
  const bool blue_score = over_right && in_mouth, red_score = over_left && in_mouth;
  const float sparse_blue = d.scoring_reward * (blue_score ? 1.f : 0.f) - d.scoring_reward * (red_score ? 1.f : 0.f);
  float dense[2] = {0.f, 0.f}, term[8];
  if (d.dense_reward) {
    const bool ball_moving = vnorm(bvel) > 1e-6f;
#pragma unroll
    for (int team = 0; team < 2; ++team) {  // 0 blue (attacks the right goal), 1 red
      const v2 goal = V(team == 0 ? d.goal_x : -d.goal_x, 0.f);
      const float shaping = vnorm(bpos - goal) * d.pos_shaping_factor_ball_goal;  // reward_ball_to_goal
      float min_dist = kInf;                                                       // reward_all_agent_to_ball
      const int a0 = team == 0 ? 0 : d.n_blue, a1 = team == 0 ? d.n_blue : n;
      for (int a = a0; a < a1; ++a) min_dist = min_t(min_dist, vnorm(P2(a, 0) - bpos));
      const float shaping_agent = min_dist * d.pos_shaping_factor_agent_ball;
      term[team] = shaping; term[2 + team] = shaping_agent; term[4 + team] = min_dist;
      term[6 + team] = prev[team] - shaping;                                       // ball.pos_rew_<team>
      const bool quiet = (min_dist < d.distance_to_ball_trigger) || ball_moving;
      const float rew_agent = quiet ? 0.f : prev[2 + team] - shaping_agent;
      dense[team] = term[6 + team] + rew_agent;
      if (C.wave == 0 && C.live) {
        o.pos_shaping[(long)team * batch + C.env] = shaping;
        o.pos_shaping[(long)(2 + team) * batch + C.env] = shaping_agent;
        o.terms[(long)(1 + team) * batch + C.env] = term[6 + team];
        o.terms[(long)(3 + team) * batch + C.env] = rew_agent;
        o.terms[(long)(5 + team) * batch + C.env] = min_dist;
        o.terms[(long)(7 + team) * batch + C.env] = shaping / d.pos_shaping_factor_ball_goal;
        o.touching[(long)team * batch + C.env] = min_dist <= d.touch_dist ? 1 : 0;
      }
      prev[team] = shaping;  // (every wave computed them: the next step of a multi-step launch starts from these)
      prev[2 + team] = shaping_agent;
    }
  }
  if (C.wave == 0) {
    const bool done = apply_step_limit(o.limit, C, steps_in, blue_score || red_score);
    if (C.live) {
      o.terms[C.env] = sparse_blue;
      o.done[C.env] = done ? 1 : 0;
    }
    if (o.limit.steps != nullptr) steps_in = steps_in + 1.f;
  }
  const float rew_team[2] = {sparse_blue + dense[0], (0.f - sparse_blue) + dense[1]};

  // ---- observation football.py:1221-1460; red agents see everything mirrored in x.  The three forms of the header:
  //      the tile's waves on one agent at a time through the shared array (contiguous runs) ...
  if constexpr (SHARED) {
    const int R = -chunk, tid = C.wave * 64 + C.lane, nthreads = C.nw * 64;
    for (int a = 0; a < n; ++a) {
      const bool blue = a < d.n_blue;
      const float sx = blue ? 1.f : -1.f;
      auto M = [&](v2 v) { return V(v.x * sx, v.y); };
      const v2 goal = V(blue ? d.goal_x : -d.goal_x, 0.f);
      const int n_adv = blue ? n_adv_b : n_adv_r;
      const int mate0 = blue ? 0 : d.n_blue, n_team = blue ? d.n_blue : d.n_red;
      const int n_others = n_adv + (d.observe_teammates ? n_team - 1 : 0);
      const int D = 16 + 8 * n_others, S = D + 2, w4 = D >> 2, n_groups = D >> 3;
      const float inv = 1.f / (float)w4;
      float* out = o.obs + ((long)a * batch + C.b0) * D;
      for (int r0 = 0; r0 < C.n_rows; r0 += R) {
        const int r = C.lane - r0;
        if (r >= 0 && r < R) {
          const v2 pos = P2(a, 0), vel = P2(a, 2);
          for (int g = C.wave; g < n_groups; g += C.nw) {
            v2 q0, q1, q2, q3;
            if (g == 0) {
              q0 = M(P2(a, 4)); q1 = M(pos - bpos); q2 = M(vel - bvel); q3 = M(bpos - goal);
            } else if (g == 1) {
              q0 = M(bvel); q1 = M(bforce); q2 = M(pos - goal); q3 = M(vel);
            } else {
              const int j = g - 2;
              int other;
              if (j < n_adv) {
                other = (blue ? d.n_blue : 0) + j;  // the other team, in order
              } else {
                other = mate0 + (j - n_adv);
                if (other >= a) other += 1;         // my team, skipping myself
              }
              const v2 opos = P2(other, 0), ovel = P2(other, 2);
              q0 = M(pos - opos); q1 = M(vel - ovel); q2 = M(ovel); q3 = M(P2(other, 4));
            }
            float2* dst = (float2*)(slab + r * S + 8 * g);  // (8-byte aligned: S and the slab's offset are even)
            dst[0] = make_float2(q0.x, q0.y); dst[1] = make_float2(q1.x, q1.y);
            dst[2] = make_float2(q2.x, q2.y); dst[3] = make_float2(q3.x, q3.y);
          }
        }
        __syncthreads();
        const int rows = C.n_rows - r0 < R ? C.n_rows - r0 : R, total4 = rows * w4;
        float4* run = (float4*)(out + (long)r0 * D);
        for (int i = tid; i < total4; i += nthreads) {
          int rr = (int)((float)i * inv);
          int c4 = i - rr * w4;
          if (c4 >= w4) { c4 -= w4; rr += 1; }
          if (c4 < 0) { c4 += w4; rr -= 1; }
          const float2* src = (const float2*)(slab + rr * S + 4 * c4);
          const float2 lo = src[0], hi = src[1];
          run[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        __syncthreads();
      }
      if (a % C.nw == C.wave && C.live) o.rew[(long)a * batch + C.env] = rew_team[blue ? 0 : 1];
    }
    return;
  }
  // ... or agents wave, wave + nw, ...: chunk by chunk - the 16 own/ball columns, then the observed others chunk / 8 at a
  // time - through the wave's LDS tile (a chunk leaves as float4 stores), or every lane its own row
  for (int a = C.wave; a < n; a += C.nw) {
    const bool blue = a < d.n_blue;
    const float sx = blue ? 1.f : -1.f;
    auto M = [&](v2 v) { return V(v.x * sx, v.y); };
    const v2 goal = V(blue ? d.goal_x : -d.goal_x, 0.f);
    const v2 pos = P2(a, 0), vel = P2(a, 2), force = P2(a, 4);
    const int n_adv = blue ? n_adv_b : n_adv_r;
    const int mate0 = blue ? 0 : d.n_blue, n_team = blue ? d.n_blue : d.n_red;
    const int n_others = n_adv + (d.observe_teammates ? n_team - 1 : 0);
    const int D = 16 + 8 * n_others;
    if (!staged) {
      // no LDS staging (a step kernel whose LDS has no room for it): every lane writes its environment's row itself, four
      // columns per store - 16 bytes per lane at a stride of the row length
      float* row = o.obs + ((long)a * batch + C.env) * D;
      auto put4 = [&](int c, v2 p, v2 q) { if (C.live) *(float4*)(row + c) = make_float4(p.x, p.y, q.x, q.y); };
      put4(0, M(force), M(pos - bpos)); put4(4, M(vel - bvel), M(bpos - goal));
      put4(8, M(bvel), M(bforce)); put4(12, M(pos - goal), M(vel));
      for (int j = 0; j < n_others; ++j) {
        int other;
        if (j < n_adv) {
          other = (blue ? d.n_blue : 0) + j;  // the other team, in order
        } else {
          other = mate0 + (j - n_adv);
          if (other >= a) other += 1;         // my team, skipping myself
        }
        const v2 opos = P2(other, 0), ovel = P2(other, 2), oforce = P2(other, 4);
        put4(16 + 8 * j, M(pos - opos), M(vel - ovel));
        put4(20 + 8 * j, M(ovel), M(oforce));
      }
      if (C.live) o.rew[(long)a * batch + C.env] = rew_team[blue ? 0 : 1];
      continue;
    }
    float* out = o.obs + ((long)a * batch + C.b0) * D;
    auto put = [&](int c, v2 v) { my_row[c] = v.x; my_row[c + 1] = v.y; };
    // the chunk [c0, c0 + w) of the tile -> out; w is a multiple of 8
    auto flush = [&](int c0, int w) {
      wave_lds_fence();
      const int w4 = w >> 2, total4 = C.n_rows * w4;
      const float inv = 1.f / (float)w4;
      for (int i0 = C.lane; i0 < total4; i0 += 128) {
        float4 v[2];
        int dst[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = i0 + 64 * k < total4 ? i0 + 64 * k : total4 - 1;
          int r = (int)((float)i * inv);
          int c4 = i - r * w4;
          if (c4 >= w4) { c4 -= w4; r += 1; }
          if (c4 < 0) { c4 += w4; r -= 1; }
          const float* src = slab + r * (chunk + 1) + 4 * c4;
          v[k] = make_float4(src[0], src[1], src[2], src[3]);
          dst[k] = r * D + c0 + 4 * c4;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (i0 + 64 * k < total4) *(float4*)(out + dst[k]) = v[k];
      }
      wave_lds_fence();
    };
    put(0, M(force)); put(2, M(pos - bpos)); put(4, M(vel - bvel)); put(6, M(bpos - goal));
    put(8, M(bvel)); put(10, M(bforce)); put(12, M(pos - goal)); put(14, M(vel));
    flush(0, 16);
    const int per = chunk >> 3;  // observed others per chunk
    for (int j0 = 0; j0 < n_others; j0 += per) {
      const int m = n_others - j0 < per ? n_others - j0 : per;
      for (int jj = 0; jj < m; ++jj) {
        const int j = j0 + jj;
        int other;
        if (j < n_adv) {
          other = (blue ? d.n_blue : 0) + j;  // the other team, in order
        } else {
          other = mate0 + (j - n_adv);
          if (other >= a) other += 1;         // my team, skipping myself
        }
        const v2 opos = P2(other, 0), ovel = P2(other, 2), oforce = P2(other, 4);
        put(8 * jj, M(pos - opos)); put(8 * jj + 2, M(vel - ovel)); put(8 * jj + 4, M(ovel)); put(8 * jj + 6, M(oforce));
      }
      flush(16 + 8 * j0, 8 * m);
    }
    if (C.live) o.rew[(long)a * batch + C.env] = rew_team[blue ? 0 : 1];
  }
}

}  // namespace vmas
