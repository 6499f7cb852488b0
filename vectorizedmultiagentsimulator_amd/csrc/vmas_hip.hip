// vmas_hip.hip - MI355X (gfx950 / CDNA4) implementation of the C ABI in include/vmas_hip.h.
//
// One fused kernel advances ALL substeps of World.step() (core.py:1972-2015) for a tile
// of environments:
//
//   HBM (SoA planes, env fastest)  --coalesced row copy-->  LDS tile [row][env]
//   per substep, out of LDS:   A  entity lanes : trig + action/friction/gravity prologue
//                              B  task lanes   : joints, then pairs (narrow phase +
//                                                penalty force), ds_add_f32 into the
//                                                per-entity force/torque accumulators
//                              C  entity lanes : semi-implicit Euler + clamps
//   LDS tile  --coalesced row copy (dynamic rows only)-->  HBM
//
// Work decomposition: G lanes cooperate on one environment (template parameter, power
// of two, 1..64).  A 64-wide wavefront therefore carries 64/G environments; lane l owns
// environment (l % (64/G)) of its wave and is "worker" g = l / (64/G) inside it.
// Entities (phases A, C) and tasks (phase B) are dealt round-robin to the G workers.
// Every environment of a batch has the SAME static world, so lanes with equal g execute
// the same task type in lock-step: divergence only arises between the G workers, and the
// task list is type-major (joints, SS, LS, LL, BS, BL, BB - also the reference's
// accumulation order) so neighbouring workers mostly share a type too.
//
// No MFMA: the path is fp32 elementwise/transcendental work with a 2-vector inner
// dimension; nothing here is a contraction (BASELINE.json north_star).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/vmas_hip.h"
#include "vmas_device.h"

using namespace vmas;

// ------------------------------------------------------------------------------------
// device-side constant block
// ------------------------------------------------------------------------------------
enum : uint32_t {
  GATE_A_MOV = 1u << 0, GATE_A_ROT = 1u << 1, GATE_B_MOV = 1u << 2, GATE_B_ROT = 1u << 3,
  GATE_A_HOLLOW = 1u << 4, GATE_B_HOLLOW = 1u << 5, GATE_LOCK = 1u << 6 /* joint: rotate == False */
};

// One unit of phase-B work, everything static pre-resolved on the host (fp32 ops done
// exactly as the reference does them at run time).
struct DevTask {
  int32_t a, b;
  int32_t type;   // VMAS_PAIR_* or TASK_JOINT
  uint32_t gate;  // GATE_*
  float thr2;     // pairs: (R_a + R_b + LINE_MIN_DIST + slack)^2, per-env conservative skip
  float p0, p1, p2, p3;  // type-specific dims, see build_tasks()
  float q0, q1;          // joints: delta_b
  int32_t index;         // pair index (mask bit) or joint index (per-env fixed_rot row)
};
constexpr int TASK_JOINT = 6;

struct DevEntity {
  uint32_t flags;
  int32_t shape;
  int32_t agent_index;
  float mass, inertia, one_minus_drag;
  float max_speed, v_range, lin_friction, ang_friction;
  float gx, gy;  // constant entity gravity
  float max_f, f_range, max_t, t_range;
};

struct DevWorld {
  int32_t nE, nA, nT, nJ, substeps;
  float sub_dt, gx, gy;
  int32_t has_gravity;
  float xs, ys;  // NaN = unbounded
  float k, tcf;  // contact_margin, torque_constraint_force
  float c_coll, c_joint_att, c_joint_rep;  // fp32(sign * force_multiplier)
  const DevEntity* ent;
  const DevTask* task;
};

struct DevStepArgs {
  const uint32_t* pair_mask;
  const float* joint_fixed_rot;
  const float* entity_gravity;
  int32_t first_substep, n_substeps;
};

// rows of the LDS tile
struct Tile {
  float* st;  // [nE*6][EPB]  pos.x pos.y vel.x vel.y rot ang_vel
  float* af;  // [nA*3][EPB]  force.x force.y torque
  float* tr;  // [nE*4][EPB]  cos(rot) sin(rot) cos(rot+pi/2) sin(rot+pi/2)
  float* fa;  // [nE*3][EPB]  force.x force.y torque accumulators
  int* bad;   // [EPB]        env has a non-finite pos/rot: no broad-phase skipping (NaN parity)
};

// ------------------------------------------------------------------------------------
// the fused step kernel
// ------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void step_kernel(DevWorld W, float* __restrict__ state,
                                                   float* __restrict__ agent_ft, long ld, int batch,
                                                   DevStepArgs args) {
  constexpr int EPB = 256 / G;  // environments per 256-thread block
  constexpr int EPW = 64 / G;   // environments per wavefront
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int nE = W.nE, nA = W.nA;
  Tile T;
  T.st = lds;
  T.af = T.st + nE * 6 * EPB;
  T.tr = T.af + nA * 3 * EPB;
  T.fa = T.tr + nE * 4 * EPB;
  T.bad = (int*)(T.fa + nE * 3 * EPB);
  const long env0 = (long)blockIdx.x * EPB;

  // ---- HBM -> LDS: every wave instruction reads one contiguous run of a plane ----
  {
    const int col = tid % EPB, r0 = tid / EPB;
    const long env = env0 + col;
    const bool ok = env < batch;
    for (int r = r0; r < nE * 6; r += G) T.st[r * EPB + col] = ok ? state[r * ld + env] : 0.f;
    for (int r = r0; r < nA * 3; r += G) T.af[r * EPB + col] = ok ? agent_ft[r * ld + env] : 0.f;
    if (r0 == 0) T.bad[col] = 0;
  }
  __syncthreads();

  const int lane = tid & 63, wv = tid >> 6;
  const int el = wv * EPW + (lane % EPW);  // environment column of this lane inside the tile
  const int g = lane / EPW;                // worker index inside the environment
  const long env = env0 + el;
  const float sub_dt = W.sub_dt, k = W.k;
#define ST(e, f) T.st[((e) * 6 + (f)) * EPB + el]
#define AF(a, f) T.af[((a) * 3 + (f)) * EPB + el]
#define TR(e, f) T.tr[((e) * 4 + (f)) * EPB + el]
#define FA(e, f) T.fa[((e) * 3 + (f)) * EPB + el]

  const int s_begin = args.first_substep;
  const int s_end = s_begin + (args.n_substeps > 0 ? args.n_substeps : W.substeps - s_begin);
  for (int substep = s_begin; substep < s_end; ++substep) {
    // ================= phase A: per-entity trig + force prologue (core.py:1976-2004)
    for (int e = g; e < nE; e += G) {
      const DevEntity D = W.ent[e];
      const uint32_t fl = D.flags;
      {  // the reference lets a non-finite entity poison every pair it is in, however far
         // apart (cos(inf) = NaN): such environments must not use the distance skip
        const float px = ST(e, 0), py = ST(e, 1), rt = ST(e, 4);
        if (!(fabsf(px) < kInf) || !(fabsf(py) < kInf) || !(fabsf(rt) < kInf)) T.bad[el] = 1;
      }
      if (D.shape != VMAS_SHAPE_SPHERE) {  // the only trig the narrow phase needs
        const float rot = ST(e, 4);
        TR(e, 0) = cosf(rot);
        TR(e, 1) = sinf(rot);
        if (D.shape == VMAS_SHAPE_BOX) {
          const float rot2 = rot + kHalfPi;  // physics.py:301
          TR(e, 2) = cosf(rot2);
          TR(e, 3) = sinf(rot2);
        }
      }
      v2 F = V(0.f, 0.f);
      float Tq = 0.f;
      if (fl & VMAS_F_AGENT) {
        const int a = D.agent_index;
        if (fl & VMAS_F_MOVABLE) {  // _apply_action_force core.py:2018-2028
          v2 f = V(AF(a, 0), AF(a, 1));
          if (fl & VMAS_F_MAX_F) f = clamp_with_norm(f, D.max_f);
          if (fl & VMAS_F_F_RANGE) f = V(clamp_t(f.x, D.f_range), clamp_t(f.y, D.f_range));
          if (fl & (VMAS_F_MAX_F | VMAS_F_F_RANGE)) { AF(a, 0) = f.x; AF(a, 1) = f.y; }
          F = F + f;
        }
        if (fl & VMAS_F_ROTATABLE) {  // _apply_action_torque core.py:2030-2041
          float t = AF(a, 2);
          if (fl & VMAS_F_MAX_T) {
            const float n = fabsf(t);
            const float nt = (t / n) * D.max_t;
            t = n > D.max_t ? nt : t;
          }
          if (fl & VMAS_F_T_RANGE) t = clamp_t(t, D.t_range);
          if (fl & (VMAS_F_MAX_T | VMAS_F_T_RANGE)) AF(a, 2) = t;
          Tq = Tq + t;
        }
      }
      // _apply_friction_force core.py:2054-2102
      if (fl & VMAS_F_LIN_FRICTION) F = F + friction2(V(ST(e, 2), ST(e, 3)), D.lin_friction, D.mass, sub_dt);
      if (fl & VMAS_F_ANG_FRICTION) Tq = Tq + friction1(ST(e, 5), D.ang_friction, D.inertia, sub_dt);
      // _apply_gravity core.py:2043-2052
      if (fl & VMAS_F_MOVABLE) {
        if (W.has_gravity) F = F + V(D.mass * W.gx, D.mass * W.gy);
        if (fl & VMAS_F_GRAVITY) {
          v2 ge = V(D.gx, D.gy);
          if (args.entity_gravity && env < batch) {
            const float* gp = args.entity_gravity + (long)e * 2 * ld + env;
            ge = V(gp[0], gp[ld]);
          }
          F = F + V(D.mass * ge.x, D.mass * ge.y);
        }
      }
      FA(e, 0) = F.x; FA(e, 1) = F.y; FA(e, 2) = Tq;
    }
    __syncthreads();

    // ================= phase B: joints, then collision pairs (core.py:2104-2189)
    const bool may_skip = T.bad[el] == 0;
    for (int ti = g; ti < W.nT; ti += G) {
      const DevTask K = W.task[ti];
      const int a = K.a, b = K.b;
      const v2 pa = V(ST(a, 0), ST(a, 1)), pb = V(ST(b, 0), ST(b, 1));
      v2 fa = V(0.f, 0.f), fb = V(0.f, 0.f);
      float ta = 0.f, tb = 0.f;
      if (K.type == TASK_JOINT) {  // _vectorized_joint_constraints core.py:2201-2292
        const float ra = ST(a, 4), rb = ST(b, 4);
        const v2 pja = pa + rotate(V(K.p0, K.p1), cosf(ra), sinf(ra));  // joints.py:209-216
        const v2 pjb = pb + rotate(V(K.q0, K.q1), cosf(rb), sinf(rb));
        const v2 f_att = constraint_force<true>(pja, pjb, K.p2, W.c_joint_att, k);
        const v2 f_rep = constraint_force<false>(pja, pjb, K.p2, W.c_joint_rep, k);
        fa = f_att + f_rep;
        fb = (-f_att) + (-f_rep);
        ta = vcross(pja - pa, fa);
        tb = vcross(pjb - pb, fb);
        if (K.gate & GATE_LOCK) {
          float fr = K.p3;
          if (args.joint_fixed_rot && env < batch) fr = args.joint_fixed_rot[(long)K.index * ld + env];
          const float t = constraint_torque(ra, rb + fr, W.tcf);
          ta = ta + (-t);
          tb = tb + t;
        }
      } else {
        if (args.pair_mask && !((args.pair_mask[K.index >> 5] >> (K.index & 31)) & 1u)) continue;
        {  // per-environment conservative broad phase: beyond this no force can be non-zero
          const float dx = pa.x - pb.x, dy = pa.y - pb.y;
          if (may_skip && dx * dx + dy * dy > K.thr2) continue;
        }
        switch (K.type) {
          case VMAS_PAIR_SS: {  // core.py:2294-2339; p0 = r_a + r_b
            fa = constraint_force<false>(pa, pb, K.p0, W.c_coll, k);
            fb = -fa;
          } break;
          case VMAS_PAIR_LS: {  // a = line, b = sphere; p0 = L/2, p1 = r + LMD  core.py:2341-2392
            const v2 cp = closest_point_line<true>(pa, TR(a, 0), TR(a, 1), K.p0, pb);
            fb = constraint_force<false>(pb, cp, K.p1, W.c_coll, k);
            fa = -fb;
            ta = vcross(cp - pa, fa);
          } break;
          case VMAS_PAIR_LL: {  // p0 = La/2, p1 = Lb/2  core.py:2394-2457
            seg_t l1 = {pa, TR(a, 0), TR(a, 1), K.p0};
            seg_t l2 = {pb, TR(b, 0), TR(b, 1), K.p1};
            v2 qa, qb;
            closest_points_seg_seg(l1, l2, qa, qb);
            fa = constraint_force<false>(qa, qb, kLineMinDist, W.c_coll, k);
            fb = -fa;
            ta = vcross(qa - pa, fa);
            tb = vcross(qb - pb, fb);
          } break;
          case VMAS_PAIR_BS: {  // a = box, b = sphere; p0 = L, p1 = W, p2 = r + LMD  core.py:2459-2552
            seg_t be[4];
            box_edges(pa, TR(a, 0), TR(a, 1), TR(a, 2), TR(a, 3), K.p0, K.p1, be);
            const v2 cp = closest_point_box(be, pb);
            v2 ip = cp;
            float d = 0.f;
            if (!(K.gate & GATE_A_HOLLOW)) ip = inner_point_box(pb, cp, pa, d);
            fb = constraint_force<false>(pb, ip, K.p2 + d, W.c_coll, k);
            fa = -fb;
            ta = vcross(cp - pa, fa);
          } break;
          case VMAS_PAIR_BL: {  // a = box, b = line; p0 = L, p1 = W, p2 = Lb/2  core.py:2554-2653
            seg_t be[4];
            box_edges(pa, TR(a, 0), TR(a, 1), TR(a, 2), TR(a, 3), K.p0, K.p1, be);
            seg_t ln = {pb, TR(b, 0), TR(b, 1), K.p2};
            v2 qb, ql;
            closest_seg_box(be, ln, qb, ql);
            v2 ip = qb;
            float d = 0.f;
            if (!(K.gate & GATE_A_HOLLOW)) ip = inner_point_box(ql, qb, pa, d);
            fa = constraint_force<false>(ip, ql, kLineMinDist + d, W.c_coll, k);
            fb = -fa;
            ta = vcross(qb - pa, fa);
            tb = vcross(ql - pb, fb);
          } break;
          case VMAS_PAIR_BB: {  // p0,p1 = L,W of a; p2,p3 = L,W of b  core.py:2655-2786
            seg_t ea[4], eb[4];
            box_edges(pa, TR(a, 0), TR(a, 1), TR(a, 2), TR(a, 3), K.p0, K.p1, ea);
            box_edges(pb, TR(b, 0), TR(b, 1), TR(b, 2), TR(b, 3), K.p2, K.p3, eb);
            v2 qa, qb;
            closest_box_box(ea, eb, qa, qb);
            v2 ia = qa, ib = qb;
            float da = 0.f, db = 0.f;
            if (!(K.gate & GATE_A_HOLLOW)) ia = inner_point_box(qb, qa, pa, da);
            if (!(K.gate & GATE_B_HOLLOW)) ib = inner_point_box(qa, qb, pb, db);
            fa = constraint_force<false>(ia, ib, da + db + kLineMinDist, W.c_coll, k);
            fb = -fa;
            ta = vcross(qa - pa, fa);
            tb = vcross(qb - pb, fb);
          } break;
          default: break;
        }
      }
      // update_env_forces core.py:2191-2199 (LDS float atomics: several workers of one
      // environment may hit the same entity in the same round)
      if (K.gate & GATE_A_MOV) { atomicAdd(&FA(a, 0), fa.x); atomicAdd(&FA(a, 1), fa.y); }
      if (K.gate & GATE_A_ROT) atomicAdd(&FA(a, 2), ta);
      if (K.gate & GATE_B_MOV) { atomicAdd(&FA(b, 0), fb.x); atomicAdd(&FA(b, 1), fb.y); }
      if (K.gate & GATE_B_ROT) atomicAdd(&FA(b, 2), tb);
    }
    __syncthreads();

    // ================= phase C: _integrate_state core.py:2862-2908
    for (int e = g; e < nE; e += G) {
      const DevEntity D = W.ent[e];
      const uint32_t fl = D.flags;
      if (fl & VMAS_F_MOVABLE) {
        v2 vel = V(ST(e, 2), ST(e, 3));
        if (substep == 0) vel = V(vel.x * D.one_minus_drag, vel.y * D.one_minus_drag);
        const v2 acc = V(FA(e, 0) / D.mass, FA(e, 1) / D.mass);
        vel = V(vel.x + acc.x * sub_dt, vel.y + acc.y * sub_dt);
        if (fl & VMAS_F_MAX_SPEED) vel = clamp_with_norm(vel, D.max_speed);
        if (fl & VMAS_F_V_RANGE) vel = V(clamp_t(vel.x, D.v_range), clamp_t(vel.y, D.v_range));
        v2 np = V(ST(e, 0) + vel.x * sub_dt, ST(e, 1) + vel.y * sub_dt);
        if (W.xs == W.xs) np.x = clamp_t(np.x, W.xs);
        if (W.ys == W.ys) np.y = clamp_t(np.y, W.ys);
        ST(e, 0) = np.x; ST(e, 1) = np.y; ST(e, 2) = vel.x; ST(e, 3) = vel.y;
      }
      if (fl & VMAS_F_ROTATABLE) {
        float av = ST(e, 5);
        if (substep == 0) av = av * D.one_minus_drag;
        av = av + (FA(e, 2) / D.inertia) * sub_dt;
        ST(e, 4) = ST(e, 4) + av * sub_dt;
        ST(e, 5) = av;
      }
    }
    __syncthreads();
  }

  // ---- LDS -> HBM: only the planes the reference rebinds (core.py:2871-2908, 2021-2039)
  {
    const int col = tid % EPB, r0 = tid / EPB;
    const long envc = env0 + col;
    if (envc < batch) {
      for (int r = r0; r < nE * 6; r += G) {
        const uint32_t fl = W.ent[r / 6].flags;
        const bool dyn = (r % 6 < 4) ? (fl & VMAS_F_MOVABLE) : (fl & VMAS_F_ROTATABLE);
        if (dyn) state[r * ld + envc] = T.st[r * EPB + col];
      }
      for (int e = 0; e < nE; ++e) {
        const uint32_t fl = W.ent[e].flags;
        if (!(fl & VMAS_F_AGENT)) continue;
        const int a = W.ent[e].agent_index;
        const bool wf = (fl & VMAS_F_MOVABLE) && (fl & (VMAS_F_MAX_F | VMAS_F_F_RANGE));
        const bool wt = (fl & VMAS_F_ROTATABLE) && (fl & (VMAS_F_MAX_T | VMAS_F_T_RANGE));
        for (int f = r0; f < 3; f += G) {
          if (f < 2 ? wf : wt) agent_ft[(a * 3 + f) * ld + envc] = T.af[(a * 3 + f) * EPB + col];
        }
      }
    }
  }
#undef ST
#undef AF
#undef TR
#undef FA
}

// ------------------------------------------------------------------------------------
// batch-global broad phase (World.collides core.py:2797-2801)
// ------------------------------------------------------------------------------------
struct DevMaskPair { int32_t a, b; float bound_sum; };

__global__ __launch_bounds__(256) void pair_mask_kernel(const DevMaskPair* __restrict__ pairs, int nP,
                                                        const float* __restrict__ state, long ld, int batch,
                                                        uint32_t* __restrict__ mask) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = env < batch;
  for (int p = 0; p < nP; ++p) {
    const DevMaskPair P = pairs[p];
    bool hit = false;
    if (ok) {
      const float* sa = state + (long)P.a * 6 * ld + env;
      const float* sb = state + (long)P.b * 6 * ld + env;
      hit = norm2(sa[0] - sb[0], sa[ld] - sb[ld]) <= P.bound_sum;
    }
    if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(&mask[p >> 5], 1u << (p & 31));
  }
}

// ------------------------------------------------------------------------------------
// LIDAR (World.cast_rays core.py:1662-1786): one thread per (environment, sensor)
// ------------------------------------------------------------------------------------
struct DevLidar {
  int32_t entity, n_rays, n_targets, target_off, angle_off;
  float max_range, half_range;
};
struct DevTarget { int32_t entity, shape; float length, width, radius; };

constexpr int RAY_CHUNK = 8;

__global__ __launch_bounds__(256) void lidar_kernel(const DevLidar* __restrict__ lidars,
                                                    const DevTarget* __restrict__ targets,
                                                    const float* __restrict__ angles, int max_rays,
                                                    const float* __restrict__ state, long ld, int batch,
                                                    float* __restrict__ out) {
  const long env = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= batch) return;
  const int l = blockIdx.y;
  const DevLidar L = lidars[l];
  const float* sp = state + (long)L.entity * 6 * ld + env;
  const v2 o = V(sp[0], sp[ld]);
  const float arot = sp[4 * ld];
  const float R = L.max_range;
  for (int r0 = 0; r0 < L.n_rays; r0 += RAY_CHUNK) {
    float best[RAY_CHUNK], c[RAY_CHUNK], s[RAY_CHUNK];
#pragma unroll
    for (int i = 0; i < RAY_CHUNK; ++i) {
      const int r = r0 + i < L.n_rays ? r0 + i : L.n_rays - 1;
      const float th = angles[L.angle_off + r] + arot;  // sensors.py:118
      c[i] = cosf(th);
      s[i] = sinf(th);
      best[i] = R;  // core.py:1672-1674
    }
    for (int ti = 0; ti < L.n_targets; ++ti) {
      const DevTarget Tg = targets[L.target_off + ti];
      const float* tp = state + (long)Tg.entity * 6 * ld + env;
      const v2 tpos = V(tp[0], tp[ld]);
      if (Tg.shape == VMAS_SHAPE_SPHERE) {  // _cast_rays_to_sphere core.py:1414-1490
        const v2 u = tpos - o;
#pragma unroll
        for (int i = 0; i < RAY_CHUNK; ++i) {
          const v2 dir = V(c[i], s[i]);
          const v2 lp = V(o.x + dir.x * L.half_range, o.y + dir.y * L.half_range);
          const v2 cp = closest_point_line<false>(lp, c[i], s[i], 0.f, tpos);
          const float dn = vnorm(tpos - cp);
          const float a = Tg.radius * Tg.radius - dn * dn;
          const float m = __fsqrt_rn(a > 0.f ? a : 1e-8f);
          float dist = vnorm(cp - o) - m;
          const bool ok = (dn < Tg.radius) && (vdot(u, dir) > 0.f);
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
#pragma unroll
    for (int i = 0; i < RAY_CHUNK; ++i)
      if (r0 + i < L.n_rays) out[((long)l * max_rays + r0 + i) * ld + env] = best[i];
  }
}

// ------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
#define HIP_TRY(x)                                                                      \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct VmasWorld {
  int device = 0;
  int batch = 0;
  int lanes = 4;
  DevWorld dw{};
  DevEntity* d_ent = nullptr;
  DevTask* d_task = nullptr;
  DevMaskPair* d_mpairs = nullptr;
  int n_pairs = 0;
  int n_dyn = 0;
  size_t lds_rows = 0;
  // lidars
  DevLidar* d_lidars = nullptr;
  DevTarget* d_targets = nullptr;
  float* d_angles = nullptr;
  int n_lidars = 0, max_rays = 0;
  std::vector<VmasEntityDesc> ents;
};

static float slack_thr2(float bound_sum) {
  // forces vanish once shapes are farther apart than LINE_MIN_DIST (see DESIGN.md,
  // "per-environment broad phase"); 1e-3 absorbs rounding of the closest-point math.
  float t = bound_sum + kLineMinDist + 1e-3f;
  return t * t;
}

static void build_tasks(const VmasWorldDesc* d, std::vector<DevTask>& tasks) {
  const VmasEntityDesc* E = d->entities;
  auto gate_of = [&](int a, int b) {
    uint32_t g = 0;
    if (E[a].flags & VMAS_F_MOVABLE) g |= GATE_A_MOV;
    if (E[a].flags & VMAS_F_ROTATABLE) g |= GATE_A_ROT;
    if (E[b].flags & VMAS_F_MOVABLE) g |= GATE_B_MOV;
    if (E[b].flags & VMAS_F_ROTATABLE) g |= GATE_B_ROT;
    if (E[a].flags & VMAS_F_HOLLOW) g |= GATE_A_HOLLOW;
    if (E[b].flags & VMAS_F_HOLLOW) g |= GATE_B_HOLLOW;
    return g;
  };
  for (int j = 0; j < d->n_joints; ++j) {
    const VmasJointDesc& J = d->joints[j];
    DevTask t{};
    t.a = J.a; t.b = J.b; t.type = TASK_JOINT; t.index = j;
    t.gate = gate_of(J.a, J.b) | (J.rotate ? 0u : GATE_LOCK);
    t.p0 = J.delta_a[0]; t.p1 = J.delta_a[1]; t.q0 = J.delta_b[0]; t.q1 = J.delta_b[1];
    t.p2 = J.dist; t.p3 = J.fixed_rotation;
    tasks.push_back(t);
  }
  for (int p = 0; p < d->n_pairs; ++p) {
    const VmasPairDesc& P = d->pairs[p];
    DevTask t{};
    t.a = P.a; t.b = P.b; t.type = P.type; t.index = p;
    t.gate = gate_of(P.a, P.b);
    t.thr2 = slack_thr2(P.bound_sum);
    const VmasEntityDesc &A = E[P.a], &B = E[P.b];
    switch (P.type) {
      case VMAS_PAIR_SS: {  // force is exactly 0 for dist > r_a + r_b (core.py:2836)
        t.p0 = A.radius + B.radius;
        float m = t.p0 + 1e-4f;
        t.thr2 = m * m;
      } break;
      case VMAS_PAIR_LS: t.p0 = A.length / 2.f; t.p1 = B.radius + kLineMinDist; break;
      case VMAS_PAIR_LL: t.p0 = A.length / 2.f; t.p1 = B.length / 2.f; break;
      case VMAS_PAIR_BS: t.p0 = A.length; t.p1 = A.width; t.p2 = B.radius + kLineMinDist; break;
      case VMAS_PAIR_BL: t.p0 = A.length; t.p1 = A.width; t.p2 = B.length / 2.f; break;
      case VMAS_PAIR_BB: t.p0 = A.length; t.p1 = A.width; t.p2 = B.length; t.p3 = B.width; break;
      default: break;
    }
    tasks.push_back(t);
  }
}

static int default_lanes(int n_tasks, int n_entities, int batch) {
  // enough lanes per environment to (a) put >= ~4 waves on each of the 1024 SIMDs and
  // (b) keep the serial task chain per lane short; tuned on MI355X.
  int work = n_tasks + n_entities;
  int g = 1;
  while (g < 64 && (g * 6 < work || (long)batch * g < 4096L * 64)) g <<= 1;
  return g;
}

template <int G>
static int launch_step(VmasWorld* w, float* state, float* aft, long ld, const DevStepArgs& a, hipStream_t s) {
  constexpr int EPB = 256 / G;
  const size_t lds = w->lds_rows * EPB * sizeof(float);
  if (lds > 64 * 1024) {
    static thread_local size_t set_for = 0;
    if (set_for < lds) {
      HIP_TRY(hipFuncSetAttribute((const void*)step_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set_for = lds;
    }
  }
  const int blocks = (w->batch + EPB - 1) / EPB;
  hipLaunchKernelGGL(step_kernel<G>, dim3(blocks), dim3(256), lds, s, w->dw, state, aft, ld, w->batch, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" {

int vmas_abi_version(void) { return VMAS_ABI_VERSION; }
const char* vmas_last_error(void) { return g_err; }

int vmas_world_create(const VmasWorldDesc* d, int32_t batch, int32_t device_id, VmasWorld** out) {
  if (!d || !out) return fail("vmas_world_create: null argument");
  if (d->abi_version != VMAS_ABI_VERSION)
    return fail("vmas_world_create: desc ABI version %d, library %d", d->abi_version, VMAS_ABI_VERSION);
  if (batch <= 0) return fail("vmas_world_create: batch must be > 0, got %d", batch);
  if (d->n_entities <= 0) return fail("vmas_world_create: world has no entities");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail("vmas_world_create: device %d of %d", device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));

  VmasWorld* w = new VmasWorld();
  w->device = device_id;
  w->batch = batch;
  w->ents.assign(d->entities, d->entities + d->n_entities);
  w->n_pairs = d->n_pairs;

  std::vector<DevEntity> ents(d->n_entities);
  for (int e = 0; e < d->n_entities; ++e) {
    const VmasEntityDesc& s = d->entities[e];
    DevEntity& t = ents[e];
    t.flags = s.flags; t.shape = s.shape; t.agent_index = s.agent_index;
    t.mass = s.mass; t.inertia = s.inertia; t.one_minus_drag = s.one_minus_drag;
    t.max_speed = s.max_speed; t.v_range = s.v_range;
    t.lin_friction = s.lin_friction; t.ang_friction = s.ang_friction;
    t.gx = s.gravity[0]; t.gy = s.gravity[1];
    t.max_f = s.max_f; t.f_range = s.f_range; t.max_t = s.max_t; t.t_range = s.t_range;
    if (s.flags & (VMAS_F_MOVABLE | VMAS_F_ROTATABLE)) w->n_dyn++;
    if ((s.flags & VMAS_F_AGENT) && (s.agent_index < 0 || s.agent_index >= d->n_agents)) {
      delete w;
      return fail("vmas_world_create: entity %d has agent_index %d outside [0,%d)", e, s.agent_index, d->n_agents);
    }
  }
  std::vector<DevTask> tasks;
  build_tasks(d, tasks);
  std::vector<DevMaskPair> mp(d->n_pairs);
  for (int p = 0; p < d->n_pairs; ++p) mp[p] = {d->pairs[p].a, d->pairs[p].b, d->pairs[p].bound_sum};

  auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  HIP_TRY(upload((void**)&w->d_ent, ents.data(), ents.size() * sizeof(DevEntity)));
  HIP_TRY(upload((void**)&w->d_task, tasks.data(), tasks.size() * sizeof(DevTask)));
  HIP_TRY(upload((void**)&w->d_mpairs, mp.data(), mp.size() * sizeof(DevMaskPair)));

  DevWorld& W = w->dw;
  W.nE = d->n_entities; W.nA = d->n_agents; W.nT = (int)tasks.size(); W.nJ = d->n_joints;
  W.substeps = d->substeps; W.sub_dt = d->sub_dt;
  W.gx = d->gravity[0]; W.gy = d->gravity[1]; W.has_gravity = d->has_gravity;
  W.xs = d->x_semidim; W.ys = d->y_semidim;
  W.k = d->contact_margin; W.tcf = d->torque_constraint_force;
  W.c_coll = d->collision_force;      // sign = +1
  W.c_joint_att = -d->joint_force;    // sign = -1
  W.c_joint_rep = d->joint_force;
  W.ent = w->d_ent; W.task = w->d_task;
  w->lds_rows = (size_t)W.nE * (6 + 4 + 3) + (size_t)W.nA * 3 + 1;
  w->lanes = default_lanes(W.nT, W.nE, batch);
  // a tile must fit the 160 KiB LDS of a CU (64 KiB is the default dynamic limit)
  while (w->lanes < 64 && w->lds_rows * (256 / w->lanes) * sizeof(float) > 64 * 1024) w->lanes <<= 1;
  *out = w;
  return 0;
}

void vmas_world_destroy(VmasWorld* w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  (void)hipFree(w->d_ent); (void)hipFree(w->d_task); (void)hipFree(w->d_mpairs);
  (void)hipFree(w->d_lidars); (void)hipFree(w->d_targets); (void)hipFree(w->d_angles);
  delete w;
}

int vmas_world_set_lanes_per_env(VmasWorld* w, int32_t lanes) {
  if (!w) return fail("vmas_world_set_lanes_per_env: null world");
  if (lanes == 0) lanes = default_lanes(w->dw.nT, w->dw.nE, w->batch);
  if (lanes < 1 || lanes > 64 || (lanes & (lanes - 1))) return fail("lanes_per_env must be a power of two in 1..64, got %d", lanes);
  if (w->lds_rows * (256 / lanes) * sizeof(float) > 160 * 1024)
    return fail("lanes_per_env=%d needs %zu B of LDS per block (> 160 KiB)", lanes, w->lds_rows * (256 / lanes) * sizeof(float));
  w->lanes = lanes;
  return 0;
}
int vmas_world_get_lanes_per_env(const VmasWorld* w) { return w ? w->lanes : -1; }

int64_t vmas_world_step_bytes_per_env(const VmasWorld* w) {
  if (!w) return -1;
  return 24LL * w->dw.nE + 12LL * w->dw.nA + 24LL * w->n_dyn;
}

int vmas_world_step(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args, void* stream) {
  if (!w || !state) return fail("vmas_world_step: null argument");
  if (w->dw.nA > 0 && !agent_ft) return fail("vmas_world_step: world has agents but agent_ft is null");
  if (ld < w->batch) return fail("vmas_world_step: ld (%lld) < batch (%d)", (long long)ld, w->batch);
  DevStepArgs a{};
  a.n_substeps = 0;
  if (args) {
    a.pair_mask = args->pair_mask; a.joint_fixed_rot = args->joint_fixed_rot; a.entity_gravity = args->entity_gravity;
    a.first_substep = args->first_substep; a.n_substeps = args->n_substeps;
    if (a.first_substep < 0 || a.first_substep >= w->dw.substeps)
      return fail("vmas_world_step: first_substep %d outside [0,%d)", a.first_substep, w->dw.substeps);
  }
  hipStream_t s = (hipStream_t)stream;
  switch (w->lanes) {
    case 1: return launch_step<1>(w, state, agent_ft, ld, a, s);
    case 2: return launch_step<2>(w, state, agent_ft, ld, a, s);
    case 4: return launch_step<4>(w, state, agent_ft, ld, a, s);
    case 8: return launch_step<8>(w, state, agent_ft, ld, a, s);
    case 16: return launch_step<16>(w, state, agent_ft, ld, a, s);
    case 32: return launch_step<32>(w, state, agent_ft, ld, a, s);
    case 64: return launch_step<64>(w, state, agent_ft, ld, a, s);
  }
  return fail("vmas_world_step: bad lanes_per_env %d", w->lanes);
}

int vmas_world_step_n(VmasWorld* w, float* state, float* agent_ft, int64_t ld, int64_t ft_step_stride, int32_t n_steps,
                      const VmasStepArgs* args, void* stream) {
  if (n_steps < 0) return fail("vmas_world_step_n: n_steps %d < 0", n_steps);
  for (int i = 0; i < n_steps; ++i) {
    int rc = vmas_world_step(w, state, agent_ft ? agent_ft + (int64_t)i * ft_step_stride : nullptr, ld, args, stream);
    if (rc) return rc;
  }
  return 0;
}

int vmas_world_pair_mask(VmasWorld* w, const float* state, int64_t ld, uint32_t* mask, void* stream) {
  if (!w || !state || !mask) return fail("vmas_world_pair_mask: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int words = (w->n_pairs + 31) / 32;
  HIP_TRY(hipMemsetAsync(mask, 0, sizeof(uint32_t) * (words ? words : 1), s));
  if (w->n_pairs == 0) return 0;
  hipLaunchKernelGGL(pair_mask_kernel, dim3((w->batch + 255) / 256), dim3(256), 0, s, w->d_mpairs, w->n_pairs, state,
                     (long)ld, w->batch, mask);
  HIP_TRY(hipGetLastError());
  return 0;
}

int vmas_world_set_lidars(VmasWorld* w, const VmasLidarDesc* lidars, int32_t n) {
  if (!w || (n > 0 && !lidars)) return fail("vmas_world_set_lidars: null argument");
  HIP_TRY(hipSetDevice(w->device));
  (void)hipFree(w->d_lidars); (void)hipFree(w->d_targets); (void)hipFree(w->d_angles);
  w->d_lidars = nullptr; w->d_targets = nullptr; w->d_angles = nullptr;
  w->n_lidars = 0; w->max_rays = 0;
  if (n <= 0) return 0;
  std::vector<DevLidar> dl(n);
  std::vector<DevTarget> dt;
  std::vector<float> da;
  for (int i = 0; i < n; ++i) {
    const VmasLidarDesc& L = lidars[i];
    if (L.entity < 0 || L.entity >= w->dw.nE || L.n_rays <= 0) return fail("vmas_world_set_lidars: bad sensor %d", i);
    dl[i] = {L.entity, L.n_rays, L.n_targets, (int)dt.size(), (int)da.size(), L.max_range,
             (float)((double)L.max_range / 2.0)};
    for (int t = 0; t < L.n_targets; ++t) {
      int e = L.targets[t];
      if (e < 0 || e >= w->dw.nE) return fail("vmas_world_set_lidars: sensor %d target %d out of range", i, e);
      const VmasEntityDesc& E = w->ents[e];
      dt.push_back({e, E.shape, E.length, E.width, E.radius});
    }
    for (int r = 0; r < L.n_rays; ++r) da.push_back(L.angles[r]);
    if (L.n_rays > w->max_rays) w->max_rays = L.n_rays;
  }
  HIP_TRY(hipMalloc((void**)&w->d_lidars, dl.size() * sizeof(DevLidar)));
  HIP_TRY(hipMemcpy(w->d_lidars, dl.data(), dl.size() * sizeof(DevLidar), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&w->d_targets, (dt.size() ? dt.size() : 1) * sizeof(DevTarget)));
  if (!dt.empty()) HIP_TRY(hipMemcpy(w->d_targets, dt.data(), dt.size() * sizeof(DevTarget), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&w->d_angles, da.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(w->d_angles, da.data(), da.size() * sizeof(float), hipMemcpyHostToDevice));
  w->n_lidars = n;
  return 0;
}

int vmas_world_cast_rays(VmasWorld* w, const float* state, int64_t ld, float* out, void* stream) {
  if (!w || !state || !out) return fail("vmas_world_cast_rays: null argument");
  if (w->n_lidars <= 0) return fail("vmas_world_cast_rays: no sensors registered (vmas_world_set_lidars)");
  hipLaunchKernelGGL(lidar_kernel, dim3((w->batch + 255) / 256, w->n_lidars), dim3(256), 0, (hipStream_t)stream,
                     w->d_lidars, w->d_targets, w->d_angles, w->max_rays, state, (long)ld, w->batch, out);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
