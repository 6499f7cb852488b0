/*
 * vmas_oracle.c - CPU restatement of the VMAS 1.5.2 physics step and LIDAR ray cast.
 *
 * TEST INFRASTRUCTURE ONLY.  Imported/linked by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the CHECKER of the HIP path - never shipped, never
 * a fallback: the product (vectorizedmultiagentsimulator_amd) fails loudly when its
 * HIP library is missing and has no code path into this file.
 *
 * Parity pinning: the reference holds no golden vectors for this path (SURVEY.md
 * section 8c).  This restatement is pinned against outputs of the reference itself,
 * generated in the build container by tests/golden/make_golden.py (imports
 * /root/reference unmodified) and committed as tests/golden/*.npz; see
 * tests/test_oracle_golden.py.
 *
 * One environment at a time, scalar fp32, FMA contraction OFF (build flag
 * -ffp-contract=off); the only fused operation is the 2-vector norm, because
 * torch's CPU `linalg.vector_norm` over a size-2 dim is bitwise
 * sqrt(fma(y, y, x*x)) (probed on torch 2.10 CPU, all shapes used by the path).
 *
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/vmas/simulator/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vmas_hip.h"

#define LINE_MIN_DIST ((float)(4.0 / 6e2)) /* utils.py:28 */
#define HALF_PI_F ((float)(3.14159265358979323846 / 2.0)) /* torch.pi / 2 as fp32 scalar */
#define MAX_E 64

/* ---- libm jitter (tolerance calibration only) -------------------------------------
 * torch's CPU kernels use SLEEF (<=1 ulp), this file glibc, the HIP path ocml: three
 * correct libms that differ in the last bit.  With a non-zero jitter seed every
 * sin/cos/exp/log1p result is moved by -1/0/+1 ulp pseudo-randomly, which lets the
 * tests MEASURE how far two correct fp32 implementations can drift apart on a given
 * input (tests/golden_util.ulp_sensitivity) instead of guessing a tolerance. */
static uint32_t g_jitter = 0;
void vmas_oracle_set_jitter(uint32_t seed) { g_jitter = seed; }
static inline float jit(float r, float x) {
  if (!g_jitter || !(r == r) || isinf(r)) return r;
  uint32_t b;
  memcpy(&b, &x, 4);
  b = (b ^ g_jitter) * 2654435761u;
  b ^= b >> 15; b *= 2246822519u; b ^= b >> 13;
  uint32_t k = b % 3u;
  return k == 0 ? r : nextafterf(r, k == 1 ? INFINITY : -INFINITY);
}
#define cosf(x) jit(cosf(x), (x))
#define sinf(x) jit(sinf(x), (x) + 1.f)
#define expf(x) jit(expf(x), (x))
#define log1pf(x) jit(log1pf(x), (x))

typedef struct { float x, y; } v2;

static inline v2 V(float x, float y) { v2 r = {x, y}; return r; }
static inline v2 vadd(v2 a, v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline v2 vscale(v2 a, float s) { return V(a.x * s, a.y * s); }
static inline v2 vneg(v2 a) { return V(-a.x, -a.y); }
/* torch.linalg.vector_norm(dim=-1) on a size-2 dim (see header comment) */
static inline float norm2(float x, float y) { return sqrtf(fmaf(y, y, x * x)); }
static inline float vnorm(v2 a) { return norm2(a.x, a.y); }
/* (a * b).sum(-1): two rounded products, one rounded add */
static inline float vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
/* TorchUtils.cross utils.py:193-197 */
static inline float vcross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
/* torch.sign: sign(0) = 0, sign(nan) = nan */
static inline float signf_(float x) { return x != x ? x : (float)((x > 0.f) - (x < 0.f)); }
/* torch.minimum / torch.maximum / torch.min/max reductions propagate NaN */
static inline float tmin(float a, float b) { return a != a ? a : (b != b ? b : (a < b ? a : b)); }
static inline float tmax(float a, float b) { return a != a ? a : (b != b ? b : (a > b ? a : b)); }
/* torch.clamp(x, -r, r) */
static inline float clampf(float x, float r) { return x != x ? x : (x < -r ? -r : (x > r ? r : x)); }

/* TorchUtils.rotate_vector utils.py:175-191 */
static inline v2 rotate(v2 v, float c, float s) { return V(v.x * c - v.y * s, v.x * s + v.y * c); }

/* TorchUtils.clamp_with_norm utils.py:167-173 */
static inline v2 clamp_with_norm(v2 t, float max_norm) {
  float n = vnorm(t);
  v2 nt = V((t.x / n) * max_norm, (t.y / n) * max_norm);
  return n > max_norm ? nt : t;
}

/* torch.logaddexp(0, x) (aten LogAddExp CPU kernel): max(0,x) + log1p(exp(-|0-x|)) */
static inline float softplus0(float x) {
  if (isinf(x) && x == 0.f) return x; /* unreachable, kept for symmetry with aten */
  float m = tmax(0.f, x);
  return m + log1pf(expf(-fabsf(0.f - x)));
}

/* World._get_constraint_forces core.py:2805-2839; returns the force on a (force on b = -f) */
static v2 constraint_force(v2 pa, v2 pb, float dist_min, float force_multiplier, float k, int attractive) {
  const float min_dist = 1e-6f;
  v2 d = vsub(pa, pb);
  float dist = vnorm(d);
  float sign = attractive ? -1.f : 1.f;
  float pen = softplus0((dist_min - dist) * sign / k) * k;
  float c = (float)((double)sign * (double)force_multiplier); /* python: sign * force_multiplier */
  float den = dist > 0.f ? dist : 1e-8f;
  v2 f = V(c * d.x / den * pen, c * d.y / den * pen);
  if (dist < min_dist) f = V(0.f, 0.f);
  if (!attractive) { if (dist > dist_min) f = V(0.f, 0.f); }
  else             { if (dist < dist_min) f = V(0.f, 0.f); }
  return f;
}

/* World._get_constraint_torques core.py:2841-2858; returns torque on b (torque on a = -t) */
static float constraint_torque(float rot_a, float rot_b, float force_multiplier) {
  float delta = rot_a - rot_b;
  float ad = fabsf(delta);
  float pen = expf(ad) - 1.f;
  float t = force_multiplier * signf_(delta) * pen;
  if (ad < 1e-9f) t = 0.f;
  return t;
}

/* physics._get_closest_point_line physics.py:400-429; (c, s) = cos/sin(line_rot) */
static inline v2 closest_point_line(v2 pos, float c, float s, float half_len, v2 p, int limit) {
  v2 d = vsub(pos, p);
  float dot = d.x * c + d.y * s;
  float sg = signf_(dot);
  float m = fabsf(dot);
  if (limit) m = tmin(m, half_len);
  float t = sg * m;
  return V(pos.x - t * c, pos.y - t * s);
}

/* A line segment: centre, cos/sin of its rotation, half length */
typedef struct { v2 pos; float c, s, half; } seg_t;

/* physics._get_all_lines_box physics.py:298-325 */
static void box_edges(v2 pos, float rot, float length, float width, seg_t e[4]) {
  float c = cosf(rot), s = sinf(rot);
  float rot2 = rot + HALF_PI_F;
  float c2 = cosf(rot2), s2 = sinf(rot2);
  float hl = length / 2.f, hw = width / 2.f;
  e[0].pos = V(pos.x + c * hl, pos.y + s * hl);
  e[1].pos = V(pos.x - c * hl, pos.y - s * hl);
  e[2].pos = V(pos.x + c2 * hw, pos.y + s2 * hw);
  e[3].pos = V(pos.x - c2 * hw, pos.y - s2 * hw);
  e[0].c = e[1].c = c2; e[0].s = e[1].s = s2; e[0].half = e[1].half = hw;
  e[2].c = e[3].c = c;  e[2].s = e[3].s = s;  e[2].half = e[3].half = hl;
}

/* physics._get_closest_point_box physics.py:263-295 (+ _get_all_points_box 385-397) */
static v2 closest_point_box(const seg_t e[4], v2 p) {
  v2 best = V(INFINITY, INFINITY);
  float dist = INFINITY;
  for (int i = 0; i < 4; ++i) {
    v2 q = closest_point_line(e[i].pos, e[i].c, e[i].s, e[i].half, p, 1);
    float d = vnorm(vsub(p, q));
    if (d < dist) { best = q; dist = d; }
  }
  return best;
}

/* physics._get_inner_point_box physics.py:13-23 */
static v2 inner_point_box(v2 outside, v2 surface, v2 box_pos, float* depth) {
  v2 v = vsub(surface, outside);
  v2 u = vsub(box_pos, surface);
  float n = vnorm(v);
  float xm = vdot(v, u) / n;
  v2 x = V((v.x / n) * xm, (v.y / n) * xm);
  if (n == 0.f) { x = surface; xm = 0.f; }
  *depth = fabsf(xm);
  return vadd(surface, x);
}

/* physics._get_closest_points_line_line physics.py:144-219
 * (+ _get_line_extrema 132-141, _get_intersection_point_line_line 222-260) */
static void closest_points_seg_seg(const seg_t* l1, const seg_t* l2, v2* p1, v2* p2) {
  v2 xy1 = V(l1->half * l1->c, l1->half * l1->s);
  v2 xy2 = V(l2->half * l2->c, l2->half * l2->s);
  v2 a1 = vadd(l1->pos, xy1), a2 = vsub(l1->pos, xy1);
  v2 b1 = vadd(l2->pos, xy2), b2 = vsub(l2->pos, xy2);
  /* intersection */
  v2 r = vsub(a2, a1), s = vsub(b2, b1), qp = vsub(b1, a1);
  float cqpr = vcross(qp, r), cqps = vcross(qp, s), crs = vcross(r, s);
  float u = cqpr / crs, t = cqps / crs;
  int hit = (crs != 0.f) && (0.f <= u) && (u <= 1.f) && (0.f <= t) && (t <= 1.f);
  v2 pi = V(a1.x + t * r.x, a1.y + t * r.y);
  /* four endpoint projections */
  v2 c1[4], c2[4];
  c1[0] = a1; c2[0] = closest_point_line(l2->pos, l2->c, l2->s, l2->half, a1, 1);
  c1[1] = a2; c2[1] = closest_point_line(l2->pos, l2->c, l2->s, l2->half, a2, 1);
  c2[2] = b1; c1[2] = closest_point_line(l1->pos, l1->c, l1->s, l1->half, b1, 1);
  c2[3] = b2; c1[3] = closest_point_line(l1->pos, l1->c, l1->s, l1->half, b2, 1);
  v2 q1 = V(INFINITY, INFINITY), q2 = V(INFINITY, INFINITY);
  float best = INFINITY;
  for (int i = 0; i < 4; ++i) {
    float d = vnorm(vsub(c1[i], c2[i]));
    if (d < best) { q1 = c1[i]; q2 = c2[i]; best = d; }
  }
  if (hit) { q1 = pi; q2 = pi; }
  *p1 = q1; *p2 = q2;
}

/* physics._get_closest_line_box physics.py:328-382: returns (on box, on line) */
static void closest_seg_box(const seg_t be[4], const seg_t* line, v2* p_box, v2* p_line) {
  v2 qb = V(INFINITY, INFINITY), ql = V(INFINITY, INFINITY);
  float best = INFINITY;
  for (int i = 0; i < 4; ++i) {
    v2 pb, pl;
    closest_points_seg_seg(&be[i], line, &pb, &pl);
    float d = vnorm(vsub(pb, pl));
    if (d < best) { qb = pb; ql = pl; best = d; }
  }
  *p_box = qb; *p_line = ql;
}

/* physics._get_closest_box_box physics.py:26-129: returns (on A, on B) */
static void closest_box_box(const seg_t ea[4], const seg_t eb[4], v2* pa, v2* pb) {
  v2 qa = V(INFINITY, INFINITY), qb = V(INFINITY, INFINITY);
  float best = INFINITY;
  for (int i = 0; i < 4; ++i) { /* A's edges against box B: (on B, on A's edge) */
    v2 on_b, on_a;
    closest_seg_box(eb, &ea[i], &on_b, &on_a);
    float d = vnorm(vsub(on_a, on_b));
    if (d < best) { qa = on_a; qb = on_b; best = d; }
  }
  for (int i = 0; i < 4; ++i) { /* B's edges against box A: (on A, on B's edge) */
    v2 on_a, on_b;
    closest_seg_box(ea, &eb[i], &on_a, &on_b);
    float d = vnorm(vsub(on_a, on_b));
    if (d < best) { qa = on_a; qb = on_b; best = d; }
  }
  *pa = qa; *pb = qb;
}

typedef struct { v2 pos, vel; float rot, ang; } ent_t;

/* narrow phase + penalty force of ONE pair: forces/torques on a and b before the
 * movable/rotatable gating of update_env_forces (core.py:2294-2786) */
static void pair_force(const VmasWorldDesc* W, const ent_t* S, int p, v2* fa_o, float* ta_o, v2* fb_o, float* tb_o) {
  const VmasEntityDesc* E = W->entities;
  const VmasPairDesc* P = &W->pairs[p];
  const float k = W->contact_margin;
  const float cfm = W->collision_force;
  int a = P->a, b = P->b;
  v2 fa = V(0.f, 0.f), fb = V(0.f, 0.f);
  float ta = 0.f, tb = 0.f;
  switch (P->type) {
    case VMAS_PAIR_SS: { /* core.py:2294-2339 */
      fa = constraint_force(S[a].pos, S[b].pos, E[a].radius + E[b].radius, cfm, k, 0);
      fb = vneg(fa);
    } break;
    case VMAS_PAIR_LS: { /* a = line, b = sphere core.py:2341-2392 */
      float c = cosf(S[a].rot), s = sinf(S[a].rot);
      v2 cp = closest_point_line(S[a].pos, c, s, E[a].length / 2.f, S[b].pos, 1);
      fb = constraint_force(S[b].pos, cp, E[b].radius + LINE_MIN_DIST, cfm, k, 0);
      fa = vneg(fb);
      ta = vcross(vsub(cp, S[a].pos), fa);
    } break;
    case VMAS_PAIR_LL: { /* core.py:2394-2457 */
      seg_t l1 = {S[a].pos, cosf(S[a].rot), sinf(S[a].rot), E[a].length / 2.f};
      seg_t l2 = {S[b].pos, cosf(S[b].rot), sinf(S[b].rot), E[b].length / 2.f};
      v2 pa, pb;
      closest_points_seg_seg(&l1, &l2, &pa, &pb);
      fa = constraint_force(pa, pb, LINE_MIN_DIST, cfm, k, 0);
      fb = vneg(fa);
      ta = vcross(vsub(pa, S[a].pos), fa);
      tb = vcross(vsub(pb, S[b].pos), fb);
    } break;
    case VMAS_PAIR_BS: { /* a = box, b = sphere core.py:2459-2552 */
      seg_t be[4];
      box_edges(S[a].pos, S[a].rot, E[a].length, E[a].width, be);
      v2 cp = closest_point_box(be, S[b].pos);
      v2 ip = cp;
      float d = 0.f;
      if (!(E[a].flags & VMAS_F_HOLLOW)) ip = inner_point_box(S[b].pos, cp, S[a].pos, &d);
      fb = constraint_force(S[b].pos, ip, E[b].radius + LINE_MIN_DIST + d, cfm, k, 0);
      fa = vneg(fb);
      ta = vcross(vsub(cp, S[a].pos), fa);
    } break;
    case VMAS_PAIR_BL: { /* a = box, b = line core.py:2554-2653 */
      seg_t be[4];
      box_edges(S[a].pos, S[a].rot, E[a].length, E[a].width, be);
      seg_t ln = {S[b].pos, cosf(S[b].rot), sinf(S[b].rot), E[b].length / 2.f};
      v2 pb, pl;
      closest_seg_box(be, &ln, &pb, &pl);
      v2 ip = pb;
      float d = 0.f;
      if (!(E[a].flags & VMAS_F_HOLLOW)) ip = inner_point_box(pl, pb, S[a].pos, &d);
      fa = constraint_force(ip, pl, LINE_MIN_DIST + d, cfm, k, 0);
      fb = vneg(fa);
      ta = vcross(vsub(pb, S[a].pos), fa);
      tb = vcross(vsub(pl, S[b].pos), fb);
    } break;
    case VMAS_PAIR_BB: { /* core.py:2655-2786 */
      seg_t ea[4], eb[4];
      box_edges(S[a].pos, S[a].rot, E[a].length, E[a].width, ea);
      box_edges(S[b].pos, S[b].rot, E[b].length, E[b].width, eb);
      v2 pa, pb;
      closest_box_box(ea, eb, &pa, &pb);
      v2 ia = pa, ib = pb;
      float da = 0.f, db = 0.f;
      if (!(E[a].flags & VMAS_F_HOLLOW)) ia = inner_point_box(pb, pa, S[a].pos, &da);
      if (!(E[b].flags & VMAS_F_HOLLOW)) ib = inner_point_box(pa, pb, S[b].pos, &db);
      fa = constraint_force(ia, ib, da + db + LINE_MIN_DIST, cfm, k, 0);
      fb = vneg(fa);
      ta = vcross(vsub(pa, S[a].pos), fa);
      tb = vcross(vsub(pb, S[b].pos), fb);
    } break;
    default: break;
  }
  *fa_o = fa; *ta_o = ta; *fb_o = fb; *tb_o = tb;
}


static inline void upd(const VmasEntityDesc* E, v2* F, float* T, int a, v2 fa, float ta, int b, v2 fb, float tb) {
  /* World.update_env_forces core.py:2191-2199 */
  if (E[a].flags & VMAS_F_MOVABLE) F[a] = vadd(F[a], fa);
  if (E[a].flags & VMAS_F_ROTATABLE) T[a] = T[a] + ta;
  if (E[b].flags & VMAS_F_MOVABLE) F[b] = vadd(F[b], fb);
  if (E[b].flags & VMAS_F_ROTATABLE) T[b] = T[b] + tb;
}

/* get_friction_force core.py:2055-2073, one component set */
static inline v2 friction2(v2 vel, float coeff, float mass, float sub_dt) {
  float speed = vnorm(vel);
  if (speed == 0.f) return V(0.f, 0.f);
  float ffc = coeff * mass;
  v2 f;
  f.x = -(vel.x / speed) * tmin(ffc, (fabsf(vel.x) / sub_dt) * mass);
  f.y = -(vel.y / speed) * tmin(ffc, (fabsf(vel.y) / sub_dt) * mass);
  return f;
}
static inline float friction1(float vel, float coeff, float inertia, float sub_dt) {
  float speed = fabsf(vel);
  if (speed == 0.f) return 0.f;
  float ffc = coeff * inertia;
  return -(vel / speed) * tmin(ffc, (fabsf(vel) / sub_dt) * inertia);
}

static void step_env(const VmasWorldDesc* W, float* state, float* agent_ft, int64_t ld, int64_t env,
                     const VmasStepArgs* args) {
  const int nE = W->n_entities;
  const VmasEntityDesc* E = W->entities;
  ent_t S[MAX_E];
  v2 F[MAX_E];
  float T[MAX_E];
  for (int e = 0; e < nE; ++e) {
    const float* p = state + (int64_t)e * VMAS_STATE_FIELDS * ld + env;
    S[e].pos = V(p[0], p[ld]);
    S[e].vel = V(p[2 * ld], p[3 * ld]);
    S[e].rot = p[4 * ld];
    S[e].ang = p[5 * ld];
  }
  int first = args ? args->first_substep : 0;
  int count = (args && args->n_substeps > 0) ? args->n_substeps : W->substeps - first;
  const float k = W->contact_margin;
  const float sub_dt = W->sub_dt;

  for (int substep = first; substep < first + count; ++substep) {
    /* ---- prologue core.py:1976-2004 ---- */
    for (int e = 0; e < nE; ++e) {
      F[e] = V(0.f, 0.f);
      T[e] = 0.f;
      uint32_t fl = E[e].flags;
      if (fl & VMAS_F_AGENT) {
        float* af = agent_ft + (int64_t)E[e].agent_index * VMAS_AGENT_FIELDS * ld + env;
        if (fl & VMAS_F_MOVABLE) { /* _apply_action_force core.py:2018-2028 */
          v2 f = V(af[0], af[ld]);
          if (fl & VMAS_F_MAX_F) f = clamp_with_norm(f, E[e].max_f);
          if (fl & VMAS_F_F_RANGE) f = V(clampf(f.x, E[e].f_range), clampf(f.y, E[e].f_range));
          if (fl & (VMAS_F_MAX_F | VMAS_F_F_RANGE)) { af[0] = f.x; af[ld] = f.y; }
          F[e] = vadd(F[e], f);
        }
        if (fl & VMAS_F_ROTATABLE) { /* _apply_action_torque core.py:2030-2041 */
          float t = af[2 * ld];
          if (fl & VMAS_F_MAX_T) { /* clamp_with_norm on a size-1 dim: norm = |t| */
            float n = fabsf(t);
            float nt = (t / n) * E[e].max_t;
            t = n > E[e].max_t ? nt : t;
          }
          if (fl & VMAS_F_T_RANGE) t = clampf(t, E[e].t_range);
          if (fl & (VMAS_F_MAX_T | VMAS_F_T_RANGE)) af[2 * ld] = t;
          T[e] = T[e] + t;
        }
      }
      /* _apply_friction_force core.py:2054-2102 */
      if (fl & VMAS_F_LIN_FRICTION) F[e] = vadd(F[e], friction2(S[e].vel, E[e].lin_friction, E[e].mass, sub_dt));
      if (fl & VMAS_F_ANG_FRICTION) T[e] = T[e] + friction1(S[e].ang, E[e].ang_friction, E[e].inertia, sub_dt);
      /* _apply_gravity core.py:2043-2052 */
      if (fl & VMAS_F_MOVABLE) {
        if (W->has_gravity) F[e] = vadd(F[e], V(E[e].mass * W->gravity[0], E[e].mass * W->gravity[1]));
        if (fl & VMAS_F_GRAVITY) {
          v2 g = V(E[e].gravity[0], E[e].gravity[1]);
          if (args && args->entity_gravity) {
            const float* gp = args->entity_gravity + (int64_t)e * 2 * ld + env;
            g = V(gp[0], gp[ld]);
          }
          F[e] = vadd(F[e], V(E[e].mass * g.x, E[e].mass * g.y));
        }
      }
    }

    /* ---- joints core.py:2201-2292, joints.py:186-216 ---- */
    for (int j = 0; j < W->n_joints; ++j) {
      const VmasJointDesc* J = &W->joints[j];
      int a = J->a, b = J->b;
      v2 pja = vadd(S[a].pos, rotate(V(J->delta_a[0], J->delta_a[1]), cosf(S[a].rot), sinf(S[a].rot)));
      v2 pjb = vadd(S[b].pos, rotate(V(J->delta_b[0], J->delta_b[1]), cosf(S[b].rot), sinf(S[b].rot)));
      v2 fa_att = constraint_force(pja, pjb, J->dist, W->joint_force, k, 1);
      v2 fa_rep = constraint_force(pja, pjb, J->dist, W->joint_force, k, 0);
      v2 fa = vadd(fa_att, fa_rep);
      v2 fb = vadd(vneg(fa_att), vneg(fa_rep));
      float ta = vcross(vsub(pja, S[a].pos), fa);
      float tb = vcross(vsub(pjb, S[b].pos), fb);
      if (!J->rotate) {
        float fr = J->fixed_rotation;
        if (args && args->joint_fixed_rot) fr = args->joint_fixed_rot[(int64_t)j * ld + env];
        float t = constraint_torque(S[a].rot, S[b].rot + fr, W->torque_constraint_force);
        ta = ta + (-t);
        tb = tb + t;
      }
      upd(E, F, T, a, fa, ta, b, fb, tb);
    }

    /* ---- collisions core.py:2294-2786, type-major order ---- */
    {
      /* jitter mode also permutes the ACCUMULATION order (the HIP path adds pair forces
       * with LDS atomics from several lanes, so its order is not the reference's):
       * p -> (p * stride + offset) mod n with gcd(stride, n) = 1. */
      int n = W->n_pairs;
      int64_t stride = 1, offset = 0;
      if (g_jitter && n > 1) {
        offset = (g_jitter >> 3) % n;
        stride = 1 + (g_jitter >> 9) % (n - 1);
        for (;;) {
          int64_t a = stride, b = n;
          while (b) { int64_t t = a % b; a = b; b = t; }
          if (a == 1) break;
          stride = stride % (n - 1) + 1;
        }
      }
      for (int i = 0; i < n; ++i) {
        int p = (int)(((int64_t)i * stride + offset) % n);
        if (args && args->pair_mask && !((args->pair_mask[p >> 5] >> (p & 31)) & 1u)) continue;
        v2 fa, fb;
        float ta, tb;
        pair_force(W, S, p, &fa, &ta, &fb, &tb);
        upd(E, F, T, W->pairs[p].a, fa, ta, W->pairs[p].b, fb, tb);
      }
    }

    /* ---- _integrate_state core.py:2862-2908 ---- */
    for (int e = 0; e < nE; ++e) {
      uint32_t fl = E[e].flags;
      if (fl & VMAS_F_MOVABLE) {
        if (substep == 0) S[e].vel = vscale(S[e].vel, E[e].one_minus_drag);
        v2 acc = V(F[e].x / E[e].mass, F[e].y / E[e].mass);
        S[e].vel = V(S[e].vel.x + acc.x * sub_dt, S[e].vel.y + acc.y * sub_dt);
        if (fl & VMAS_F_MAX_SPEED) S[e].vel = clamp_with_norm(S[e].vel, E[e].max_speed);
        if (fl & VMAS_F_V_RANGE) S[e].vel = V(clampf(S[e].vel.x, E[e].v_range), clampf(S[e].vel.y, E[e].v_range));
        v2 np = V(S[e].pos.x + S[e].vel.x * sub_dt, S[e].pos.y + S[e].vel.y * sub_dt);
        if (W->x_semidim == W->x_semidim) np.x = clampf(np.x, W->x_semidim);
        if (W->y_semidim == W->y_semidim) np.y = clampf(np.y, W->y_semidim);
        S[e].pos = np;
      }
      if (fl & VMAS_F_ROTATABLE) {
        if (substep == 0) S[e].ang = S[e].ang * E[e].one_minus_drag;
        S[e].ang = S[e].ang + (T[e] / E[e].inertia) * sub_dt;
        S[e].rot = S[e].rot + S[e].ang * sub_dt;
      }
    }
  }

  for (int e = 0; e < nE; ++e) {
    float* p = state + (int64_t)e * VMAS_STATE_FIELDS * ld + env;
    uint32_t fl = E[e].flags;
    if (fl & VMAS_F_MOVABLE) { p[0] = S[e].pos.x; p[ld] = S[e].pos.y; p[2 * ld] = S[e].vel.x; p[3 * ld] = S[e].vel.y; }
    if (fl & VMAS_F_ROTATABLE) { p[4 * ld] = S[e].rot; p[5 * ld] = S[e].ang; }
  }
}

int vmas_oracle_step(const VmasWorldDesc* W, int32_t batch, float* state, float* agent_ft, int64_t ld,
                     const VmasStepArgs* args, int32_t n_threads) {
  if (!W || W->abi_version != VMAS_ABI_VERSION || W->n_entities > MAX_E) return -1;
  int64_t env;
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
  for (env = 0; env < batch; ++env) step_env(W, state, agent_ft, ld, env, args);
  return 0;
}

/* batch-global broad phase, World.collides core.py:2797-2801 */
int vmas_oracle_pair_mask(const VmasWorldDesc* W, int32_t batch, const float* state, int64_t ld, uint32_t* mask) {
  if (!W || W->abi_version != VMAS_ABI_VERSION) return -1;
  int words = (W->n_pairs + 31) / 32;
  memset(mask, 0, sizeof(uint32_t) * (size_t)words);
  for (int p = 0; p < W->n_pairs; ++p) {
    int a = W->pairs[p].a, b = W->pairs[p].b;
    float rsum = W->pairs[p].bound_sum;
    const float* pa = state + (int64_t)a * VMAS_STATE_FIELDS * ld;
    const float* pb = state + (int64_t)b * VMAS_STATE_FIELDS * ld;
    for (int64_t env = 0; env < batch; ++env) {
      float d = norm2(pa[env] - pb[env], pa[ld + env] - pb[ld + env]);
      if (d <= rsum) { mask[p >> 5] |= 1u << (p & 31); break; }
    }
  }
  return 0;
}

/* World.cast_rays core.py:1662-1786 for a set of sensors; out[(l*max_rays + r)*ld + env] */
int vmas_oracle_cast_rays(const VmasWorldDesc* W, int32_t batch, const float* state, int64_t ld,
                          const VmasLidarDesc* lidars, int32_t n_lidars, float* out, int32_t n_threads) {
  if (!W || W->abi_version != VMAS_ABI_VERSION) return -1;
  int max_rays = 0;
  for (int l = 0; l < n_lidars; ++l) if (lidars[l].n_rays > max_rays) max_rays = lidars[l].n_rays;
  const VmasEntityDesc* E = W->entities;
  int64_t env;
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
  for (env = 0; env < batch; ++env) {
    for (int l = 0; l < n_lidars; ++l) {
      const VmasLidarDesc* L = &lidars[l];
      const float* sp = state + (int64_t)L->entity * VMAS_STATE_FIELDS * ld + env;
      v2 o = V(sp[0], sp[ld]);
      float arot = sp[4 * ld];
      float R = L->max_range;
      for (int r = 0; r < L->n_rays; ++r) {
        float th = L->angles[r] + arot; /* sensors.py:118 */
        float c = cosf(th), s = sinf(th);
        v2 dir = V(c, s);
        float best = R; /* core.py:1672-1674 */
        for (int ti = 0; ti < L->n_targets; ++ti) {
          int t = L->targets[ti];
          const float* tp = state + (int64_t)t * VMAS_STATE_FIELDS * ld + env;
          v2 tpos = V(tp[0], tp[ld]);
          float trot = tp[4 * ld];
          float dist;
          if (E[t].shape == VMAS_SHAPE_SPHERE) { /* _cast_rays_to_sphere core.py:1414-1490 */
            float half = (float)((double)R / 2.0);
            v2 lp = V(o.x + dir.x * half, o.y + dir.y * half);
            v2 cp = closest_point_line(lp, c, s, 0.f, tpos, 0);
            v2 d = vsub(tpos, cp);
            float dn = vnorm(d);
            int inter = dn < E[t].radius;
            float a = E[t].radius * E[t].radius - dn * dn;
            float m = sqrtf(a > 0.f ? a : 1e-8f);
            v2 u = vsub(tpos, o);
            int front = vdot(u, dir) > 0.f;
            dist = vnorm(vsub(cp, o)) - m;
            if (!(inter && front)) dist = R;
          } else if (E[t].shape == VMAS_SHAPE_BOX) { /* _cast_rays_to_box core.py:1281-1372 */
            float cn = cosf(-trot), sn = sinf(-trot);
            v2 p = rotate(vsub(o, tpos), cn, sn);
            v2 q = rotate(dir, cn, sn);
            float L2 = E[t].length, W2 = E[t].width;
            float tx1 = (-L2 / 2.f - p.x) / q.x, tx2 = (L2 / 2.f - p.x) / q.x;
            float txmin = tmin(tx1, tx2), txmax = tmax(tx1, tx2);
            float ty1 = (-W2 / 2.f - p.y) / q.y, ty2 = (W2 / 2.f - p.y) / q.y;
            float tymin = tmin(ty1, ty2), tymax = tmax(ty1, ty2);
            float t0 = tmax(txmin, tymin), t1 = tmin(txmax, tymax);
            v2 ia = V(t0 * q.x + p.x, t0 * q.y + p.y);
            v2 iw = vadd(rotate(ia, cosf(trot), sinf(trot)), tpos);
            int coll = (t1 >= t0) && (t0 > 0.f);
            dist = vnorm(vsub(o, iw));
            if (!coll) dist = R;
          } else { /* _cast_rays_to_line core.py:1544-1626 */
            v2 rr = V(cosf(trot) * E[t].length, sinf(trot) * E[t].length);
            float rxs = vcross(rr, dir);
            v2 qo = vsub(o, tpos);
            float tt = vcross(qo, V(dir.x / rxs, dir.y / rxs));
            float uu = vcross(qo, V(rr.x / rxs, rr.y / rxs));
            dist = norm2(uu * dir.x, uu * dir.y);
            if (rxs == 0.f || tt > 0.5f || tt < -0.5f || uu < 0.f) dist = R;
          }
          best = tmin(best, dist);
        }
        out[((int64_t)l * max_rays + r) * ld + env] = best;
      }
    }
  }
  return 0;
}

/* debug/unit-test export: out[(f)*ld + env], f = fa.x fa.y ta fb.x fb.y tb of pair p */
int vmas_oracle_pair_forces(const VmasWorldDesc* W, int32_t batch, const float* state, int64_t ld, int32_t p, float* out) {
  if (!W || W->abi_version != VMAS_ABI_VERSION || p < 0 || p >= W->n_pairs) return -1;
  for (int64_t env = 0; env < batch; ++env) {
    ent_t S[MAX_E];
    for (int e = 0; e < W->n_entities; ++e) {
      const float* q = state + (int64_t)e * VMAS_STATE_FIELDS * ld + env;
      S[e].pos = V(q[0], q[ld]); S[e].vel = V(q[2 * ld], q[3 * ld]); S[e].rot = q[4 * ld]; S[e].ang = q[5 * ld];
    }
    v2 fa, fb; float ta, tb;
    pair_force(W, S, p, &fa, &ta, &fb, &tb);
    out[0 * ld + env] = fa.x; out[1 * ld + env] = fa.y; out[2 * ld + env] = ta;
    out[3 * ld + env] = fb.x; out[4 * ld + env] = fb.y; out[5 * ld + env] = tb;
  }
  return 0;
}

/* World.get_distance core.py:1822-1905 / World.is_overlapping core.py:1907-1969 for one env */
static float query_env(const VmasWorldDesc* W, const ent_t* S, int kind, int a, int b) {
  const VmasEntityDesc* E = W->entities;
  int sa = E[a].shape, sb = E[b].shape;
  float dist;
  int overlap = -1;
  if (sa == VMAS_SHAPE_SPHERE && sb == VMAS_SHAPE_SPHERE) {
    dist = (vnorm(vsub(S[a].pos, S[b].pos)) - E[a].radius) - E[b].radius;
  } else if ((sa == VMAS_SHAPE_BOX && sb == VMAS_SHAPE_SPHERE) || (sb == VMAS_SHAPE_BOX && sa == VMAS_SHAPE_SPHERE)) {
    int bx = sa == VMAS_SHAPE_BOX ? a : b, sp = sa == VMAS_SHAPE_BOX ? b : a;
    seg_t be[4];
    box_edges(S[bx].pos, S[bx].rot, E[bx].length, E[bx].width, be);
    v2 cp = closest_point_box(be, S[sp].pos);
    float d_sphere_cp = vnorm(vsub(S[sp].pos, cp));
    float d_sphere_box = vnorm(vsub(S[sp].pos, S[bx].pos));
    float d_box_cp = vnorm(vsub(S[bx].pos, cp));
    overlap = (d_sphere_box < d_box_cp) || (d_sphere_cp < E[sp].radius + LINE_MIN_DIST);
    dist = (d_sphere_cp - LINE_MIN_DIST) - E[sp].radius;
    if (overlap) dist = -1.f;
  } else if ((sa == VMAS_SHAPE_LINE && sb == VMAS_SHAPE_SPHERE) || (sb == VMAS_SHAPE_LINE && sa == VMAS_SHAPE_SPHERE)) {
    int ln = sa == VMAS_SHAPE_LINE ? a : b, sp = sa == VMAS_SHAPE_LINE ? b : a;
    v2 cp = closest_point_line(S[ln].pos, cosf(S[ln].rot), sinf(S[ln].rot), E[ln].length / 2.f, S[sp].pos, 1);
    dist = (vnorm(vsub(S[sp].pos, cp)) - LINE_MIN_DIST) - E[sp].radius;
  } else if (sa == VMAS_SHAPE_LINE && sb == VMAS_SHAPE_LINE) {
    seg_t l1 = {S[a].pos, cosf(S[a].rot), sinf(S[a].rot), E[a].length / 2.f};
    seg_t l2 = {S[b].pos, cosf(S[b].rot), sinf(S[b].rot), E[b].length / 2.f};
    v2 p1, p2;
    closest_points_seg_seg(&l1, &l2, &p1, &p2);
    dist = vnorm(vsub(p1, p2)) - LINE_MIN_DIST;
  } else if ((sa == VMAS_SHAPE_BOX && sb == VMAS_SHAPE_LINE) || (sb == VMAS_SHAPE_BOX && sa == VMAS_SHAPE_LINE)) {
    int bx = sa == VMAS_SHAPE_BOX ? a : b, ln = sa == VMAS_SHAPE_BOX ? b : a;
    seg_t be[4];
    box_edges(S[bx].pos, S[bx].rot, E[bx].length, E[bx].width, be);
    seg_t l = {S[ln].pos, cosf(S[ln].rot), sinf(S[ln].rot), E[ln].length / 2.f};
    v2 pb, pl;
    closest_seg_box(be, &l, &pb, &pl);
    dist = vnorm(vsub(pb, pl)) - LINE_MIN_DIST;
  } else {
    seg_t ea[4], eb[4];
    box_edges(S[a].pos, S[a].rot, E[a].length, E[a].width, ea);
    box_edges(S[b].pos, S[b].rot, E[b].length, E[b].width, eb);
    v2 pa, pb;
    closest_box_box(ea, eb, &pa, &pb);
    dist = vnorm(vsub(pa, pb)) - LINE_MIN_DIST;
  }
  if (kind == VMAS_QUERY_DISTANCE) return dist;
  if (overlap < 0) overlap = dist < 0.f;
  return overlap ? 1.f : 0.f;
}

int vmas_oracle_queries(const VmasWorldDesc* W, int32_t batch, const float* state, int64_t ld, const VmasQuery* q,
                        int32_t n, float* out) {
  if (!W || W->abi_version != VMAS_ABI_VERSION || W->n_entities > MAX_E) return -1;
  for (int64_t env = 0; env < batch; ++env) {
    ent_t S[MAX_E];
    for (int e = 0; e < W->n_entities; ++e) {
      const float* p = state + (int64_t)e * VMAS_STATE_FIELDS * ld + env;
      S[e].pos = V(p[0], p[ld]); S[e].vel = V(p[2 * ld], p[3 * ld]); S[e].rot = p[4 * ld]; S[e].ang = p[5 * ld];
    }
    for (int i = 0; i < n; ++i) out[(int64_t)i * ld + env] = query_env(W, S, q[i].kind, q[i].a, q[i].b);
  }
  return 0;
}

int vmas_oracle_abi_version(void) { return VMAS_ABI_VERSION; }
