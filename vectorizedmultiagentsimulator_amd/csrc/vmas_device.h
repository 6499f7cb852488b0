// vmas_device.h - device-side math of the VMAS physics step for gfx950 (CDNA4).
//
// Every function restates one reference routine (paths relative to
// /root/reference/vmas/simulator/) with the reference's operation ORDER, in IEEE fp32
// with FMA contraction off (build flag -ffp-contract=off).  The one fused operation is
// the 2-vector norm: torch's CPU `linalg.vector_norm` over a size-2 dim is bitwise
// sqrt(fma(y, y, x*x)), so that is what norm2() computes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vmas {

#define VD __device__ __forceinline__

constexpr float kLineMinDist = (float)(4.0 / 6e2);                         // utils.py:28
constexpr float kHalfPi = (float)(3.14159265358979323846 / 2.0);           // torch.pi / 2 -> fp32
constexpr float kInf = __builtin_huge_valf();

typedef float v2 __attribute__((ext_vector_type(2)));  // (+, -, unary - are element-wise: v_pk_add_f32 where they pair up)
VD v2 V(float x, float y) { v2 r; r.x = x; r.y = y; return r; }
// IEEE sqrt and division as hipcc expands them for gfx950, minus the exponent-range scaffolding:
//  * sqrt: the compiler wraps v_sqrt_f32 in a denormal-input rescue (cmp, cndmask, ldexp, ..., cndmask, ldexp: 6 VALU per
//    root).  Arguments here are squared lengths; one below 2^-126 means a distance below 1e-19, which every caller treats
//    as zero anyway (core.py:2836-2838).  The bare instruction gives the same bits for every normal input.
//  * division: v_div_scale x2, v_rcp, Newton step, two quotient corrections, v_div_fmas, v_div_fixup (12 VALU).  The
//    scaling only acts when an exponent sits at the edge of the format; without it the SAME instruction sequence needs 9,
//    and quotients that share a denominator (x/n, y/n) share its refined reciprocal: 6 each.  v_div_fixup keeps the IEEE
//    results for 0, inf and NaN operands.  Same bits as a/b whenever div_scale would not have scaled.
VD float sqrt_n(float x) { return __builtin_amdgcn_sqrtf(x); }
struct rcp_t { float b, y; };
VD rcp_t rcp_of(float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.f);
  y = __builtin_fmaf(e, y, y);
  rcp_t r; r.b = b; r.y = y;
  return r;
}
VD float operator/(float a, rcp_t d) {
  float q = a * d.y;
  float r = __builtin_fmaf(-d.b, q, a);
  q = __builtin_fmaf(r, d.y, q);
  r = __builtin_fmaf(-d.b, q, a);
  q = __builtin_fmaf(r, d.y, q);
  return __builtin_amdgcn_div_fixupf(q, d.b, a);
}
VD float norm2(float x, float y) { return sqrt_n(__fmaf_rn(y, y, x * x)); }
VD float vnorm(v2 a) { return norm2(a.x, a.y); }
VD float vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }               // (a*b).sum(-1)
VD float vcross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }             // utils.py:193-197
VD float sign_t(float x) { return x != x ? x : (float)((x > 0.f) - (x < 0.f)); }  // torch.sign
VD float min_t(float a, float b) { const float r = a < b ? a : b; return a != a ? a : r; }  // torch.minimum: NaN from either side
VD float max_t(float a, float b) { const float r = a > b ? a : b; return a != a ? a : r; }
VD float clamp_t(float x, float r) { const float lo = 0.f - r; const float t = x < lo ? lo : x; return t > r ? r : t; }  // NaN stays
VD v2 rotate(v2 v, float c, float s) { return V(v.x * c - v.y * s, v.x * s + v.y * c); }  // utils.py:175-191

// TorchUtils.clamp_with_norm utils.py:167-173
VD v2 clamp_with_norm(v2 t, float max_norm) {
  float n = vnorm(t);
#ifndef VMAS_X_CLAMP
#define VMAS_X_CLAMP 1
#endif
  if (VMAS_X_CLAMP && !__any(n > max_norm)) return t;  // no lane of the wave is over the limit (a NaN norm keeps t in the reference too)
  const rcp_t rn = rcp_of(n);
  v2 nt = V((t.x / rn) * max_norm, (t.y / rn) * max_norm);
  return n > max_norm ? nt : t;
}

// log1p(y) for y in [0, 1] (y = exp(-|x|)): u = fl(1 + y), e = y - (u - 1) is the exact rounding
// error of that sum (Sterbenz), log1p(y) = log(u) + e/u to first order.  ~20 VALU instead of
// ocml log1pf's ~120 - the softplus dominated the contact force - at the same <= 1-2 ulp
// accuracy class as the libms it is compared with (tests/test_hip_math.py measures it).
VD float log1p_unit(float y) {
  const float u = 1.f + y;
  const float e = y - (u - 1.f);
  // logf(u) as ocml evaluates it for a normal, finite u (here u is in [1, 2]): log2 times ln2 as a two-term product
  const float r = __builtin_amdgcn_logf(u);
  const float h = 0x1.62e42ep-1f * r;
  float l = __builtin_fmaf(r, 0x1.62e42ep-1f, -h);
  l = __builtin_fmaf(r, 0x1.efa39ep-25f, l);
  return (h + l) + e * __builtin_amdgcn_rcpf(u);
}
// torch.logaddexp(0, x): max(0, x) + log1p(exp(-|x|))
VD float softplus0(float x) { return max_t(0.f, x) + log1p_unit(expf(-fabsf(0.f - x))); }

// World._get_constraint_forces core.py:2805-2839 -> force on a (force on b is -f).
// `c` = fp32(sign * force_multiplier), precomputed on the host like Python does.
template <bool ATTRACTIVE>
VD v2 constraint_force(v2 pa, v2 pb, float dist_min, float c, float k) {
  v2 d = pa - pb;
  float dist = vnorm(d);
  float sign = ATTRACTIVE ? -1.f : 1.f;
  float pen = softplus0((dist_min - dist) * sign / rcp_of(k)) * k;
  const rcp_t den = rcp_of(dist > 0.f ? dist : 1e-8f);
  v2 f = V(c * d.x / den * pen, c * d.y / den * pen);
  bool zero = dist < 1e-6f;
  zero = zero || (ATTRACTIVE ? (dist < dist_min) : (dist > dist_min));
  return zero ? V(0.f, 0.f) : f;
}

// Repulsive contact force with a wave-level early out: when NO lane of the wave is within
// dist_min the reference zeroes every lane's force (core.py:2836), so the softplus and the
// divisions are skipped for the whole wave.  Bitwise identical results.
VD v2 contact_force(v2 pa, v2 pb, float dist_min, float c, float k) {
  const v2 d = pa - pb;
  const float dist = vnorm(d);
  if (!__any(!(dist > dist_min))) return V(0.f, 0.f);
  const float pen = softplus0((dist_min - dist) / rcp_of(k)) * k;  // sign = +1
  const rcp_t den = rcp_of(dist > 0.f ? dist : 1e-8f);
  const v2 f = V(c * d.x / den * pen, c * d.y / den * pen);
  const bool zero = (dist < 1e-6f) || (dist > dist_min);
  return zero ? V(0.f, 0.f) : f;
}

// World._get_constraint_torques core.py:2841-2858 -> torque on b (torque on a is -t)
VD float constraint_torque(float rot_a, float rot_b, float force_multiplier) {
  float delta = rot_a - rot_b;
  float ad = fabsf(delta);
  float pen = expf(ad) - 1.f;
  float t = force_multiplier * sign_t(delta) * pen;
  return ad < 1e-9f ? 0.f : t;
}

// physics._get_closest_point_line physics.py:400-429; (c, s) = cos/sin(line_rot).
// The reference's  sign(dot) * min(|dot|, L/2)  is dot clamped to [-L/2, L/2] - the same bits for every
// finite dot (and a NaN still propagates: both comparisons are false) in 4 instructions instead of ~14; it is
// the most-called primitive of the narrow phase (4 per box-sphere, 16 per box-line pair).
template <bool LIMIT>
VD v2 closest_point_line(v2 pos, float c, float s, float half_len, v2 p) {
  v2 d = pos - p;
  float dot = d.x * c + d.y * s;
  float t = dot;
  if (LIMIT) {
    const float lo = 0.f - half_len;
    t = dot < lo ? lo : dot;
    t = t > half_len ? half_len : t;
  }
  return V(pos.x - t * c, pos.y - t * s);
}

struct seg_t { v2 pos; float c, s, half; };  // centre, cos/sin(rot), half length

// physics._get_all_lines_box physics.py:298-325; (c,s) = cos/sin(rot), (c2,s2) = cos/sin(rot + pi/2)
VD void box_edges(v2 pos, float c, float s, float c2, float s2, float length, float width, seg_t e[4]) {
  float hl = length / 2.f, hw = width / 2.f;
  e[0].pos = V(pos.x + c * hl, pos.y + s * hl);
  e[1].pos = V(pos.x - c * hl, pos.y - s * hl);
  e[2].pos = V(pos.x + c2 * hw, pos.y + s2 * hw);
  e[3].pos = V(pos.x - c2 * hw, pos.y - s2 * hw);
  e[0].c = e[1].c = c2; e[0].s = e[1].s = s2; e[0].half = e[1].half = hw;
  e[2].c = e[3].c = c;  e[2].s = e[3].s = s;  e[2].half = e[3].half = hl;
}

// physics._get_closest_point_box physics.py:263-295: nearest perimeter point over the four edges, strict < in edge
// order.  As in closest_seg_box below, an edge is only solved if it can win: |coordinate - edge| in the box frame is a
// lower bound of the point's distance to that edge, its distance to the perimeter an upper bound of the winner's.
VD v2 closest_point_box(const seg_t e[4], v2 p) {
  const float c = e[2].c, s = e[2].s, hl = e[2].half, hw = e[0].half;
  const float dx = p.x - (e[0].pos.x + e[1].pos.x) * 0.5f, dy = p.y - (e[0].pos.y + e[1].pos.y) * 0.5f;
  const float px = dx * c + dy * s, py = dy * c - dx * s;
  const float ax = fabsf(px) - hl, ay = fabsf(py) - hw;
  const float ox = fmaxf(ax, 0.f), oy = fmaxf(ay, 0.f);
  const float inside = (ax <= 0.f && ay <= 0.f) ? fminf(0.f - ax, 0.f - ay) : 0.f;
  const float ub = sqrt_n(ox * ox + oy * oy) + inside;
  const float thr = ub + 1e-4f * ub + 1e-5f * (1.f + hl + hw + fabsf(px) + fabsf(py));
  const float lb[4] = {fabsf(px - hl), fabsf(px + hl), fabsf(py - hw), fabsf(py + hw)};
  v2 best = V(kInf, kInf);
  float dist = kInf;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (__any(!(lb[i] > thr))) {
      v2 q = closest_point_line<true>(e[i].pos, e[i].c, e[i].s, e[i].half, p);
      float d = vnorm(p - q);
      bool cl = d < dist;
      best = cl ? q : best;
      dist = cl ? d : dist;
    }
  }
  return best;
}

// physics._get_inner_point_box physics.py:13-23
VD v2 inner_point_box(v2 outside, v2 surface, v2 box_pos, float& depth) {
  v2 v = surface - outside;
  v2 u = box_pos - surface;
  float n = vnorm(v);
  const rcp_t rn = rcp_of(n);
  float xm = vdot(v, u) / rn;
  v2 x = V((v.x / rn) * xm, (v.y / rn) * xm);
  bool z = n == 0.f;
  x = z ? surface : x;
  xm = z ? 0.f : xm;
  depth = fabsf(xm);
  return surface + x;
}

// physics._get_closest_points_line_line physics.py:144-219 (+132-141, 222-260)
VD void closest_points_seg_seg(const seg_t& l1, const seg_t& l2, v2& p1, v2& p2) {
  v2 xy1 = V(l1.half * l1.c, l1.half * l1.s);
  v2 xy2 = V(l2.half * l2.c, l2.half * l2.s);
  v2 a1 = l1.pos + xy1, a2 = l1.pos - xy1;
  v2 b1 = l2.pos + xy2, b2 = l2.pos - xy2;
  v2 r = a2 - a1, s = b2 - b1, qp = b1 - a1;
  float cqpr = vcross(qp, r), cqps = vcross(qp, s), crs = vcross(r, s);
  const rcp_t rc = rcp_of(crs);
  float u = cqpr / rc, t = cqps / rc;
  bool hit = (crs != 0.f) && (0.f <= u) && (u <= 1.f) && (0.f <= t) && (t <= 1.f);
  v2 pi = V(a1.x + t * r.x, a1.y + t * r.y);
  v2 q1 = V(kInf, kInf), q2 = V(kInf, kInf);
  float best = kInf;
  {
    v2 c2 = closest_point_line<true>(l2.pos, l2.c, l2.s, l2.half, a1);
    float d = vnorm(a1 - c2);
    bool cl = d < best; q1 = cl ? a1 : q1; q2 = cl ? c2 : q2; best = cl ? d : best;
  }
  {
    v2 c2 = closest_point_line<true>(l2.pos, l2.c, l2.s, l2.half, a2);
    float d = vnorm(a2 - c2);
    bool cl = d < best; q1 = cl ? a2 : q1; q2 = cl ? c2 : q2; best = cl ? d : best;
  }
  {
    v2 c1 = closest_point_line<true>(l1.pos, l1.c, l1.s, l1.half, b1);
    float d = vnorm(c1 - b1);
    bool cl = d < best; q1 = cl ? c1 : q1; q2 = cl ? b1 : q2; best = cl ? d : best;
  }
  {
    v2 c1 = closest_point_line<true>(l1.pos, l1.c, l1.s, l1.half, b2);
    float d = vnorm(c1 - b2);
    bool cl = d < best; q1 = cl ? c1 : q1; q2 = cl ? b2 : q2; best = cl ? d : best;
  }
  p1 = hit ? pi : q1;
  p2 = hit ? pi : q2;
}

// physics._get_closest_line_box physics.py:328-382 -> (on box, on line): the closest pair over the four edges, strict <
// in edge order.  An edge is only SOLVED if it can win.  In the box frame the segment's centre (px, py) and half
// extents (ex, ey) give, per edge, a lower bound of its distance to the segment (distance of the segment's interval to
// the edge's supporting line), and the distance of the segment's centre to the perimeter is an upper bound of the
// winning distance; an edge whose bound exceeds it (with slack for the rounding of both sides) cannot be the minimum for
// any lane of the wave and is skipped - exactly: the edges that are solved are compared in the reference's order.
// (balance: the line over the 10 x 1 floor solves the top edge only: 3 of 4 segment-segment solves gone.)  A NaN
// anywhere makes every test fail towards "solve".
VD void closest_seg_box(const seg_t be[4], const seg_t& line, v2& p_box, v2& p_line) {
  const float c = be[2].c, s = be[2].s, hl = be[2].half, hw = be[0].half;  // (physics.py:298-325: edges 2, 3 run along the box)
  const float dx = line.pos.x - (be[0].pos.x + be[1].pos.x) * 0.5f, dy = line.pos.y - (be[0].pos.y + be[1].pos.y) * 0.5f;
  const float px = dx * c + dy * s, py = dy * c - dx * s;
  const float ex = fabsf(line.half * (line.c * c + line.s * s)), ey = fabsf(line.half * (line.s * c - line.c * s));
  const float ax = fabsf(px) - hl, ay = fabsf(py) - hw;
  const float ox = fmaxf(ax, 0.f), oy = fmaxf(ay, 0.f);
  const float inside = (ax <= 0.f && ay <= 0.f) ? fminf(0.f - ax, 0.f - ay) : 0.f;
  const float ub = sqrt_n(ox * ox + oy * oy) + inside;
  const float thr = ub + 1e-4f * ub + 1e-5f * (1.f + hl + hw + fabsf(px) + fabsf(py));
  const float lb[4] = {fmaxf(px - ex - hl, hl - px - ex), fmaxf(0.f - px - ex - hl, hl + px - ex),
                       fmaxf(py - ey - hw, hw - py - ey), fmaxf(0.f - py - ey - hw, hw + py - ey)};
  v2 qb = V(kInf, kInf), ql = V(kInf, kInf);
  float best = kInf;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (__any(!(lb[i] > thr))) {
      v2 pb, pl;
      closest_points_seg_seg(be[i], line, pb, pl);
      float d = vnorm(pb - pl);
      bool cl = d < best;
      qb = cl ? pb : qb; ql = cl ? pl : ql; best = cl ? d : best;
    }
  }
  p_box = qb; p_line = ql;
}

// physics._get_closest_box_box physics.py:26-129 -> (on A, on B)
VD void closest_box_box(const seg_t ea[4], const seg_t eb[4], v2& pa, v2& pb) {
  v2 qa = V(kInf, kInf), qb = V(kInf, kInf);
  float best = kInf;
#pragma unroll 1  // (32 segment-segment solves: unrolled they cost 90 KB of code and every VGPR the wave has)
  for (int i = 0; i < 4; ++i) {  // A's edges against box B -> (on B, on A's edge)
    v2 on_b, on_a;
    closest_seg_box(eb, ea[i], on_b, on_a);
    float d = vnorm(on_a - on_b);
    bool cl = d < best;
    qa = cl ? on_a : qa; qb = cl ? on_b : qb; best = cl ? d : best;
  }
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {  // B's edges against box A -> (on A, on B's edge)
    v2 on_a, on_b;
    closest_seg_box(ea, eb[i], on_a, on_b);
    float d = vnorm(on_a - on_b);
    bool cl = d < best;
    qa = cl ? on_a : qa; qb = cl ? on_b : qb; best = cl ? d : best;
  }
  pa = qa; pb = qb;
}

// squared distance from point p to the solid oriented box (centre c, axes (cs,sn)), 0 inside
VD float obb_dist2(v2 p, v2 c, float cs, float sn, float half_l, float half_w) {
  const float dx = p.x - c.x, dy = p.y - c.y;
  const float lx = fabsf(dx * cs + dy * sn) - half_l;
  const float ly = fabsf(dy * cs - dx * sn) - half_w;
  const float ex = fmaxf(lx, 0.f), ey = fmaxf(ly, 0.f);
  return ex * ex + ey * ey;
}

// separating-axis lower bound of the distance between a segment (centre p, direction (lc,ls),
// half length h) and the oriented box: the larger of the two gaps along the box axes
VD float seg_obb_gap(v2 p, float lc, float ls, float h, v2 c, float cs, float sn, float half_l,
                                             float half_w) {
  const float dx = p.x - c.x, dy = p.y - c.y;
  const float px = dx * cs + dy * sn, py = dy * cs - dx * sn;            // segment centre in the box frame
  const float ex = fabsf(h * (lc * cs + ls * sn)), ey = fabsf(h * (ls * cs - lc * sn));  // its half extents
  const float gx = fabsf(px) - ex - half_l, gy = fabsf(py) - ey - half_w;
  return fmaxf(gx, gy);
}

// get_friction_force core.py:2055-2073
VD v2 friction2(v2 vel, float coeff, float mass, float sub_dt) {
  float speed = vnorm(vel);
  float ffc = coeff * mass;
  v2 f;
  f.x = -(vel.x / speed) * min_t(ffc, (fabsf(vel.x) / sub_dt) * mass);
  f.y = -(vel.y / speed) * min_t(ffc, (fabsf(vel.y) / sub_dt) * mass);
  return speed == 0.f ? V(0.f, 0.f) : f;
}
VD float friction1(float vel, float coeff, float inertia, float sub_dt) {
  float speed = fabsf(vel);
  float ffc = coeff * inertia;
  float f = -(vel / speed) * min_t(ffc, (fabsf(vel) / sub_dt) * inertia);
  return speed == 0.f ? 0.f : f;
}

// ------------------------------------------------------------------------------------
// scenario-side queries: World.get_distance core.py:1822-1905, World.is_overlapping core.py:1907-1969
// for one ordered pair (a, b) = (box, sphere) / (line, sphere) / (box, line) / same-shape, one
// environment per lane (the shape dispatch is uniform over the wave)
// ------------------------------------------------------------------------------------
constexpr int kSphere = 0, kBox = 1, kLine = 2;  // == VMAS_SHAPE_* of include/vmas_hip.h

struct DevQuery {
  int32_t kind, a, b;            // a/b already ordered as above
  int32_t sa, sb;                // shape codes
  float la, wa, ra, lb, wb, rb;  // length / width / radius of a and b
};

// returns the distance; `overlap` = is_overlapping(a, b)
VD float pair_distance(const DevQuery& Q, const float* __restrict__ state, long ld, long env, int& overlap) {
  const float* A = state + (long)Q.a * 6 * ld + env;
  const float* B = state + (long)Q.b * 6 * ld + env;
  const v2 pa = V(A[0], A[ld]), pb = V(B[0], B[ld]);
  float dist;
  int ov = -1;
  if (Q.sa == kSphere) {  // sphere - sphere
    dist = (vnorm(pa - pb) - Q.ra) - Q.rb;
  } else if (Q.sa == kBox && Q.sb == kSphere) {
    const float rot = A[4 * ld], rot2 = rot + kHalfPi;
    seg_t be[4];
    box_edges(pa, cosf(rot), sinf(rot), cosf(rot2), sinf(rot2), Q.la, Q.wa, be);
    const v2 cp = closest_point_box(be, pb);
    const float d_sphere_cp = vnorm(pb - cp), d_sphere_box = vnorm(pb - pa), d_box_cp = vnorm(pa - cp);
    ov = (d_sphere_box < d_box_cp) || (d_sphere_cp < Q.rb + kLineMinDist);
    dist = ov ? -1.f : (d_sphere_cp - kLineMinDist) - Q.rb;
  } else if (Q.sa == kLine && Q.sb == kSphere) {
    const float rot = A[4 * ld];
    const v2 cp = closest_point_line<true>(pa, cosf(rot), sinf(rot), Q.la / 2.f, pb);
    dist = (vnorm(pb - cp) - kLineMinDist) - Q.rb;
  } else if (Q.sa == kLine) {  // line - line
    const float r1 = A[4 * ld], r2 = B[4 * ld];
    seg_t l1 = {pa, cosf(r1), sinf(r1), Q.la / 2.f};
    seg_t l2 = {pb, cosf(r2), sinf(r2), Q.lb / 2.f};
    v2 p1, p2;
    closest_points_seg_seg(l1, l2, p1, p2);
    dist = vnorm(p1 - p2) - kLineMinDist;
  } else if (Q.sb == kLine) {  // box - line
    const float rot = A[4 * ld], rot2 = rot + kHalfPi, rl = B[4 * ld];
    seg_t be[4];
    box_edges(pa, cosf(rot), sinf(rot), cosf(rot2), sinf(rot2), Q.la, Q.wa, be);
    seg_t l = {pb, cosf(rl), sinf(rl), Q.lb / 2.f};
    v2 qb, ql;
    closest_seg_box(be, l, qb, ql);
    dist = vnorm(qb - ql) - kLineMinDist;
  } else {  // box - box
    const float r1 = A[4 * ld], r1b = r1 + kHalfPi, r2 = B[4 * ld], r2b = r2 + kHalfPi;
    seg_t ea[4], eb[4];
    box_edges(pa, cosf(r1), sinf(r1), cosf(r1b), sinf(r1b), Q.la, Q.wa, ea);
    box_edges(pb, cosf(r2), sinf(r2), cosf(r2b), sinf(r2b), Q.lb, Q.wb, eb);
    v2 qa, qb;
    closest_box_box(ea, eb, qa, qb);
    dist = vnorm(qa - qb) - kLineMinDist;
  }
  overlap = ov < 0 ? (dist < 0.f) : ov;
  return dist;
}

}  // namespace vmas
