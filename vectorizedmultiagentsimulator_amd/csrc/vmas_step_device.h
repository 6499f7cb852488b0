// vmas_step_device.h - device code shared by the schedule interpreter (vmas_hip.hip: step_kernel), its world-specialised
// form (vmas_spec_kernel.h) and the specialisations compiled at run time for a world (vmas_specialize.*): the item /
// segment / owner records of a schedule and the evaluators of one item.  Moved out of vmas_hip.hip unchanged, so that a
// run-time translation unit can include exactly the code the library itself was built from.
#pragma once
#include "vmas_step_types.h"

// ------------------------------------------------------------------------------------
// device-side constant block (all wave-uniform => scalar loads)
// ------------------------------------------------------------------------------------
constexpr int ITEMS_LDS_BUDGET = 48 * 1024;  // stage the item descriptors in LDS when they fit
constexpr int TASK_JOINT = 6;  // item type next to VMAS_PAIR_*
#ifndef VMAS_X_TIGHT_LS
#define VMAS_X_TIGHT_LS 1
#endif
#ifndef VMAS_X_FIRED
#define VMAS_X_FIRED 1
#endif
constexpr int FIRED_RECS = VMAS_X_FIRED ? 16 : 0;       // shared sphere-sphere records whose fired bits are published in LDS (two words per
                                     // substep parity: 4 bits per record, 64 pairs)
constexpr int TASK_SSQ = 7;    // up to four sphere-sphere partners of one entity in one record
constexpr int TASK_LSQ = 8;    // up to four lines against one (owning) sphere in one record
constexpr int TASK_SSP = 9;    // up to four SHARED sphere-sphere pairs (both spheres dynamic) in one record

enum : uint32_t { IT_A_HOLLOW = 1u << 0, IT_B_HOLLOW = 1u << 1, IT_LOCK = 1u << 2 /* joint rotate == False */ };

// One evaluation of a joint/pair FOR ONE SIDE of it.  Static data pre-resolved on the host
// with the same fp32 operations the reference performs at run time.  All LDS positions are
// float offsets into the tile, so row accesses become ds_read with an immediate offset.
struct DevItem {
  int32_t type;   // VMAS_PAIR_* or TASK_JOINT
  int32_t side;   // low 2 bits  0: the force on a, 1: on b, 2: SHARED - evaluated once, results to the LDS rows at
                  // tile offset (side >> 2): [fx fy] of a, then the torque on a, then the torque on b
  uint32_t flags; // IT_*
  int32_t index;  // pair index (mask bit) or joint index (per-env fixed-rotation row)
  int32_t oa, ob;    // tile offsets of the first state row of a / b (role order of the reference)
  int32_t tra, trb;  // tile offsets of their trig rows (unused for spheres)
  float thr2;     // (R_a + R_b + LINE_MIN_DIST + slack)^2: bounding-circle skip
  float reach;    // box items: the other shape's reach + LINE_MIN_DIST + slack (oriented-box skip)
  float p0, p1, p2, p3, q0, q1;  // type-specific dims, see build_items()
};

struct DevSegment {
  int32_t entity;    // -1: a run of shared pairs/joints (no prologue, no partial rows; every item has its own rows)
  int32_t oe;        // tile offset of the entity's first state row
  int32_t item_begin, item_end;
  int32_t first;     // 1: this segment starts from the entity's prologue force
  int32_t part_off;  // tile offset of its 3 partial-sum rows (fx, fy, torque)
  uint32_t eflags;   // the entity's VMAS_F_* flags (saves the dependent fetch of its descriptor)
  int32_t pad;       // (32 bytes: the record is two ds_read_b128)
};

struct DevOwned {  // phase C work unit: everything the integration of one entity needs, ONE LDS round trip (6 x 16 bytes)
  int32_t entity;
  int32_t oe;
  int32_t part_off, n_parts;  // partial rows of the entity's segments, in order
  // the entity's side of every shared pair/joint it is in, in the reference's accumulation order; one word each:
  //   bits 0..15 row of [fx fy] | bit 16 side (1: the entity is b, the force flips its sign) | bits 17..18 row delta
  //   of its torque (0: none).  The first eight are inlined below, the rest follow at blob[b_refs + ref_begin + 8].
  int32_t ref_begin, n_refs;
  uint32_t flags;
  int32_t shape;
  int32_t tr_off;
  float mass, inertia, one_minus_drag;
  float max_speed, v_range;
  uint32_t pad[2];
  uint32_t refs8[8];
};
static_assert(sizeof(DevOwned) == 96, "DevOwned is read as six uint4");



// (DevMaskPair, DevLidar, DevTarget: vmas_env_device.h - the navigation epilogue uses them too)





// cos/sin of the rotation(s) the narrow phase needs (physics.py:300-302, 413)
__device__ __forceinline__ void write_trig(float* tr, float rot, int shape) {
  float sn, cs;
  sincosf(rot, &sn, &cs);  // one range reduction for both (ocml), same values as cosf/sinf
  tr[0 * ROWF] = cs;
  tr[1 * ROWF] = sn;
  if (shape == VMAS_SHAPE_BOX) {
    const float rot2 = rot + kHalfPi;
    sincosf(rot2, &sn, &cs);
    tr[2 * ROWF] = cs;
    tr[3 * ROWF] = sn;
  }
}

// (obb_dist2, seg_obb_gap: vmas_device.h - the balance epilogue uses the separating-axis gap too)

// Register views of descriptors read from the LDS blob (uniform address => broadcast read):
// control words go to SGPRs, offsets and float parameters stay in (uniform) VGPRs.
struct ItemV {
  int32_t type, side; uint32_t flags; int32_t index;  // scalar
  int32_t oa, ob, tra, trb;
  float thr2, reach, p0, p1, p2, p3, q0, q1;
};
// The 16 words of one item record, as fetched (uniform address: broadcast read).  The gather loops fetch record i + 1
// while record i is evaluated: the record's LDS round trip is off the wave's dependent chain.
struct ItemW { uint4 w0, w1, w2, w3; };
__device__ __forceinline__ ItemW fetch_words(const uint32_t* p) {
  ItemW r;
  r.w0 = ((const uint4*)p)[0]; r.w1 = ((const uint4*)p)[1]; r.w2 = ((const uint4*)p)[2]; r.w3 = ((const uint4*)p)[3];
  return r;
}
__device__ __forceinline__ ItemV load_item(const ItemW& I) {
  const uint4 w0 = I.w0, w1 = I.w1, w2 = I.w2, w3 = I.w3;
  ItemV K;
  K.type = sgpr((int)w0.x); K.side = sgpr((int)w0.y); K.flags = (uint32_t)sgpr((int)w0.z); K.index = sgpr((int)w0.w);
  K.oa = (int)w1.x; K.ob = (int)w1.y; K.tra = (int)w1.z; K.trb = (int)w1.w;
  K.thr2 = __uint_as_float(w2.x); K.reach = __uint_as_float(w2.y); K.p0 = __uint_as_float(w2.z); K.p1 = __uint_as_float(w2.w);
  K.p2 = __uint_as_float(w3.x); K.p3 = __uint_as_float(w3.y); K.q0 = __uint_as_float(w3.z); K.q1 = __uint_as_float(w3.w);
  return K;
}
static_assert(sizeof(DevItem) == 64, "descriptor layout");

// Sphere-sphere partners packed four to a record (same 16 words as a DevItem):
//   w0: type, n, -, -   w1: own offset, partner offsets 0..2   w2: partner 3, r_sum 0..2
//   w3: r_sum 3, pair index 0|1<<16, pair index 2|3<<16, -
// One descriptor fetch and eight position reads in flight instead of four dependent round
// trips; the forces are added to F one by one, in the reference's order.  Both sides of a
// sphere pair see force(own, other): cf(a,b) == -cf(b,a) bit for bit, so no sign flip is needed.
__device__ __forceinline__ void eval_ssq(const ItemW& I, const DevWorld& W, const DevStepArgs& args,
                                         const float* tile, bool movable, v2& F, bool live) {
  const uint4 w0 = I.w0, w1 = I.w1, w2 = I.w2, w3 = I.w3;
  const int n = sgpr((int)w0.y);
  const float* E = tile + (int)w1.x;
  const v2 pe = V(E[0], E[ROWF]);
  const int ob[4] = {(int)w1.y, (int)w1.z, (int)w1.w, (int)w2.x};
  const float rs[4] = {__uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w), __uint_as_float(w3.x)};
  const int idx[4] = {(int)(w3.y & 0xffffu), (int)(w3.y >> 16), (int)(w3.z & 0xffffu), (int)(w3.z >> 16)};
  v2 po[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* O = tile + ob[k];
    po[k] = V(O[0], O[ROWF]);  // unused slots repeat partner 0 on the host: always a valid row
  }
  uint32_t needbits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = pe.x - po[k].x, dy = pe.y - po[k].y;
    const float m = rs[k] + 1e-4f;  // the force is exactly 0 for dist > r_a + r_b (core.py:2836)
    bool need = !far_apart(dx * dx + dy * dy, m * m);
    bool on = k < n;
    on = on && pair_on(args, sgpr(idx[k]));
    needbits |= (on && __any(need)) ? (1u << k) : 0u;
  }
  if (!needbits || (ABLATE(args) & 32)) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (needbits & (1u << k)) {
      const v2 f = contact_force(pe, po[k], rs[k], W.c_coll, W.k);
      // (lazy form: a sphere pair's only band events are non-finite poses - NaN force, circles "apart": rs is the pair's
      //  bounding-circle sum, host-checked)
      if (lazy_noting(args)) {
        const bool ov = live && vnorm(pe - po[k]) <= rs[k];
        lazy_acc_overlap(args, sgpr(idx[k]), ov);  // (overlapping spheres are within reach: this is the only place to look)
        lazy_note_band(args, sgpr(idx[k]), live && !ov && (f.x != 0.f || f.y != 0.f));
      }
      if (movable) F = F + f;
    }
  }
}

// Line-sphere pairs seen from the SPHERE (the owner; lines are mostly static walls), up to THREE lines to a record (four
// until round 6: the fourth line's words now carry the pairs' bounding-circle thresholds, circles_overlap):
//   w0: type, n, own offset, r + LINE_MIN_DIST   w1: line offsets 0|1, 2|-, trig offsets 0|1, 2|- (16 bit each)
//   w2: half lengths 0..2, threshold 0   w3: pair index 0|1<<16, 2, thresholds 1..2
// Same arithmetic as the unpacked item (own(-cf(sphere, cp)) with the b-side sign flip == cf(sphere, cp) bit for
// bit), one descriptor fetch and twelve operand reads in flight instead of three dependent round trips.
constexpr int LSQ_N = 3;
__device__ __forceinline__ void eval_lsq(const ItemW& I, const DevWorld& W, const DevStepArgs& args,
                                         const float* tile, bool movable, v2& F, bool live) {
  const uint4 w0 = I.w0, w1 = I.w1, w2 = I.w2, w3 = I.w3;
  const int n = sgpr((int)w0.y);
  const float* E = tile + (int)w0.z;
  const float dist_min = __uint_as_float(w0.w);
  const v2 ps = V(E[0], E[ROWF]);
  const int lo[LSQ_N] = {(int)(w1.x & 0xffffu), (int)(w1.x >> 16), (int)(w1.y & 0xffffu)};
  const int to[LSQ_N] = {(int)(w1.z & 0xffffu), (int)(w1.z >> 16), (int)(w1.w & 0xffffu)};
  const float half[LSQ_N] = {__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z)};
  const float thr[LSQ_N] = {__uint_as_float(w2.w), __uint_as_float(w3.z), __uint_as_float(w3.w)};
  const int idx[LSQ_N] = {(int)(w3.x & 0xffffu), (int)(w3.x >> 16), (int)(w3.y & 0xffffu)};
  v2 pl[LSQ_N];
  float cs[LSQ_N], sn[LSQ_N];
#pragma unroll
  for (int k = 0; k < LSQ_N; ++k) {  // unused slots repeat line 0 on the host: always valid rows
    const float* L = tile + lo[k];
    const float* T = tile + to[k];
    pl[k] = V(L[0], L[ROWF]);
    cs[k] = T[0];
    sn[k] = T[ROWF];
  }
  uint32_t needbits = 0;
#pragma unroll
  for (int k = 0; k < LSQ_N; ++k) {
    // In the line's frame: the sphere's centre is farther than `m` from the segment's supporting line, or farther than
    // `m` beyond one of its ends -> the closest point of the segment is farther than dist_min and the force is exactly 0
    // (core.py:2836).  Football's walls span the pitch, their bounding circles never reject anything.  Only FINITE
    // gaps skip: a NaN cos (non-finite rotation) or an infinite position makes both gaps NaN / inf.
    const float dx = pl[k].x - ps.x, dy = pl[k].y - ps.y;
#if VMAS_X_TIGHT_LS
    const float m = dist_min + kSkipSlack;
    const float along = fabsf(dx * cs[k] + dy * sn[k]) - half[k];
    const float perp = fabsf(dy * cs[k] - dx * sn[k]);
    bool need = !((along > m && along < kInf) || (perp > m && perp < kInf));
#else
    const float m = half[k] + dist_min + kSkipSlack;  // bounding circles
    bool need = !far_apart(dx * dx + dy * dy, m * m) || cs[k] != cs[k];
#endif
    bool on = k < n;
    on = on && pair_on(args, sgpr(idx[k]));
    if (lazy_noting(args) && k < n) lazy_acc_overlap(args, sgpr(idx[k]), live && circles_overlap(dx, dy, thr[k]));
    needbits |= (on && __any(need)) ? (1u << k) : 0u;
  }
  if (!needbits || (ABLATE(args) & 32)) return;
#pragma unroll
  for (int k = 0; k < LSQ_N; ++k) {
    if (needbits & (1u << k)) {  // core.py:2341-2392
      const v2 cp = closest_point_line<true>(pl[k], cs[k], sn[k], half[k], ps);
      const v2 f = contact_force(ps, cp, dist_min, W.c_coll, W.k);
      if (lazy_noting(args))  // (the padding columns of a tail tile are no environments)
        lazy_note_band(args, sgpr(idx[k]), live && !circles_overlap(pl[k].x - ps.x, pl[k].y - ps.y, thr[k]) && (f.x != 0.f || f.y != 0.f));
      if (movable) F = F + f;
    }
  }
}

// SHARED sphere-sphere pairs (both spheres dynamic), four pairs to a record, each evaluated ONCE: the force on a goes
// to the pair's two LDS rows, b's owner reads it with the sign flipped (cf(a,b) == -cf(b,a) bit for bit).
//   w0: type, n, tile offset of the first pair's rows (pair k: + 2k rows), -
//   w1: a offsets 0|1<<16, 2|3<<16, b offsets 0|1<<16, 2|3<<16   w2: r_sum 0..3   w3: pair index 0|1<<16, 2|3<<16
__device__ __forceinline__ void eval_ssp(const ItemW& I, const DevWorld& W, const DevStepArgs& args, float* tile,
                                         uint32_t* fired, bool live) {
  const uint4 w0 = I.w0, w1 = I.w1, w2 = I.w2, w3 = I.w3;
  const int n = sgpr((int)w0.y);
  const int rec = sgpr((int)w0.w);  // ordinal among the shared sphere-sphere records
  float* R = tile + (int)w0.z;
  const int oa[4] = {(int)(w1.x & 0xffffu), (int)(w1.x >> 16), (int)(w1.y & 0xffffu), (int)(w1.y >> 16)};
  const int ob[4] = {(int)(w1.z & 0xffffu), (int)(w1.z >> 16), (int)(w1.w & 0xffffu), (int)(w1.w >> 16)};
  const float rs[4] = {__uint_as_float(w2.x), __uint_as_float(w2.y), __uint_as_float(w2.z), __uint_as_float(w2.w)};
  const int idx[4] = {(int)(w3.x & 0xffffu), (int)(w3.x >> 16), (int)(w3.y & 0xffffu), (int)(w3.y >> 16)};
  v2 pa[4], pb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // unused slots repeat pair 0 on the host: always valid rows
    const float* A = tile + oa[k];
    const float* B = tile + ob[k];
    pa[k] = V(A[0], A[ROWF]);
    pb[k] = V(B[0], B[ROWF]);
  }
  uint32_t needbits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = pa[k].x - pb[k].x, dy = pa[k].y - pb[k].y;
    const float m = rs[k] + 1e-4f;  // the force is exactly 0 for dist > r_a + r_b (core.py:2836)
    bool need = !far_apart(dx * dx + dy * dy, m * m);
    bool on = k < n;
    on = on && pair_on(args, sgpr(idx[k]));
    needbits |= (on && __any(need)) ? (1u << k) : 0u;
  }
  if (ABLATE(args) & 32) needbits = 0;
  // Which pairs fired (some environment of the tile within reach) is published, four bits per record: the owners skip
  // the rows of a pair that did not fire in phase C - its force is +0 on a, -0 on b in every environment, and
  // F + (-0) == F, F + (+0) == F except for F == -0 (handled there) - so those rows are not even written.
  const bool published = rec < FIRED_RECS;
  if (published && needbits && (threadIdx.x & (TILE - 1)) == 0)
    atomicOr(fired + (rec >> 3), needbits << ((rec & 7) * 4));  // (LDS atomic, no return value)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < n && (!published || (needbits & (1u << k)))) {
      v2 f = V(0.f, 0.f);
      if (needbits & (1u << k)) {
        f = contact_force(pa[k], pb[k], rs[k], W.c_coll, W.k);
        if (lazy_noting(args)) {  // (non-finite poses: see eval_ssq)
          const bool ov = live && vnorm(pa[k] - pb[k]) <= rs[k];
          lazy_acc_overlap(args, sgpr(idx[k]), ov);
          lazy_note_band(args, sgpr(idx[k]), live && !ov && (f.x != 0.f || f.y != 0.f));
        }
      }
      R[(2 * k) * ROWF] = f.x;
      R[(2 * k + 1) * ROWF] = f.y;
    }
  }
}

// Force (and torque) one item contributes to ITS side.  LEVEL prunes code (and registers):
// 0: SS LS BS   1: + LL BL joints   2: + BB
template <int LEVEL>
__device__ __forceinline__ void eval_item(const ItemV& K, const DevWorld& W, const DevStepArgs& args,
                                          const float* tile, long env, bool live, long ld, v2& f_out, float& t_out,
                                          float& tb_out) {
  const float* A = tile + K.oa;
  const float* B = tile + K.ob;
  const v2 pa = V(A[0], A[ROWF]), pb = V(B[0], B[ROWF]);
  const float k = W.k;
  // Every joint/pair applies f to a and exactly -f to b (core.py:2839), so only the force on
  // a is computed and the b side flips its sign bit.  (Written as an integer xor on purpose:
  // hipcc 7.2 mis-folds `side ? -f : f` - the negation is dropped - which the golden parity
  // tests caught.)  Torques use the side's own lever arm.
  // A SHARED item (side 2, both entities dynamic) is evaluated once: f_out/t_out are a's, tb_out is the torque on b.
  const int side = K.side & 3;
  const uint32_t flip = side == 1 ? 0x80000000u : 0u;
  auto own = [&](v2 fa) { return V(__uint_as_float(__float_as_uint(fa.x) ^ flip), __uint_as_float(__float_as_uint(fa.y) ^ flip)); };
  auto neg = [&](v2 fa) { return V(__uint_as_float(__float_as_uint(fa.x) ^ 0x80000000u), __uint_as_float(__float_as_uint(fa.y) ^ 0x80000000u)); };
  // torques of a pair whose force on a is fa: each side's own lever arm, b's with the flipped force
  auto levers = [&](v2 lever_a, v2 lever_b, v2 fa) {
    if (side != 1) t_out = vcross(lever_a, fa);
    if (side == 1) t_out = vcross(lever_b, neg(fa));
    if (side == 2) tb_out = vcross(lever_b, neg(fa));
  };
  if (LEVEL >= 1 && K.type == TASK_JOINT) {  // _vectorized_joint_constraints core.py:2201-2292
    const float ra = A[4 * ROWF], rb = B[4 * ROWF];
    float sa, ca, sb, cb;
    sincosf(ra, &sa, &ca);
    sincosf(rb, &sb, &cb);
    const v2 pja = pa + rotate(V(K.p0, K.p1), ca, sa);  // joints.py:209-216
    const v2 pjb = pb + rotate(V(K.q0, K.q1), cb, sb);
    const v2 f_att = constraint_force<true>(pja, pjb, K.p2, W.c_joint_att, k);
    const v2 f_rep = constraint_force<false>(pja, pjb, K.p2, W.c_joint_rep, k);
    const v2 fa = f_att + f_rep;  // (-f_att) + (-f_rep) == -(f_att + f_rep) bitwise
    levers(pja - pa, pjb - pb, fa);
    if (K.flags & IT_LOCK) {
      float fr = K.p3;
      if (args.joint_fixed_rot && live) fr = args.joint_fixed_rot[(long)K.index * ld + env];
      const float lock = constraint_torque(ra, rb + fr, W.tcf);
      t_out = t_out + (side == 1 ? lock : -lock);
      if (side == 2) tb_out = tb_out + lock;
    }
    f_out = own(fa);
    return;
  }
  if (!pair_on(args, K.index)) return;
  const float* TA = tile + K.tra;
  const float* TB = tile + K.trb;
  {  // conservative per-environment broad phase: beyond it the force is exactly zero
    const float dx = pa.x - pb.x, dy = pa.y - pb.y;
    if (lazy_noting(args)) lazy_acc_overlap(args, K.index, live && circles_overlap(dx, dy, K.q0));
    bool need = !far_apart(dx * dx + dy * dy, K.thr2);
    if (VMAS_X_TIGHT_LS && K.type == VMAS_PAIR_LS) {  // a is a line: the sphere's gaps to the segment in the line's frame (see eval_lsq)
      const float along = fabsf(dx * TA[0] + dy * TA[ROWF]) - K.p0;
      const float perp = fabsf(dy * TA[0] - dx * TA[ROWF]);
      need = need && !((along > K.reach && along < kInf) || (perp > K.reach && perp < kInf));
    }
    if (K.type >= VMAS_PAIR_BS) {  // a is a box: test against the oriented box, much tighter
      if (K.type == VMAS_PAIR_BL) {
        const float gap = seg_obb_gap(pb, TB[0], TB[ROWF], K.p2, pa, TA[0], TA[ROWF], K.p0 * 0.5f, K.p1 * 0.5f);
        need = need && !(gap > K.reach);
      } else {
        const float d2 = obb_dist2(pb, pa, TA[0], TA[ROWF], K.p0 * 0.5f, K.p1 * 0.5f);
        need = need && !(d2 > K.reach * K.reach);
      }
    }
    // a Line/Box with a non-finite rotation has NaN edges at any distance
    if (K.type != VMAS_PAIR_SS) need = need || TA[0] != TA[0];
    if (K.type == VMAS_PAIR_LL || K.type == VMAS_PAIR_BL || K.type == VMAS_PAIR_BB) need = need || TB[0] != TB[0];
    if (!__any(need) || (ABLATE(args) & 32)) return;  // 32: broad phase only (profiling)
  }
  switch (K.type) {
    case VMAS_PAIR_SS: {  // core.py:2294-2339; p0 = r_a + r_b
      f_out = own(contact_force(pa, pb, K.p0, W.c_coll, k));
    } break;
    case VMAS_PAIR_LS: {  // a = line, b = sphere; p0 = L/2, p1 = r + LMD  core.py:2341-2392
      const v2 cp = closest_point_line<true>(pa, TA[0], TA[ROWF], K.p0, pb);
      const v2 f = own(-contact_force(pb, cp, K.p1, W.c_coll, k));
      f_out = f;
      t_out = side == 1 ? 0.f : vcross(cp - pa, f);
    } break;
    case VMAS_PAIR_BS: {  // a = box, b = sphere; p0 = L, p1 = W, p2 = r + LMD  core.py:2459-2552
      seg_t be[4];
      box_edges(pa, TA[0], TA[ROWF], TA[2 * ROWF], TA[3 * ROWF], K.p0, K.p1, be);
      const v2 cp = closest_point_box(be, pb);
      v2 ip = cp;
      float d = 0.f;
      if (!(K.flags & IT_A_HOLLOW)) ip = inner_point_box(pb, cp, pa, d);
      const v2 f = own(-contact_force(pb, ip, K.p2 + d, W.c_coll, k));
      f_out = f;
      t_out = side == 1 ? 0.f : vcross(cp - pa, f);
    } break;
    case VMAS_PAIR_LL:
      if (LEVEL >= 1) {  // p0 = La/2, p1 = Lb/2  core.py:2394-2457
        seg_t l1 = {pa, TA[0], TA[ROWF], K.p0};
        seg_t l2 = {pb, TB[0], TB[ROWF], K.p1};
        v2 qa, qb;
        closest_points_seg_seg(l1, l2, qa, qb);
        const v2 fa = contact_force(qa, qb, kLineMinDist, W.c_coll, k);
        f_out = own(fa);
        levers(qa - pa, qb - pb, fa);
      }
      break;
    case VMAS_PAIR_BL:
      if (LEVEL >= 1) {  // a = box, b = line; p0 = L, p1 = W, p2 = Lb/2  core.py:2554-2653
        seg_t be[4];
        box_edges(pa, TA[0], TA[ROWF], TA[2 * ROWF], TA[3 * ROWF], K.p0, K.p1, be);
        seg_t ln = {pb, TB[0], TB[ROWF], K.p2};
        v2 qb, ql;
        closest_seg_box(be, ln, qb, ql);
        v2 ip = qb;
        float d = 0.f;
        if (!(K.flags & IT_A_HOLLOW)) ip = inner_point_box(ql, qb, pa, d);
        const v2 fa = contact_force(ip, ql, kLineMinDist + d, W.c_coll, k);
        f_out = own(fa);
        levers(qb - pa, ql - pb, fa);
      }
      break;
    case VMAS_PAIR_BB:
      if (LEVEL >= 2) {  // p0,p1 = L,W of a; p2,p3 = L,W of b  core.py:2655-2786
        seg_t ea[4], eb[4];
        box_edges(pa, TA[0], TA[ROWF], TA[2 * ROWF], TA[3 * ROWF], K.p0, K.p1, ea);
        box_edges(pb, TB[0], TB[ROWF], TB[2 * ROWF], TB[3 * ROWF], K.p2, K.p3, eb);
        v2 qa, qb;
        closest_box_box(ea, eb, qa, qb);
        v2 ia = qa, ib = qb;
        float da = 0.f, db = 0.f;
        if (!(K.flags & IT_A_HOLLOW)) ia = inner_point_box(qb, qa, pa, da);
        if (!(K.flags & IT_B_HOLLOW)) ib = inner_point_box(qa, qb, pb, db);
        const v2 fa = contact_force(ia, ib, da + db + kLineMinDist, W.c_coll, k);
        f_out = own(fa);
        levers(qa - pa, qb - pb, fa);
      }
      break;
    default: break;
  }
  // the lazy form's optimistic pass (K.q0: circles_overlap's threshold of the pair; the padding columns of a tail tile are no
  // environments)
  if (lazy_noting(args))
    lazy_note_band(args, K.index, live && !circles_overlap(pa.x - pb.x, pa.y - pb.y, K.q0) &&
                                      (f_out.x != 0.f || f_out.y != 0.f || t_out != 0.f || tb_out != 0.f));
}

// ---- the lazy exact broad phase around an optimistic gather pass (vmas_env_device.h), for the interpreter and its
// world-specialised forms.
// LDS layout at lz_words (lzp = pair words rounded up to 4): [2 parities][overlap words | band words] | need | batch | flag [4].
//   lazy_overlap_blob  IN FRONT of the gather (the tile's positions are in LDS): which pairs some environment of the tile has
//                      overlapping bounding circles - the pair table of the blob, pair p by wave p mod nw, two in flight -
//                      then a block barrier and the tile's words go out (lazy_publish): they travel while the tile gathers,
//                      and the tiles that will ask find them there.  (Published behind the gather in the first version,
//                      the write-through store's latency sat on the launch's tail: transport 16 384: 4.5 -> 6.7 us.)
//   lazy_after_pass    behind the barrier that ends the gather: did an item note a band event (lz_s)?  Then one wave asks
//                      the batch (lazy_collect).  Returns true - with args.lz_g = the batch's words and the gather counter
//                      re-armed - if the pass has to be made again with a pair off.
constexpr int SPEC_LZP = 4;                          // the specialised kernels: worlds of at most 128 pairs
constexpr int SPEC_TAIL_WORDS = 8 + 6 * SPEC_LZP + 4;  // their LDS behind the tile: counters [4] | fired words [4] | the lazy words
__device__ __forceinline__ void lazy_overlap_blob(const DevStepArgs& args, const uint32_t* table, int n_pairs, const float* tile,
                                                  bool live, int wv, int nw, int pass) {
  for (int p0 = 2 * wv; p0 < n_pairs; p0 += 2 * nw) {
    const int p1 = p0 + 1 < n_pairs ? p0 + 1 : p0;
    const uint4 d0 = *(const uint4*)(table + 4 * p0), d1 = *(const uint4*)(table + 4 * p1);
    const float* A0 = tile + (int)d0.x; const float* B0 = tile + (int)d0.y;
    const float* A1 = tile + (int)d1.x; const float* B1 = tile + (int)d1.y;
    const float ax0 = A0[0], ay0 = A0[ROWF], bx0 = B0[0], by0 = B0[ROWF], ax1 = A1[0], ay1 = A1[ROWF], bx1 = B1[0], by1 = B1[ROWF];
    const bool h0 = live && circles_overlap(ax0 - bx0, ay0 - by0, __uint_as_float(d0.z));
    const bool h1 = live && circles_overlap(ax1 - bx1, ay1 - by1, __uint_as_float(d1.z));
    if (__any(h0) && (threadIdx.x & (TILE - 1)) == 0) atomicOr(&args.lz_x[p0 >> 5], 1u << (p0 & 31));  // (LDS)
    if (__any(h1) && (threadIdx.x & (TILE - 1)) == 0) atomicOr(&args.lz_x[p1 >> 5], 1u << (p1 & 31));
  }
  __syncthreads();
  extern __shared__ uint32_t lazy_lds_base[];
  lazy_publish(args.lz, pass, (int)(args.lz_x - lazy_lds_base));
}
__device__ __forceinline__ bool lazy_after_pass(DevStepArgs& args, int pass, uint32_t* lz_words, int lzp, int* c_gather, int c_init) {
  const int mw = args.lz.words;
  bool open = false;
  for (int w = 0; w < mw; ++w) open = open || (args.lz_s[w] & ~args.lz_x[w]) != 0u;  // (uniform LDS reads)
  if (!open) return false;
  uint32_t* need = lz_words + 4 * lzp;
  uint32_t* got = need + lzp;
  uint32_t* flag = got + lzp;  // [4]: collect's two scratch words | - | -
  for (int w = threadIdx.x; w < mw; w += blockDim.x) need[w] = args.lz_s[w] & ~args.lz_x[w];  // (lz_s is complete: the caller's barrier)
  __syncthreads();
  extern __shared__ uint32_t lazy_lds_base[];
  const int again = lazy_collect(args.lz, pass, (int)(need - lazy_lds_base), (int)(got - lazy_lds_base), (int)(flag - lazy_lds_base));
  if (again == 0) return false;
  __syncthreads();
  for (int w = threadIdx.x; w < mw; w += blockDim.x) got[w] = ~(need[w] & ~got[w]);  // every pair on but the needed ones that stayed clear
  if (threadIdx.x == 0) c_gather[0] = c_init;
  __syncthreads();
  args.lz_g = got;
  return true;
}
