// vmas_step_types.h - device-side types shared by the step kernels (vmas_hip.hip: the schedule interpreter and its
// world-specialised form; vmas_compact.hip: the lane-compacted form) and the host code that launches them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vmas_hip.h"
#include "vmas_env_device.h"

using namespace vmas;

struct DevItem;

constexpr int TILE = 64;       // environments per block = lanes per wave
constexpr int MAX_WAVES = 16;  // waves (workers) per tile (8 for the register-heavy box-box level)
constexpr float kSkipSlack = 1e-3f;  // fp slack of the conservative broad-phase distances
constexpr int ROWF = TILE;     // floats per LDS row

struct DevEntity {
  uint32_t flags;
  int32_t shape;
  int32_t agent_index;
  int32_t tr_off;  // tile offset of its 4 trig rows, -1 for spheres
  float mass, inertia, one_minus_drag;
  float max_speed, v_range, lin_friction, ang_friction;
  float gx, gy;
  float max_f, f_range, max_t, t_range;
};

struct DevWorld {
  int32_t nE, nA, substeps;
  int32_t off_af;  // tile offset of the agent force rows
  // Which entities need cos/sin rows (bit e: a Line or a Box | a Box), straight from the kernel arguments so that the load
  // phase does not wait for the descriptor blob: entity e's four trig rows start at row_tr + 4 * popcount(trig_mask below e).
  // trig_in_args == 0: more than 64 entities, shapes and offsets come from the blob behind a barrier of their own.
  unsigned long long trig_mask, box_mask;
  int32_t trig_in_args, row_tr;
  float sub_dt, gx, gy;
  int32_t has_gravity;
  float xs, ys;  // NaN = unbounded
  float k, tcf;
  float c_coll, c_joint_att, c_joint_rep;  // fp32(sign * force_multiplier)
  // One descriptor blob per schedule, staged into LDS by the whole block with coalesced loads
  // while the state rows are in flight.  (Scalar/vector loads of descriptors from a cache that
  // is cold at every launch cost ~600 cycles per dependent fetch - 2-4 us of a 13 us step.)
  //   [ents | segs | owned | items]   word offsets below; counters live right after the blob
  const uint32_t* blob;
  int32_t blob_words;   // words staged (items only when they fit the LDS budget)
  int32_t off_blob;     // tile offset (floats) of the blob copy in LDS
  int32_t b_ent, b_segs, b_owned, b_refs, b_items;
  int32_t b_pairs, n_pairs;  // the pair table in front of the items: [n_pairs][4] = tile offsets of a / b, circles_overlap's
                             // threshold, - ; then 4 words {n_pairs, -, -, -} (b_pairs = b_items - 4 - 4 * n_pairs)
  int32_t n_segs, n_owned;
  int32_t items_in_lds;
  int32_t fired_recs;    // shared sphere-sphere records (eval_ssp) that publish which of their pairs fired (<= 16)
  const DevItem* items;  // global copy, used when the item list is too big for LDS
};

// (DevMaskPair, DevLidar, DevTarget: vmas_env_device.h - the navigation epilogue uses them too)
struct DevStepArgs {
  const uint32_t* pair_mask;
  // in-kernel exact broad phase (vmas_world_step with exact_broad_phase on a grid of at most one tile per CU): the
  // batch-global `.any()` of World.collides (core.py:2797-2801) evaluated at the top of every substep by all tiles
  // together - ONE device-scope 64-bit atomic per tile and pair word carries the tile's arrival bit and its pair bits into
  // the substep's slot (grid_bits_publish / grid_bits_collect, vmas_env_device.h); the batch's words are then read from LDS
  uint32_t* sync;              // [1] gave-up flag; from word 4: ring of four slots of [tile groups][mask_words] 64-bit words
  const DevMaskPair* mpairs;   // the world's static pairs with their bounding-circle sums
  uint32_t seq0;               // barrier sequence number of this launch's first substep
  uint32_t* gave_up;           // host-mapped word: set (system scope) when a grid barrier gave up waiting; the host reads it
                               // without a synchronisation at the next call on this world and fails that call
  int32_t n_mpairs, mask_words;
  const float* joint_fixed_rot;
  const float* entity_gravity;
  int32_t first_substep, n_substeps;
  int32_t n_steps;    // > 1: persistent rollout, the tile stays in LDS between steps
  long ft_stride;     // floats between the agent-force slabs of consecutive steps
  // the LAZY exact broad phase (vmas_env_device.h, LazyArgs): filled by the host (lz.slots == NULL: off) ...
  LazyArgs lz;
  // ... and, set by the kernel itself, its words in LDS: the pairs some environment of this tile overlaps (lz_x, [words]:
  // made in front of the gather and published at once), the pairs one of its environments is in the band of without an
  // overlapping environment of the tile (lz_s: noted by the items of the optimistic pass; NULL: not noting) and, in the pass
  // made again for a tile that had to, the batch's words (lz_g; NULL in the optimistic pass)
  uint32_t* lz_x;
  uint32_t* lz_s;
  const uint32_t* lz_g;
  // (the world-specialised kernels make the overlap words WHILE they gather, from the positions an item has loaded anyway:
  //  the wave's own words, in registers - pair indices are immediates there; flushed to lz_x behind the wave's last segment.
  //  NULL: lz_x was made in front of the gather - lazy_overlap_blob, the interpreter)
  uint32_t* lz_acc;
  unsigned long long* contacts;  // compacted kernel: device counter, + the contacts of every (tile, substep) (NULL: not counted)
  unsigned long long* trace;  // profiling only (env VMAS_TRACE): per-wave s_memtime stamps
  int32_t ablate;  // profiling only (env VMAS_ABLATE): 1 skip items, 2 skip integration, 4 skip prologue
};

// The Environment.step() stages fused around the physics (vmas_world_step_env): action ingest as the
// kernel's prologue, one scenario's reward / observation / done as its epilogue on the LDS tile.
enum { ENV_NONE = 0, ENV_BALANCE = 1, ENV_TRANSPORT = 2, ENV_INGEST = 3, ENV_NAVIGATION = 4, ENV_FOOTBALL = 5 };  // 3: prologue only
struct DevEnv {
  int32_t has_ingest;
  int32_t ablate;       // profiling only (env VMAS_ENV_ABLATE)
  int32_t scratch_off;  // floats from the LDS base to the epilogue's scratch (after the step's own LDS)
  int32_t gated;        // vmas_world_step_env_gated: the launch does NOTHING if *err_flags != 0 when it starts (the word of a
                        // validation launched in front of it on the same stream: a refused action never reaches the world)
  // `ingest.agents` is re-ordered by the host: slot a belongs to AGENT a (action == action_index == NULL: no action
  // for it), so the prologue reads its slot with one kernarg fetch, no indirection.  Scripts: agent -> script or -1.
  int8_t script_of_agent[VMAS_ENV_MAX_AGENTS];
  uint32_t* err_flags;
  unsigned long long* trace;  // profiling only (VMAS_TRACE=2): per-wave s_memtime stamps of the world-specialised kernel
  VmasIngestArgs ingest;
  union {
    struct { VmasBalanceDesc d; VmasBalanceBuffers o; } balance;
    struct { VmasTransportDesc d; VmasTransportBuffers o; } transport;
    struct { VmasNavigationDesc d; VmasNavigationBuffers o; NavWorld w; } navigation;
    struct { VmasFootballDesc d; VmasFootballBuffers o; } football;  // (epilogue of the compact kernel only, vmas_compact.h)
  };
};
struct NoEnv {};
// block-uniform: every thread of the launch reads the same word (a scalar load; the validation's launch in front of this one
// on the same stream wrote it - kernel boundaries make it visible)
__device__ __forceinline__ bool env_gate_closed(const DevEnv& E) {
  return E.gated != 0 && __builtin_amdgcn_readfirstlane((int)*(const volatile uint32_t*)E.err_flags) != 0;
}
__device__ __forceinline__ bool env_gate_closed(const NoEnv&) { return false; }

// What a code object compiled at run time (specialize.py) must agree with the library on besides the schedule: the layout of
// the kernel arguments.  Both sides compute this from THEIR headers - the library when it was built, the specialisation when
// it is compiled - and vmas_world_load_spec refuses a module whose word differs (a library out of step with the headers on
// disk would otherwise pass the word-for-word schedule check and read its arguments through the wrong offsets).
constexpr uint32_t kLayoutHash =
    (uint32_t)sizeof(DevWorld) * 0x9E3779B1u ^ (uint32_t)sizeof(DevEnv) * 0x85EBCA77u ^ (uint32_t)sizeof(DevStepArgs) * 0xC2B2AE3Du ^
    (uint32_t)__builtin_offsetof(DevWorld, blob) * 0x27D4EB2Fu ^ (uint32_t)__builtin_offsetof(DevEnv, ingest) * 0x165667B1u ^
    (uint32_t)__builtin_offsetof(DevEnv, balance) * 0xD3A2646Cu ^ (uint32_t)sizeof(VmasActionSlot) * 0xFD7046C5u ^
    (uint32_t)VMAS_ABI_VERSION * 0xB55A4F09u ^ (uint32_t)sizeof(LazyArgs) * 0x68E31DA4u;

// Profiling knobs (env VMAS_ABLATE / VMAS_ENV_ABLATE) exist only in -DVMAS_PROFILE builds (scripts/gpu_ablate.sh): in the
// product build they are the literal 0, so their tests - and the scalar register that carried them through every loop -
// are compiled out.
#ifdef VMAS_PROFILE
#define ABLATE(a) ((a).ablate)
#else
#define ABLATE(a) 0
#endif

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
// One word of the pair mask: the caller's recorded mask (global memory, read-only during the launch) or the copy of the
// batch's words the in-kernel exact broad phase leaves in LDS behind its grid barrier (grid_bits_collect) - a plain load
// either way (until round 4 the exact form was read straight from the grid's slots, an agent-scope atomic load - a round
// trip to the memory side - in front of every item).
__device__ __forceinline__ uint32_t mask_word(const uint32_t* mask, int w) { return mask[w]; }
// Is pair `idx` processed at all in this pass?  The caller's recorded mask / the barrier form's LDS copy (pair_mask), and -
// only in the pass a tile makes AGAIN under the lazy form - the batch's words behind lazy_collect (lz_g, LDS).
__device__ __forceinline__ bool pair_on(const DevStepArgs& a, int idx) {
  if (a.pair_mask && !((mask_word(a.pair_mask, idx >> 5) >> (idx & 31)) & 1u)) return false;
  if (a.lz_g && !((a.lz_g[idx >> 5] >> (idx & 31)) & 1u)) return false;
  return true;
}
// The lazy form's note of an item of the optimistic pass whose narrow phase ran (uniform `idx`; lane = environment): some
// environment is in the pair's band - bounding circles apart, force or torque not zero (a NaN counts: the reference would
// not have evaluated the pair for it either) - and no environment of the TILE has the circles overlapping (lz_x: the tile's
// own overlap words, complete before the gather starts - lazy_overlap_*): the batch's bit of the pair decides.
__device__ __forceinline__ void lazy_note_band(const DevStepArgs& a, int idx, bool lane_in_band) {
  if (!__any(lane_in_band)) return;
  if (a.lz_acc == nullptr && ((a.lz_x[idx >> 5] >> (idx & 31)) & 1u)) return;  // (lz_x complete: an environment of the tile overlaps)
  if ((threadIdx.x & 63) == 0) atomicOr(&a.lz_s[idx >> 5], 1u << (idx & 31));  // (LDS, no return value)
}
// (specialised kernels) some environment of the tile has pair `idx`'s bounding circles overlapping: into the wave's own words
__device__ __forceinline__ void lazy_acc_overlap(const DevStepArgs& a, int idx, bool lane_overlaps) {
  if (a.lz_acc != nullptr && __any(lane_overlaps)) a.lz_acc[idx >> 5] |= 1u << (idx & 31);
}
// is the optimistic pass of the lazy form running (the items note band events)?
__device__ __forceinline__ bool lazy_noting(const DevStepArgs& a) { return a.lz_s != nullptr && a.lz_g == nullptr; }

// A pair may be skipped only on a FINITE squared distance beyond its bound: a NaN or an infinite operand must reach the
// narrow phase, where the reference's own arithmetic decides (inf * 0, cos(inf) ... = NaN poisons the pair however far
// apart the shapes are).  Together with the NaN checks on the cos rows of Lines and Boxes this is why no separate
// "environment has a non-finite pose" flag is needed.
__device__ __forceinline__ bool far_apart(float d2, float thr2) { return d2 > thr2 && d2 < kInf; }

struct EntV {
  uint32_t flags; int32_t shape, agent_index, tr_off;  // scalar
  float mass, inertia, one_minus_drag, max_speed, v_range, lin_friction, ang_friction, gx, gy, max_f, f_range, max_t, t_range;
};
__device__ __forceinline__ EntV load_ent(const uint32_t* p) {
  EntV D;
  D.flags = (uint32_t)sgpr((int)p[0]); D.shape = sgpr((int)p[1]); D.agent_index = sgpr((int)p[2]); D.tr_off = sgpr((int)p[3]);
  D.mass = __uint_as_float(p[4]); D.inertia = __uint_as_float(p[5]); D.one_minus_drag = __uint_as_float(p[6]);
  D.max_speed = __uint_as_float(p[7]); D.v_range = __uint_as_float(p[8]);
  D.lin_friction = __uint_as_float(p[9]); D.ang_friction = __uint_as_float(p[10]);
  D.gx = __uint_as_float(p[11]); D.gy = __uint_as_float(p[12]);
  D.max_f = __uint_as_float(p[13]); D.f_range = __uint_as_float(p[14]); D.max_t = __uint_as_float(p[15]); D.t_range = __uint_as_float(p[16]);
  return D;
}
static_assert(sizeof(DevEntity) == 68, "descriptor layout");
