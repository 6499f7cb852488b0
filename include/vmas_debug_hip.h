/* vmas_debug_hip.h - TEST AND PROFILING HOOKS of libvmas_hip.so.  Not part of the drop-in boundary (include/vmas_hip.h,
 * include/vmas_env_hip.h): nothing in the product path calls them.  They expose device-side primitives of the step
 * kernel so that tests/test_hip_math.py can bound their accuracy, and the s_memtime stamps of profiling builds. */
#ifndef VMAS_DEBUG_HIP_H
#define VMAS_DEBUG_HIP_H
#include <stdint.h>
#include "vmas_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* out[i] = op(a[i], b[i]) with the device routines of csrc/vmas_device.h (device pointers; b may be NULL for unary ops) */
#define VMAS_MATH_SOFTPLUS 0 /* softplus0(a)       = torch.logaddexp(0, a), core.py:2821 */
#define VMAS_MATH_SQRT 1     /* sqrt_n(a)          the bare v_sqrt_f32 used for squared lengths */
#define VMAS_MATH_DIV 2      /* a / rcp_of(b)      the division sequence without exponent scaling */
#define VMAS_MATH_NORM 3     /* norm2(a, b)        = torch.linalg.vector_norm over a size-2 dim */
#define VMAS_MATH_COS 4      /* cos(a) as write_trig computes it (sincosf) */
#define VMAS_MATH_SIN 5      /* sin(a) as write_trig computes it (sincosf) */
int vmas_debug_math(int32_t op, const float* a, const float* b, float* out, int32_t n, void* stream);

/* The schedule the library planned for a world (waves per tile, shared rows, segments, items): the descriptor blob as it
 * is staged into LDS (`words`, may be NULL to query the size) and meta[24] = {waves per tile, share mode, kernel level,
 * blob words, b_ent, b_segs, b_owned, b_refs, b_items, n_segs, n_owned, tile rows, fired records, items in LDS, nE, nA,
 * off_af, row_tr, trig_mask lo/hi, box_mask lo/hi, trig_in_args, specialisation id (-1: none)}.  Works on PLANNING worlds
 * too: vmas_world_create(desc, batch, device_id = -1, &w) builds the schedules on the host without touching a GPU (such a
 * world cannot be stepped).  scripts/gen_spec.py generates csrc/vmas_spec_gen.h - the tables of the world-specialised
 * kernel - from it. */
int vmas_debug_schedule(VmasWorld* w, uint32_t* words, int64_t capacity, int32_t* meta /* [24] */);

/* Sets the world's "a grid-wide barrier gave up waiting" word as a timed-out barrier would (include/vmas_hip.h,
 * vmas_world_exact_status): the next launch on the world must fail loudly.  For the test of exactly that. */
int vmas_debug_force_gave_up(VmasWorld* w);

/* The plan of the lane-compacted step kernel (csrc/vmas_compact.h; works on planning worlds): the table blob as it is staged
 * into LDS (`words`, may be NULL to query the size) and meta[16] = {waves per tile, owned entities per wave, LDS bytes per tile,
 * blob words, pairs, owned entities, word offset of the load table, of the waves' unit ranges, of the unit records, dynamic /
 * static-in-a-pair / line entity masks, tile row of the agent forces, of the first cos row, has_torque, entities}.  The load
 * table: per (batch b, wave w) four words, word j for entity w + (4 b + j) * waves - bit 0 in the tile, 1 dynamic, 2 line |
 * first tile row << 3 | cos row << 13 | entity << 23.  Fails for worlds the compacted kernel does not take. */
int vmas_debug_compact_plan(VmasWorld* w, uint32_t* words, int64_t capacity, int64_t* meta /* [16] */);

/* football's Environment.step / rollout (vmas_world_step_env / vmas_world_rollout_env with VMAS_POST_FOOTBALL) has two forms
 * with the same device functions: 0 one launch (the post-step as the compacted step kernel's epilogue: every K-step rollout,
 * single steps up to one tile per CU), 1 two launches per step (step kernel, then the stand-alone post-step kernel: single
 * steps beyond).  form = 0 / 1 forces one, -1 gives the choice back to the library.  For the test that pins them against each
 * other bit for bit. */
int vmas_debug_football_form(VmasWorld* w, int32_t form);

/* The adaptive choice between the lane-compacted step kernel and the interpreter (vmas_world_set_compact(-1), the default):
 * out[0] = contacts counted by plain compacted launches, out[1] = (tile, substep)s those launches had, out[2] = times the
 * world was sent to the interpreter, out[3] = plain launches it still stays there.  Synchronises the device. */
int vmas_debug_compact_stats(VmasWorld* w, int64_t out[4]);
/* The lazy exact broad phase's counters since the world was made (synchronises the device): out[0] launches made with it,
 * out[1] tiles that had to ask for the batch's words, out[2] ... and found a pair off for the whole batch (their pass was
 * made again / the pair's contacts left out), out[3] polls that had to be repeated (a needed tile had not arrived yet). */
int vmas_debug_lazy_stats(VmasWorld* w, int64_t out[4]);

/* VMAS_TRACE=1 in a -DVMAS_TRACE build: copy out the per-wave s_memtime stamps of the last launch */
int vmas_debug_trace(VmasWorld* w, unsigned long long* host, int64_t n_words);

#ifdef __cplusplus
}
#endif
#endif /* VMAS_DEBUG_HIP_H */
