/*
 * vmas_hip.h - C ABI of the MI355X-native VMAS physics step ("libvmas_hip.so").
 *
 * The reference (proroklab/VectorizedMultiAgentSimulator 1.5.2) has no FFI: its
 * seam is the Python method `World.step(self) -> None` (vmas/simulator/core.py:1972),
 * called from `Environment.step` (vmas/simulator/environment/environment.py:395),
 * plus the sensor side call `World.cast_rays` (core.py:1662) made from
 * `Lidar.measure` (vmas/simulator/sensors.py:101-123).  The entry points below are
 * what a ctypes binding for that seam binds (see INTEGRATION.md for the stub a
 * maintainer would add to the reference).  Plain pointers and sizes only - no
 * torch types; PyTorch tensors are merely the owners of the device buffers.
 *
 * The same structs are consumed by the CPU oracle (oracle/vmas_oracle.c), which
 * is TEST INFRASTRUCTURE and exports `vmas_oracle_*` twins of the compute calls.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; `vmas_last_error()` then
 *     returns a thread-local, NUL-terminated description.  No exceptions cross
 *     the ABI.  The step itself never fails on data (NaNs propagate, exactly as
 *     in the reference, core.py:1972-2015).
 *   - all device work is enqueued asynchronously on `stream` (a hipStream_t passed
 *     as void*; NULL = the null stream).  Nothing in the step path allocates or
 *     synchronises.
 *   - all arithmetic is IEEE fp32 with FMA contraction disabled, operation order
 *     as in the reference.
 *
 * Packed state layout in HBM (structure of arrays, environment index fastest):
 *     state[(e * VMAS_STATE_FIELDS + f) * ld + env]      e < n_entities, env < batch
 *   f: 0 pos.x  1 pos.y  2 vel.x  3 vel.y  4 rot  5 ang_vel          (core.py:206-316)
 *     agent_ft[(a * VMAS_AGENT_FIELDS + f) * ld + env]   a < n_agents
 *   f: 0 force.x  1 force.y  2 torque                                (core.py:320-411)
 *   `ld` (plane stride, in floats) >= batch; a multiple of 64 keeps every plane
 *   256-byte aligned so that a wavefront's 64 lanes read one aligned segment.
 */
#ifndef VMAS_HIP_H
#define VMAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMAS_ABI_VERSION 3

#define VMAS_STATE_FIELDS 6
#define VMAS_AGENT_FIELDS 3

/* shape codes (core.py:103-203) */
#define VMAS_SHAPE_SPHERE 0
#define VMAS_SHAPE_BOX 1
#define VMAS_SHAPE_LINE 2

/* VmasEntityDesc.flags */
#define VMAS_F_MOVABLE (1u << 0)      /* Entity.movable            core.py:648 */
#define VMAS_F_ROTATABLE (1u << 1)    /* Entity.rotatable          core.py:672 */
#define VMAS_F_AGENT (1u << 2)        /* isinstance(e, Agent)      core.py:1996 */
#define VMAS_F_HOLLOW (1u << 3)       /* Box.hollow                core.py:110 */
#define VMAS_F_MAX_SPEED (1u << 4)    /* max_speed is not None     core.py:2872 */
#define VMAS_F_V_RANGE (1u << 5)      /* v_range is not None       core.py:2876 */
#define VMAS_F_LIN_FRICTION (1u << 6) /* entity or world coeff     core.py:2075-2088 */
#define VMAS_F_ANG_FRICTION (1u << 7) /*                           core.py:2089-2102 */
#define VMAS_F_GRAVITY (1u << 8)      /* entity.gravity is not None core.py:2049 */
#define VMAS_F_MAX_F (1u << 9)        /* core.py:2020 */
#define VMAS_F_F_RANGE (1u << 10)     /* core.py:2024 */
#define VMAS_F_MAX_T (1u << 11)       /* core.py:2032 */
#define VMAS_F_T_RANGE (1u << 12)     /* core.py:2036 */

/* pair type codes, in the reference's accumulation order (core.py:2178-2189) */
#define VMAS_PAIR_SS 0 /* (a,b) = spheres in entity order      core.py:2294 */
#define VMAS_PAIR_LS 1 /* a = line, b = sphere                 core.py:2341 */
#define VMAS_PAIR_LL 2 /* (a,b) = lines in entity order        core.py:2394 */
#define VMAS_PAIR_BS 3 /* a = box,  b = sphere                 core.py:2459 */
#define VMAS_PAIR_BL 4 /* a = box,  b = line                   core.py:2554 */
#define VMAS_PAIR_BB 5 /* (a,b) = boxes in entity order        core.py:2655 */

typedef struct VmasEntityDesc {
  uint32_t flags;
  int32_t shape;       /* VMAS_SHAPE_* */
  int32_t agent_index; /* row in agent_ft, or -1 for landmarks */
  float mass;          /* Entity.mass                          core.py:630 */
  float inertia;       /* Shape.moment_of_inertia(mass)        core.py:123,160,187 */
  float length;        /* Box/Line length                      */
  float width;         /* Box width                            */
  float radius;        /* Sphere radius                        */
  float bound_radius;  /* Shape.circumscribed_radius()         core.py:126,163,190 */
  float one_minus_drag; /* fp32(1 - (entity.drag ?? world.drag)) core.py:2865-2869 */
  float max_speed, v_range;
  float lin_friction, ang_friction; /* effective coefficients  core.py:2075-2102 */
  float gravity[2];                 /* constant entity gravity core.py:2049-2052 */
  float max_f, f_range, max_t, t_range; /*                     core.py:2018-2041 */
} VmasEntityDesc;

/* One statically collidable pair: everything of World.collides (core.py:2788-2796)
 * that does not depend on the state, evaluated once by the host, bucketed and
 * ordered exactly as core.py:2112-2174 discovers it (type-major, discovery order
 * inside a type). */
typedef struct VmasPairDesc {
  int32_t a, b; /* entity indices, role order given by the type code */
  int32_t type; /* VMAS_PAIR_* */
  /* fp32(a.shape.circumscribed_radius() + b.shape.circumscribed_radius()), the sum
   * taken in double as Python does before torch rounds it (core.py:2797-2799) */
  float bound_sum;
} VmasPairDesc;

/* One JointConstraint (vmas/simulator/joints.py:148-216), in the order
 * core.py:2112-2120 discovers them. */
typedef struct VmasJointDesc {
  int32_t a, b;
  float delta_a[2]; /* Shape.get_delta_from_anchor(anchor_a), body frame */
  float delta_b[2];
  float dist;           /* JointConstraint.dist (0 for every in-tree joint) */
  int32_t rotate;       /* 0 => rotation lock torque is applied core.py:2273-2282 */
  float fixed_rotation; /* used when no per-env array is passed */
} VmasJointDesc;

typedef struct VmasWorldDesc {
  int32_t abi_version; /* VMAS_ABI_VERSION */
  int32_t n_entities, n_agents, n_pairs, n_joints;
  int32_t substeps;    /* World._substeps                      core.py:1124 */
  float sub_dt;        /* fp32(dt / substeps)                  core.py:1125 */
  float gravity[2];    /* World._gravity                       core.py:1129 */
  int32_t has_gravity; /* not (gravity == 0).all()             core.py:2045 */
  float x_semidim, y_semidim; /* NaN = unbounded               core.py:2881-2895 */
  float collision_force, joint_force, contact_margin, torque_constraint_force;
  const VmasEntityDesc* entities; /* [n_entities], landmarks then agents core.py:1220 */
  const VmasPairDesc* pairs;      /* [n_pairs]  */
  const VmasJointDesc* joints;    /* [n_joints] */
} VmasWorldDesc;

/* Optional per-call inputs of the step.  All pointers are device pointers and may
 * be NULL. */
typedef struct VmasStepArgs {
  /* bit p set <=> pair p passed the batch-global bounding-circle test of
   * core.py:2797-2801 (filled by vmas_world_pair_mask).  NULL = every static pair
   * is evaluated in every environment (the reference's behaviour whenever at
   * least one environment of the batch has the pair's circles overlapping).
   * ceil(n_pairs/32) words. */
  const uint32_t* pair_mask;
  /* per-env JointConstraint.fixed_rotation, [n_joints][ld] (joints.py:141-144) */
  const float* joint_fixed_rot;
  /* per-env entity gravity, [n_entities][2][ld] (core.py:594-601, wind_flocking) */
  const float* entity_gravity;
  int32_t first_substep; /* index of the first substep to run (drag is applied on 0) */
  int32_t n_substeps;    /* how many to run; <=0 => desc.substeps - first_substep */
  /* != 0: the reference's broad phase exactly - at every substep a pair is evaluated (for all environments) iff the
   * bounding circles of SOME environment of the batch overlap (World.collides, core.py:2797-2801).  The library picks the
   * form (vmas_world_exact_form): none needed where every pair is sphere-sphere (their force is exactly 0 wherever the
   * circles do not overlap); the LAZY form INSIDE the step launch at any batch size (round 6: tiles step with every pair
   * on, publish the pairs they overlap with fire-and-forget device-scope atomics, and only a tile that has an environment
   * in a pair's band - circles apart, force non-zero - without an overlapping environment of its own reads the batch's
   * words, waiting for the other tiles only if the bit is still clear); under HIP-graph capture a mask launch + a
   * one-substep launch per substep (no fused epilogue then).  Mutually exclusive with pair_mask.
   * 0 = every static pair is evaluated per environment (see pair_mask above). */
  int32_t exact_broad_phase;
  int32_t reserved;
} VmasStepArgs;

/* LIDAR description for World.cast_rays (core.py:1662-1786): one entry per
 * (agent, sensor). */
typedef struct VmasLidarDesc {
  int32_t entity;      /* index of the casting entity */
  int32_t n_rays;
  float max_range;
  int32_t n_targets;      /* filtered entities, any order (min is order-free) */
  const int32_t* targets; /* host pointer, [n_targets] entity indices */
  const float* angles;    /* host pointer, [n_rays] sensor angles (sensors.py:61-70) */
} VmasLidarDesc;

typedef struct VmasWorld VmasWorld;

/* Build the device-side constant block for one world on one GPU.  `desc` and the
 * arrays it points to are host memory and are copied. */
int vmas_world_create(const VmasWorldDesc* desc, int32_t batch, int32_t device_id, VmasWorld** out);
/* Frees device memory (hipFree): must not be called while a stream of this process is being
 * captured into a HIP graph - the host mirror parks handles released during a capture. */
void vmas_world_destroy(VmasWorld* w);

/* Replaces World.step() (core.py:1972-2015) for all `batch` environments. */
int vmas_world_step(VmasWorld* w, float* state, float* agent_ft, int64_t ld,
                    const VmasStepArgs* args /* may be NULL */, void* stream);

/* `n_steps` consecutive World.step() calls enqueued from C with no host round trip in
 * between (rollouts with pre-computed or scripted forces; also what bench.py times):
 * step i reads - and, where clamps apply, rewrites - its agent forces at
 * agent_ft + i * ft_step_stride floats (0 = the same buffer every step). */
int vmas_world_step_n(VmasWorld* w, float* state, float* agent_ft, int64_t ld, int64_t ft_step_stride,
                      int32_t n_steps, const VmasStepArgs* args /* may be NULL */, void* stream);

/* Persistent rollout: the same `n_steps` World.step() calls as vmas_world_step_n, bit for bit,
 * but in ONE launch - the tile of 64 environments stays in LDS between steps, only the agent
 * forces of step i (at agent_ft + i * ft_step_stride) are read from HBM and the state is written
 * back once at the end.  For scripted / pre-computed forces (no policy in the loop). */
int vmas_world_rollout(VmasWorld* w, float* state, float* agent_ft, int64_t ld, int64_t ft_step_stride,
                       int32_t n_steps, const VmasStepArgs* args /* may be NULL */, void* stream);

/* A world-specialised kernel made at RUN TIME for this world: `code_object_path` is a gfx950 code object compiled from
 * csrc/vmas_spec_kernel.h with the tables of the schedule this world runs (the Python side generates and compiles it:
 * vectorizedmultiagentsimulator_amd/specialize.py, cached on disk).  The library loads it, reads the tables back out of
 * the module (`vmas_rt_check`) and accepts it only if they are word for word its own schedule - the rule of the built-in
 * specialisations; launch forms the module does not contain keep the interpreter.  0 ok, -1 refused (vmas_last_error). */
int vmas_world_load_spec(VmasWorld* w, const char* code_object_path);

/* The lane-compacted step kernel (csrc/vmas_compact.h) for worlds whose pairs are all sphere-sphere or line-sphere and
 * that have no joints: broad phase per (environment, pair) with every lane busy, narrow phase over the tile's CONTACTS
 * packed across environments and pairs, results added by the owners in the reference's order (core.py:2176-2199).  Its
 * results are bit for bit those of the other kernels wherever an entity has at most two simultaneous contacts; with three
 * or more they can differ from the interpreter's in the last bit of the summed force (the interpreter adds an entity's
 * contacts segment by segment, this kernel term by term in the reference's pair order; both are within the 1e-5 contract
 * of the reference - tests/test_compact_gpu.py pins each against the oracle on dense-contact states).
 * mode -1 (default): used when the world is dense (>= 64 pairs: football) - and left for the interpreter while the
 * measured contact density says the interpreter is faster, a choice that depends on the states alone (reruns repeat it
 * bit for bit, but a rollout can switch kernels in its course: pin one with 0 / 1 where bitwise equality between runs
 * of DIFFERENT protocols matters); 0: never, 1: whenever the world qualifies.  vmas_world_get_compact: 1 if plain steps
 * of this world run it. */
int vmas_world_set_compact(VmasWorld* w, int32_t mode);
int vmas_world_get_compact(VmasWorld* w);

/* In-kernel grid barriers (exact_broad_phase inside the step launch; the navigation epilogue's collision reduction) need
 * every tile of the launch resident at once.  The library only uses them for grids of at most one tile per CU, but other
 * work on the device (another stream, another process) can still keep a tile from starting: a barrier then gives up after
 * a bounded wait (tens of ms) instead of hanging, the step goes on with the pair bits that had arrived, and a pinned
 * host-visible word is set.  EVERY later vmas_world_step* / vmas_world_rollout* call on the world reads that word first -
 * no synchronisation - and fails (-1, vmas_last_error says why) once, clearing it: a partial mask never passes silently.
 * vmas_world_exact_status synchronises the device and reports the same condition without clearing it: 0 if every barrier
 * so far completed, 1 if one gave up, < 0 on error (tests / debugging). */
int vmas_world_exact_status(VmasWorld* w);
/* The form VmasStepArgs.exact_broad_phase takes for a whole-batch launch of this world outside graph capture: 0 none needed
 * (sphere-sphere pairs only), 1 lazy, inside the launch (any batch size; gated launches and fused epilogues allowed),
 * 2 grid barrier inside the launch (at most one tile per CU; only when pinned by VMAS_EXACT_FORM=barrier), 3 a mask launch +
 * a launch per substep.  < 0 on error. */
int vmas_world_exact_form(VmasWorld* w);

/* Batch-global broad phase of World.collides (core.py:2797-2801): mask[p/32] bit
 * p%32 = any_env(|pos_a - pos_b| <= R_a + R_b).  `mask` is zeroed on the stream
 * first.  Only needed for exact small-batch parity; see DESIGN.md. */
int vmas_world_pair_mask(VmasWorld* w, const float* state, int64_t ld, uint32_t* mask, void* stream);

/* Replaces World.cast_rays for a set of sensors (core.py:1662-1786).  Registers
 * the sensor set once ... */
int vmas_world_set_lidars(VmasWorld* w, const VmasLidarDesc* lidars, int32_t n_lidars);
/* ... then out[(l * max_rays + r) * ld + env] = measured distance; max_rays is the
 * largest n_rays of the registered set. */
int vmas_world_cast_rays(VmasWorld* w, const float* state, int64_t ld, float* out, void* stream);
/* Sensor sets whose targets are all spheres are cast lane-compacted (only the (environment, sensor, target) triples within
 * reach are evaluated; same bits).  -1 (default): the library's choice, 0: the plain kernel, 1: whenever the set qualifies.
 * vmas_world_get_lidar_compact: 1 if the next vmas_world_cast_rays takes the compacted form. */
int vmas_world_set_lidar_compact(VmasWorld* w, int32_t mode);
int vmas_world_get_lidar_compact(VmasWorld* w);

/* Scenario-side geometric queries (World.get_distance core.py:1822-1905,
 * World.is_overlapping core.py:1907-1969) evaluated for a registered list of entity pairs in
 * ONE launch: out[q * ld + env] = distance, or 1.0f / 0.0f for an overlap query. */
#define VMAS_QUERY_DISTANCE 0
#define VMAS_QUERY_OVERLAP 1
typedef struct VmasQuery {
  int32_t kind; /* VMAS_QUERY_* */
  int32_t a, b; /* entity indices, any order (as the caller would pass them) */
} VmasQuery;
int vmas_world_set_queries(VmasWorld* w, const VmasQuery* queries, int32_t n_queries);
int vmas_world_run_queries(VmasWorld* w, const float* state, int64_t ld, float* out, void* stream);

/* Kernel geometry knob: waves cooperating on one 64-environment tile ("lanes per environment", 1..16, 1..8 for
 * worlds with box-box pairs); 0 = the library's choice: together with which pairs are evaluated once into shared LDS
 * rows it maximises the waves running per CU for this world and batch (`select_config`). */
int vmas_world_set_lanes_per_env(VmasWorld* w, int32_t lanes);
int vmas_world_get_lanes_per_env(const VmasWorld* w);

/* vmas_world_step_n over several HIP queues.  Environments are independent, so a sequence of steps can be enqueued as
 * independent launch sequences over parts of the batch (cut at 64-environment tile boundaries): one on the caller's
 * stream, the others on library-owned side streams that are forked from the caller's stream (event) at the start of the
 * call and joined back into it at the end, so the call stays stream-ordered for the caller.  The ~2.9 us launch gap
 * between two dependent kernels of one part is filled by the kernels of the others; results are bit for bit those of
 * one queue.  queues: 0 = the library's choice (two when each half keeps at least one tile per CU, a launch is long
 * against the host's enqueue cost and n_steps >= 8),
 * 1..4 = that many.  Only launches without optional per-call inputs (args == NULL) are split.
 * vmas_world_get_queues returns how many queues a vmas_world_step_n of `n_steps` steps would use. */
int vmas_world_set_queues(VmasWorld* w, int32_t queues);
int vmas_world_get_queues(const VmasWorld* w, int32_t n_steps);

/* World-specialised kernels.  For the worlds and geometries the library was built with tables for
 * (csrc/vmas_spec_gen.h: balance n_agents=4 at two tiles per CU, BASELINE config 2), a plain vmas_world_step / _step_n
 * launch (one-substep world, no optional inputs) runs a form of the step kernel whose schedule - segments, items,
 * owners - is compile-time constants instead of records interpreted from LDS; it is used only when the schedule planned
 * at run time is word for word the generated one, and its results are bit for bit the interpreter's.
 * vmas_world_set_specialized(w, 0) keeps the interpreter (A/B measurements); _get_ tells which one a launch would run. */
int vmas_world_set_specialized(VmasWorld* w, int32_t on);
int vmas_world_get_specialized(VmasWorld* w);

/* Algorithmic HBM bytes one vmas_world_step moves per environment
 * (24*E read + 12*A read + 24*E_dyn written). */
int64_t vmas_world_step_bytes_per_env(const VmasWorld* w);

const char* vmas_last_error(void);
int vmas_abi_version(void);
/* A digest of the sources the loaded library was built from (part of the key of the run-time specialisations' on-disk
 * cache: a code object compiled for another build of the library is never handed to this one). */
const char* vmas_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* VMAS_HIP_H */
