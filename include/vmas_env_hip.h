/* vmas_env_hip.h - C ABI of the fused Environment.step() stages around World.step()
 * (SURVEY.md section 8f, rows 1 and 2), exported by the same libvmas_hip.so.
 *
 * The reference runs, per step and per agent, a few dozen small torch ops on either side of
 * World.step(): action ingest (`Environment._set_action`, vmas/simulator/environment/
 * environment.py:616-749, then `Dynamics.process_action`, vmas/simulator/dynamics/
 * holonomic.py:14-15 / holonomic_with_rot.py) and the scenario's reward / observation / done /
 * info (vmas/scenarios/balance.py:218-267, transport.py:131-191, navigation.py:200-285).
 * Each entry point below replaces one of those per-agent Python loops by ONE launch over the
 * packed state of vmas_hip.h (state[E][6][ld], agent_ft[A][3][ld]).
 *
 * Conventions as in vmas_hip.h: plain pointers and sizes, the caller owns every buffer (device
 * memory unless said otherwise), asynchronous on `stream`, 0 = ok / <0 = error with
 * vmas_last_error().  Row-major outputs use the reference's own shapes so that the host can
 * hand out views: obs[a] is a contiguous [batch, obs_dim] matrix, rew[a] a [batch] vector.
 * Bool outputs are one byte per environment (0/1), i.e. a torch.bool tensor.
 */
#ifndef VMAS_ENV_HIP_H
#define VMAS_ENV_HIP_H

#include <stdint.h>

#include "vmas_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VMAS_ENV_MAX_AGENTS 32
#define VMAS_ENV_MAX_PACKAGES 8

/* ---------------------------------------------------------------- action ingest (8f-2) */
typedef struct VmasActionSlot {
  const float* action;   /* [batch, action_size] row-major: what the policy produced */
  float* u_out;          /* [batch, action_size] or NULL: agent.action.u (scaled action) */
  int32_t action_size;   /* 2 = Holonomic (force), 3 = HolonomicWithRotation (force + torque) */
  int32_t agent_index;   /* row block of agent_ft */
  float u_range[3];      /* Agent.action.u_range per dimension (core.py:414-517) */
  float u_multiplier[3]; /* Agent.action.u_multiplier per dimension; a scenario's sign flip (football's red
                            team acts in a mirrored frame, football.py:1050-1057) is folded in as a negative factor */
  const int64_t* action_index; /* discrete actions (Environment(continuous_actions=False), environment.py:657-705):
                                  [batch] flat indices, decoded per dimension with `nvec`; NULL = continuous */
  int32_t nvec[3];             /* Agent.discrete_action_nvec */
} VmasActionSlot;

/* A scripted agent (Agent(action_script=...), core.py:966-982) whose script is known to the library:
 * its action is computed on the device from the world state instead of by a Python callback. */
#define VMAS_ENV_MAX_SCRIPTS 4
#define VMAS_SCRIPT_FOOTBALL_BALL 1 /* football.py:1620-1680: impulse off the pitch border, damped by |vel.y| */
typedef struct VmasAgentScript {
  int32_t kind;        /* VMAS_SCRIPT_* */
  int32_t agent_index; /* row block of agent_ft that receives the force */
  int32_t entity;      /* entity whose state drives the script (the scripted agent itself) */
  float* u_out;        /* [batch, 2] or NULL: agent.action.u */
  float params[8];     /* FOOTBALL_BALL: 2 * agent_size, pitch_width / 2, pitch_length / 2, goal_size / 2
                          (evaluated in double by the host, like the Python expressions of the reference) */
} VmasAgentScript;

typedef struct VmasIngestArgs {
  int32_t n_agents;
  int32_t clamp;         /* Environment(clamp_actions=...) : clamp to +-u_range instead of asserting */
  VmasActionSlot agents[VMAS_ENV_MAX_AGENTS];
  int32_t n_scripts;
  VmasAgentScript scripts[VMAS_ENV_MAX_SCRIPTS];
} VmasIngestArgs;

#define VMAS_ACTION_ERR_NAN 1u          /* environment.py:621 */
#define VMAS_ACTION_ERR_OUT_OF_RANGE 2u /* environment.py:651-653 */

/* Continuous-action branch of Environment._set_action + process_action for every policy agent:
 * agent_ft[agent][0:2] = clamp?(action[:, 0:2]) * u_multiplier, [2] likewise when action_size
 * is 3.  `err_flags` (one uint32, may be NULL) gets VMAS_ACTION_ERR_* OR-ed in where the
 * reference would have raised an AssertionError; the host decides when to look at it. */
int vmas_env_ingest_actions(const VmasIngestArgs* args, int32_t batch, const float* state /* scripts only, else NULL */,
                            float* agent_ft, int64_t ld, uint32_t* err_flags, void* stream);

/* The reference asserts on the host BEFORE it touches the world (environment.py:621,651-653) - two device-to-host
 * round trips per agent there.  Here the flags are one word in pinned host memory the kernels OR into directly:
 * `vmas_host_word_create` makes it (`*host`: the CPU's address, `*dev`: what to pass as `err_flags`), and
 * `vmas_env_validate_actions` = vmas_env_ingest_actions + ONE stream synchronisation + read-and-clear of the word, in one
 * call: returns the VMAS_ACTION_ERR_* flags (>= 0), or < 0 on an error of the call itself (vmas_last_error).  A caller
 * that passes `dev` to vmas_world_step_env instead and looks at `*host` before its NEXT step learns of a bad action one
 * step late, for nothing. */
int vmas_host_word_create(int32_t device_id, uint32_t** host, uint32_t** dev);
void vmas_host_word_destroy(uint32_t* host);
int vmas_env_validate_actions(const VmasIngestArgs* args, int32_t batch, const float* state /* scripts only, else NULL */,
                              float* agent_ft, int64_t ld, uint32_t* err_host, uint32_t* err_dev, void* stream);
/* The same in two halves, so that the step can be LAUNCHED while the validation is still in flight and the host's wait
 * overlaps the step's own execution: `_begin` enqueues the ingest-only launch (flags into the block's GATE word - device
 * memory, `vmas_host_word_gate`) and the marker launch (gate -> host flags, sequence number -> host) and returns the
 * sequence number (> 0); the caller then enqueues `vmas_world_step_env_gated(..., gate, ...)` - a step launch that does
 * NOTHING, not a load, if the gate word is nonzero when it starts - and calls `_end(host, seq, stream)`: polls the marker,
 * returns the flags (and re-opens the gate behind the refused step).  Semantics of the reference's asserts exactly - a
 * refused action never reaches the world - without an idle queue between validation and step. */
int vmas_env_validate_begin(const VmasIngestArgs* args, int32_t batch, const float* state /* scripts only, else NULL */,
                            float* agent_ft, int64_t ld, uint32_t* err_host, uint32_t* err_dev, void* stream);
int vmas_env_validate_end(uint32_t* err_host, int32_t seq, void* stream);
uint32_t* vmas_host_word_gate(uint32_t* host);

/* ---------------------------------------------------------------- step counter / time limit */
typedef struct VmasStepLimit {
  float* steps;     /* [batch] Environment.steps (float, environment.py:107), incremented by 1; may be NULL */
  float max_steps;  /* done |= steps >= max_steps (environment.py:407-412); < 0 = no limit */
} VmasStepLimit;

/* ---------------------------------------------------------------- balance (balance.py:218-267) */
typedef struct VmasBalanceDesc {
  int32_t n_agents;
  int32_t goal, package, line, floor, agent0; /* entity indices; agents are agent0 .. agent0+n-1 */
  float goal_radius, package_radius, line_length, floor_length, floor_width;
  float shaping_factor, fall_reward;
} VmasBalanceDesc;

typedef struct VmasBalanceBuffers {
  float* global_shaping;  /* [batch] in/out: Scenario.global_shaping */
  float* obs;             /* [n_agents][batch][16] */
  float* rew;             /* [n_agents][batch] (the reward is shared: every row equal) */
  float* pos_rew;         /* [batch] info["pos_rew"] */
  float* ground_rew;      /* [batch] info["ground_rew"] */
  uint8_t* on_the_ground; /* [batch] Scenario.on_the_ground */
  uint8_t* done;          /* [batch] */
  VmasStepLimit limit;
} VmasBalanceBuffers;

int vmas_balance_post_step(const VmasBalanceDesc* desc, const VmasBalanceBuffers* buf, int32_t batch,
                           const float* state, int64_t ld, void* stream);

/* ---------------------------------------------------------------- transport (transport.py:131-191) */
typedef struct VmasTransportDesc {
  int32_t n_agents, n_packages;
  int32_t goal, package0, agent0; /* packages are package0 .. package0+n_packages-1 */
  float goal_radius, package_length, package_width;
  float shaping_factor;
} VmasTransportDesc;

typedef struct VmasTransportBuffers {
  float* global_shaping; /* [n_packages][batch] in/out: package.global_shaping */
  uint8_t* on_goal;      /* [n_packages][batch] package.on_goal */
  float* obs;            /* [n_agents][batch][4 + 7 * n_packages] */
  float* rew;            /* [n_agents][batch] (shared) */
  uint8_t* done;         /* [batch] */
  VmasStepLimit limit;
} VmasTransportBuffers;

int vmas_transport_post_step(const VmasTransportDesc* desc, const VmasTransportBuffers* buf, int32_t batch,
                             const float* state, int64_t ld, void* stream);

/* ---------------------------------------------------------------- navigation (navigation.py:200-285) */
typedef struct VmasNavigationDesc {
  int32_t n_agents;
  int32_t agent0;                         /* agents are agent0 .. agent0+n-1 */
  int32_t goal_of[VMAS_ENV_MAX_AGENTS];   /* entity index of agent i's goal */
  int32_t shared_rew, collisions, observe_all_goals;
  int32_t n_rays;                         /* rays of each agent's LIDAR (collisions only) */
  float agent_radius, goal_radius;
  float pos_shaping_factor, final_reward, agent_collision_penalty, min_collision_distance;
  float lidar_range;
} VmasNavigationDesc;

typedef struct VmasNavigationBuffers {
  float* pos_shaping;       /* [n_agents][batch] in/out: agent.pos_shaping */
  float* obs;               /* [n_agents][batch][4 + 2*(observe_all_goals ? n_agents : 1) + n_rays] */
  float* rew;               /* [n_agents][batch] */
  float* agent_pos_rew;     /* [n_agents][batch] agent.pos_rew */
  float* pos_rew;           /* [batch] Scenario.pos_rew (sum over agents) */
  float* final_rew;         /* [batch] */
  float* collision_rew;     /* [n_agents][batch] agent.agent_collision_rew */
  uint8_t* done;            /* [batch] */
  const float* lidar;       /* vmas_world_cast_rays output, sensor l = agent l; NULL without collisions */
  int64_t lidar_max_rays;
  const uint32_t* pair_any; /* vmas_world_pair_mask output: World.collides' batch-global reduction */
  const int32_t* pair_index;/* [n_agents*n_agents] device: static pair index of (i, j), -1 = never collide */
  VmasStepLimit limit;
} VmasNavigationBuffers;

int vmas_navigation_post_step(const VmasNavigationDesc* desc, const VmasNavigationBuffers* buf, int32_t batch,
                              const float* state, int64_t ld, void* stream);

/* ---------------------------------------------------------------- football (football.py:1121-1515)
 * the learning-vs-learning game (no heuristic AI, no shooting): agents = blue 0..n_blue-1, red 0..n_red-1,
 * ball, as consecutive entities from `agent0` and consecutive agent_ft blocks from 0. */
typedef struct VmasFootballDesc {
  int32_t n_blue, n_red;
  int32_t agent0;  /* entity index of blue agent 0; the ball is entity agent0 + n_blue + n_red */
  int32_t observe_teammates, observe_adversaries, dense_reward;
  float goal_x;      /* pitch_length / 2 + ball_size / 2: the goal line for the ball and |x| of both goal points */
  float goal_half;   /* goal_size / 2 */
  float touch_dist;  /* agent_size + ball_size + 1e-2 (info["touching_ball"]) */
  float pos_shaping_factor_ball_goal, pos_shaping_factor_agent_ball, distance_to_ball_trigger, scoring_reward;
} VmasFootballDesc;

typedef struct VmasFootballBuffers {
  float* pos_shaping;       /* [4][batch] in/out: ball.pos_shaping_blue, _red, pos_shaping_agent_blue, _red */
  float* obs;               /* [n_blue + n_red][batch][obs_dim], obs_dim = 16 + 8 * observed others */
  float* rew;               /* [n_blue + n_red][batch] */
  float* terms;             /* [9][batch] out: sparse_reward_blue, pos_rew_blue, pos_rew_red, pos_rew_agent_blue,
                               pos_rew_agent_red, min_agent_dist_to_ball_blue, _red, dist_ball_to_goal_blue, _red */
  uint8_t* touching;        /* [2][batch] out: info["touching_ball"] blue, red */
  uint8_t* done;            /* [batch] */
  const float* agent_ft;    /* the agent forces of the step (agent.state.force is observed) */
  VmasStepLimit limit;
} VmasFootballBuffers;

int vmas_football_post_step(const VmasFootballDesc* desc, const VmasFootballBuffers* buf, int32_t batch,
                            const float* state, int64_t ld, void* stream);

/* ---------------------------------------------------------------- masked reset
 * Environment.reset_at(i) for EVERY environment whose mask byte is set, in one launch and without a host sync: the
 * reference resets one environment per Python call (environment.py:204-252 -> World.reset, core.py:1184-1192, then the
 * scenario's reset_world_at).  For a masked environment the kernel zeroes the whole entity state and the agent forces
 * (World.reset), then runs the scenario's spawn program - the placement law of its reset_world_at restated as a list
 * of operations - on a counter-based generator (Philox4x32-10 keyed by `seed`, counter = environment index, that
 * environment's episode number, operation, try), so a reset draws nothing from a shared stream and unmasked
 * environments are not touched at all (their bits stay).  Finally the scenario's cached terms are re-initialised.
 *
 * UNIFORM restates ScenarioUtils.spawn_entities_randomly / find_random_pos_for_entity (utils.py:241-319): the position is
 * uniform in the box and re-drawn while it is closer than `min_dist` to any entity placed by operations
 * [avoid_from, this one) - per environment the same rejection law as the reference's batch loop. */
#define VMAS_RESET_MAX_OPS 48
#define VMAS_RESET_MAX_TERMS 36
#define VMAS_SPAWN_UNIFORM 1 /* pos ~ U([x_lo,x_hi) x [y_lo,y_hi)), rejected within min_dist of ops [avoid_from, i) */
#define VMAS_SPAWN_OFFSET 2  /* pos = pos(base) + (U([x_lo,x_hi)) , y_lo): x_lo == x_hi = a constant offset, no draw */
#define VMAS_SPAWN_FIXED 3   /* pos = (x_lo, y_lo) */
#define VMAS_SPAWN_TRIES 4096 /* UNIFORM: an infeasible placement ends after this many draws (the reference loops for ever and
                               * warns, utils.py:311-317); every placement that gave up is counted in VmasResetArgs.gave_up */
typedef struct VmasSpawnOp {
  int32_t kind;       /* VMAS_SPAWN_* */
  int32_t entity;     /* entity placed */
  int32_t base;       /* OFFSET: entity it is placed relative to (placed by an earlier operation) */
  int32_t avoid_from; /* UNIFORM: first operation whose entity must be kept at min_dist */
  float x_lo, x_hi, y_lo, y_hi;
  float min_dist;
  int32_t has_rot;    /* != 0: the entity's rotation is set to `rot` as well (football.py:404-409, 686-1020) */
  float rot;
} VmasSpawnOp;
#define VMAS_TERM_DIST 0      /* out[env] = |pos(a) - pos(b)| * factor;   a < 0: out[env] = factor */
#define VMAS_TERM_DIST_POINT 1 /* out[env] = |pos(a) - (px, py)| * factor            (football.py:536-553: ball to goal) */
#define VMAS_TERM_MIN_DIST 2  /* out[env] = min over e in [a, a + n) of |pos(e) - pos(b)| * factor   (football.py:586-600) */
typedef struct VmasResetTerm { /* the scenario's cached terms (shaping ...) re-initialised from the new placement */
  int32_t a, b;
  float factor;
  float* out; /* [batch] */
  int32_t kind, n; /* VMAS_TERM_* */
  float px, py;
} VmasResetTerm;
typedef struct VmasResetArgs {
  int32_t n_ops, n_terms, n_flags;
  VmasSpawnOp ops[VMAS_RESET_MAX_OPS];
  VmasResetTerm terms[VMAS_RESET_MAX_TERMS];
  uint8_t* flags[8];  /* [batch] bool tensors cleared for a reset environment (on_goal, on_the_ground, ...) */
  float* steps;       /* [batch] Environment.steps, zeroed; may be NULL */
  uint32_t* episode;  /* [batch] in/out: resets this environment has had (part of the generator's counter) */
  uint64_t seed;
  uint32_t* gave_up;  /* [1] += placements that hit VMAS_SPAWN_TRIES and kept an overlapping position; may be NULL */
} VmasResetArgs;
int vmas_env_reset_where(const VmasResetArgs* args, int32_t batch, int32_t n_entities, int32_t n_agents,
                         const uint8_t* mask /* [batch] bool */, float* state, float* agent_ft, int64_t ld, void* stream);

/* ---------------------------------------------------------------- the whole step in one launch
 * World.step() with the action ingest as its prologue and one scenario's post-step as its
 * epilogue: the tile of 64 environments a block integrates is still in LDS when the physics is
 * done, so reward / observation / done are computed from it without a second trip through HBM
 * and without a second and third launch.  Same results, bit for bit, as
 *   vmas_env_ingest_actions(ingest) ; vmas_world_step(args) ; vmas_<scenario>_post_step(desc, buffers).
 * `ingest` may be NULL (agent_ft already holds the forces); `post_desc` / `post_buffers` point to the
 * VmasBalance* / VmasTransport* pair selected by `post_kind`.  Navigation's post-step needs the
 * LIDAR and the batch-global collision mask of the NEW state and stays a separate launch. */
#define VMAS_POST_NONE 0 /* prologue only (post_desc / post_buffers NULL): actions -> forces -> World.step() */
#define VMAS_POST_BALANCE 1
#define VMAS_POST_TRANSPORT 2
#define VMAS_POST_NAVIGATION 3 /* VmasNavigationDesc / VmasNavigationBuffers (lidar, lidar_max_rays, pair_any unused: the
                                * epilogue casts the world's registered sensors - sensor a = agent a's - on the tile, and
                                * World.collides' reduction over the batch is made in the launch itself; the collision
                                * penalties are applied behind a grid-wide barrier inside the step kernel while every tile is
                                * resident at once - else by a second small kernel behind it, same stream).
                                * vmas_world_rollout_env: with the barrier form only (at most one tile per CU, no capture). */
#define VMAS_POST_FOOTBALL 4   /* VmasFootballDesc / VmasFootballBuffers (agent_ft unused: the epilogue reads the clamped forces
                                * of its own tile).  Worlds that run the lane-compacted step kernel only
                                * (vmas_world_get_compact() == 1: sphere / line worlds such as football); the agents and the
                                * ball must be the consecutive dynamic entities agent0 .. agent0 + n, agent index = slot.
                                * vmas_world_rollout_env: per-step outputs get a leading step axis ([K][n][batch][obs_dim],
                                * terms [K][9][batch], touching [K][2][batch]); the ball's script runs on the tile. */
int vmas_world_step_env(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args /* may be NULL */,
                        const VmasIngestArgs* ingest /* may be NULL */, uint32_t* err_flags /* may be NULL */,
                        int32_t post_kind, const void* post_desc, const void* post_buffers, void* stream);

/* vmas_world_step_env as a GATED launch: if `*gate` (device memory: vmas_host_word_gate) is nonzero when the launch starts,
 * it does nothing at all - every kernel of the step reads the gate first (the step kernel, navigation's collision kernel,
 * football's post-step kernel).  Single steps; every post_kind; the exact broad phase in its lazy form only
 * (vmas_world_exact_form <= 1: the barrier form's sequence numbers advance on the host - < 0).
 * A caller that learns from vmas_env_validate_end that the gate WAS shut calls `vmas_world_gated_refused(w)` before the
 * world's next launch: it takes back what the host advanced for the launch that arrived nowhere (navigation's barrier
 * number / which of its two masks is next). */
int vmas_world_step_env_gated(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args,
                              const VmasIngestArgs* ingest, uint32_t* gate, int32_t post_kind, const void* post_desc,
                              const void* post_buffers, void* stream);
int vmas_world_gated_refused(VmasWorld* w);

/* K consecutive Environment.step() calls in ONE launch: the same results, bit for bit, as
 * `n_steps` vmas_world_step_env launches without resets in between, but the tile of 64 environments stays in LDS from
 * step to step - per step only the actions are read from HBM and the outputs written; the state goes back once, at the
 * end.  Step k reads rows [k * batch, (k + 1) * batch) of every slot's action tensor (`ingest->agents[i].action`:
 * [n_steps * batch, action_size], or action_index [n_steps * batch]) and writes the k-th slab of every PER-STEP output
 * of `post_buffers`, laid out as [n_steps][the single-step shape]: obs, rew, done (+ balance: pos_rew, ground_rew).
 * Persistent terms (global_shaping, on_goal, on_the_ground, the step counter `limit.steps`, agent_ft, u_out) hold the
 * values of the last step on return.  `err_flags` collects VMAS_ACTION_ERR_* over all steps; a flagged action has been
 * integrated by then - validate beforehand (vmas_env_ingest_actions) where the reference's asserts are wanted.  No
 * library-scripted agents (their scripts read the state in HBM).  post_kind as for vmas_world_step_env. */
int vmas_world_rollout_env(VmasWorld* w, float* state, float* agent_ft, int64_t ld, const VmasStepArgs* args /* may be NULL */,
                           const VmasIngestArgs* ingest, uint32_t* err_flags /* may be NULL */, int32_t post_kind,
                           const void* post_desc, const void* post_buffers, int32_t n_steps, void* stream);

/* Tells the library that this world's steps will be vmas_world_step_env launches with the `post_kind` epilogue
 * (`n_packages`: transport only), whose observation staging needs LDS beside the physics tile: the library's choice of
 * waves per tile and of shared pair rows (vmas_hip.h, "lanes per env") is re-made with that LDS included, instead of
 * falling back at the first launch that does not fit.  Optional; call it before the first step.  The reference has no
 * counterpart (kernel geometry). */
int vmas_world_reserve_epilogue(VmasWorld* w, int32_t post_kind, int32_t n_packages);

/* (post_kind = VMAS_POST_FOOTBALL: 0 if the world runs the lane-compacted kernel and its tile plus the epilogue's
 * observation slabs fit the CU's LDS.)
 * 0 if vmas_world_step_env(post_kind = VMAS_POST_NAVIGATION, post_desc) can run on this world as it is planned now: the
 * world's registered sensors are what the epilogue casts (sensor a on agent a, `n_rays` rays of `lidar_range`, its targets
 * the other agents in order), and the tile plus the epilogue's scratch fit the CU's LDS (shared pair rows are given up
 * for it if that is what it takes).  -1 with vmas_last_error() otherwise: the caller keeps the separate launches
 * (vmas_world_cast_rays, vmas_world_pair_mask, vmas_navigation_post_step). */
int vmas_world_step_env_check(VmasWorld* w, int32_t post_kind, const void* post_desc);

#ifdef __cplusplus
}
#endif
#endif /* VMAS_ENV_HIP_H */
