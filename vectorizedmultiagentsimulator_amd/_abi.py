"""ctypes mirror of ``include/vmas_hip.h`` and the loader of ``libvmas_hip.so``.

The library is the product: there is NO fallback.  If it has not been built (see
``__graft_entry__.build()`` / ``csrc/build.sh``) importing the loader raises
``VmasHipLibraryMissing`` with the build command - loudly, never silently.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

ABI_VERSION = 3


def leading_dim(batch: int) -> int:
    """``ld`` of the packed buffers for ``batch`` environments: the batch rounded up to whole 64-environment tiles, plus
    ``LD_PAD_TILES`` tiles where that is a large power of two (see the constant)."""
    ld = (int(batch) + 63) // 64 * 64
    pad = int(os.environ.get("VMAS_LD_PAD", LD_PAD_TILES))
    if pad and ld >= int(os.environ.get("VMAS_LD_FROM", LD_PAD_FROM)) and (ld & (ld - 1)) == 0:
        ld += 64 * pad
    return ld


#: A tile reads its 6 E rows at a stride of ld * 4 bytes: with ld a large power of two (262 144 environments: 1 MiB, 1 M: 4 MiB)
#: every row of a tile falls on the same HBM channel.  Measured (profiles/r05*_ld_pad*.jsonl): see DESIGN.md section 6 "Round 5".
LD_PAD_TILES = 0
LD_PAD_FROM = 1 << 16
STATE_FIELDS = 6
AGENT_FIELDS = 3

SHAPE_SPHERE, SHAPE_BOX, SHAPE_LINE = 0, 1, 2

F_MOVABLE = 1 << 0
F_ROTATABLE = 1 << 1
F_AGENT = 1 << 2
F_HOLLOW = 1 << 3
F_MAX_SPEED = 1 << 4
F_V_RANGE = 1 << 5
F_LIN_FRICTION = 1 << 6
F_ANG_FRICTION = 1 << 7
F_GRAVITY = 1 << 8
F_MAX_F = 1 << 9
F_F_RANGE = 1 << 10
F_MAX_T = 1 << 11
F_T_RANGE = 1 << 12

PAIR_SS, PAIR_LS, PAIR_LL, PAIR_BS, PAIR_BL, PAIR_BB = range(6)


class EntityDesc(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("shape", C.c_int32),
        ("agent_index", C.c_int32),
        ("mass", C.c_float),
        ("inertia", C.c_float),
        ("length", C.c_float),
        ("width", C.c_float),
        ("radius", C.c_float),
        ("bound_radius", C.c_float),
        ("one_minus_drag", C.c_float),
        ("max_speed", C.c_float),
        ("v_range", C.c_float),
        ("lin_friction", C.c_float),
        ("ang_friction", C.c_float),
        ("gravity", C.c_float * 2),
        ("max_f", C.c_float),
        ("f_range", C.c_float),
        ("max_t", C.c_float),
        ("t_range", C.c_float),
    ]


class PairDesc(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("type", C.c_int32), ("bound_sum", C.c_float)]


class JointDesc(C.Structure):
    _fields_ = [
        ("a", C.c_int32),
        ("b", C.c_int32),
        ("delta_a", C.c_float * 2),
        ("delta_b", C.c_float * 2),
        ("dist", C.c_float),
        ("rotate", C.c_int32),
        ("fixed_rotation", C.c_float),
    ]


class WorldDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_entities", C.c_int32),
        ("n_agents", C.c_int32),
        ("n_pairs", C.c_int32),
        ("n_joints", C.c_int32),
        ("substeps", C.c_int32),
        ("sub_dt", C.c_float),
        ("gravity", C.c_float * 2),
        ("has_gravity", C.c_int32),
        ("x_semidim", C.c_float),
        ("y_semidim", C.c_float),
        ("collision_force", C.c_float),
        ("joint_force", C.c_float),
        ("contact_margin", C.c_float),
        ("torque_constraint_force", C.c_float),
        ("entities", C.POINTER(EntityDesc)),
        ("pairs", C.POINTER(PairDesc)),
        ("joints", C.POINTER(JointDesc)),
    ]


class StepArgs(C.Structure):
    _fields_ = [
        ("pair_mask", C.c_void_p),
        ("joint_fixed_rot", C.c_void_p),
        ("entity_gravity", C.c_void_p),
        ("first_substep", C.c_int32),
        ("n_substeps", C.c_int32),
        ("exact_broad_phase", C.c_int32),
        ("reserved", C.c_int32),
    ]


class LidarDesc(C.Structure):
    _fields_ = [
        ("entity", C.c_int32),
        ("n_rays", C.c_int32),
        ("max_range", C.c_float),
        ("n_targets", C.c_int32),
        ("targets", C.POINTER(C.c_int32)),
        ("angles", C.POINTER(C.c_float)),
    ]


class Query(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32)]


QUERY_DISTANCE, QUERY_OVERLAP = 0, 1


class VmasHipLibraryMissing(ImportError):
    pass


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# VMAS_HIP_LIB: another build of the SAME sources (profiling knobs compiled in, an experimental variant) for A/B
# measurements; it must live beside the product library and export the same ABI version.
LIB_PATH = os.path.join(_PKG_DIR, "csrc", os.path.basename(os.environ.get("VMAS_HIP_LIB", "libvmas_hip.so")))

#: every symbol include/vmas_hip.h declares (checked by tests/test_abi_symbols.py)

# ---------------------------------------------------------------- include/vmas_env_hip.h
ENV_MAX_AGENTS = 32
ENV_MAX_PACKAGES = 8
ACTION_ERR_NAN, ACTION_ERR_OUT_OF_RANGE = 1, 2
POST_BALANCE, POST_TRANSPORT, POST_NAVIGATION, POST_FOOTBALL = 1, 2, 3, 4


class ActionSlot(C.Structure):
    _fields_ = [
        ("action", C.c_void_p),
        ("u_out", C.c_void_p),
        ("action_size", C.c_int32),
        ("agent_index", C.c_int32),
        ("u_range", C.c_float * 3),
        ("u_multiplier", C.c_float * 3),
        ("action_index", C.c_void_p),
        ("nvec", C.c_int32 * 3),
    ]


ENV_MAX_SCRIPTS = 4
SCRIPT_FOOTBALL_BALL = 1


class AgentScript(C.Structure):
    _fields_ = [("kind", C.c_int32), ("agent_index", C.c_int32), ("entity", C.c_int32), ("u_out", C.c_void_p),
                ("params", C.c_float * 8)]


class IngestArgs(C.Structure):
    _fields_ = [("n_agents", C.c_int32), ("clamp", C.c_int32), ("agents", ActionSlot * ENV_MAX_AGENTS),
                ("n_scripts", C.c_int32), ("scripts", AgentScript * ENV_MAX_SCRIPTS)]


RESET_MAX_OPS, RESET_MAX_TERMS = 48, 36
SPAWN_UNIFORM, SPAWN_OFFSET, SPAWN_FIXED = 1, 2, 3
TERM_DIST, TERM_DIST_POINT, TERM_MIN_DIST = 0, 1, 2


class SpawnOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("entity", C.c_int32), ("base", C.c_int32), ("avoid_from", C.c_int32),
                ("x_lo", C.c_float), ("x_hi", C.c_float), ("y_lo", C.c_float), ("y_hi", C.c_float), ("min_dist", C.c_float),
                ("has_rot", C.c_int32), ("rot", C.c_float)]


class ResetTerm(C.Structure):
    _fields_ = [("a", C.c_int32), ("b", C.c_int32), ("factor", C.c_float), ("out", C.c_void_p),
                ("kind", C.c_int32), ("n", C.c_int32), ("px", C.c_float), ("py", C.c_float)]


class ResetArgs(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("n_terms", C.c_int32), ("n_flags", C.c_int32),
                ("ops", SpawnOp * RESET_MAX_OPS), ("terms", ResetTerm * RESET_MAX_TERMS), ("flags", C.c_void_p * 8),
                ("steps", C.c_void_p), ("episode", C.c_void_p), ("seed", C.c_uint64), ("gave_up", C.c_void_p)]


class StepLimit(C.Structure):
    _fields_ = [("steps", C.c_void_p), ("max_steps", C.c_float)]


class BalanceDesc(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32),
        ("goal", C.c_int32), ("package", C.c_int32), ("line", C.c_int32), ("floor", C.c_int32), ("agent0", C.c_int32),
        ("goal_radius", C.c_float), ("package_radius", C.c_float), ("line_length", C.c_float),
        ("floor_length", C.c_float), ("floor_width", C.c_float),
        ("shaping_factor", C.c_float), ("fall_reward", C.c_float),
    ]


class BalanceBuffers(C.Structure):
    _fields_ = [
        ("global_shaping", C.c_void_p), ("obs", C.c_void_p), ("rew", C.c_void_p), ("pos_rew", C.c_void_p),
        ("ground_rew", C.c_void_p), ("on_the_ground", C.c_void_p), ("done", C.c_void_p), ("limit", StepLimit),
    ]


class TransportDesc(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32), ("n_packages", C.c_int32),
        ("goal", C.c_int32), ("package0", C.c_int32), ("agent0", C.c_int32),
        ("goal_radius", C.c_float), ("package_length", C.c_float), ("package_width", C.c_float),
        ("shaping_factor", C.c_float),
    ]


class TransportBuffers(C.Structure):
    _fields_ = [
        ("global_shaping", C.c_void_p), ("on_goal", C.c_void_p), ("obs", C.c_void_p), ("rew", C.c_void_p),
        ("done", C.c_void_p), ("limit", StepLimit),
    ]


class FootballDesc(C.Structure):
    _fields_ = [
        ("n_blue", C.c_int32), ("n_red", C.c_int32), ("agent0", C.c_int32),
        ("observe_teammates", C.c_int32), ("observe_adversaries", C.c_int32), ("dense_reward", C.c_int32),
        ("goal_x", C.c_float), ("goal_half", C.c_float), ("touch_dist", C.c_float),
        ("pos_shaping_factor_ball_goal", C.c_float), ("pos_shaping_factor_agent_ball", C.c_float),
        ("distance_to_ball_trigger", C.c_float), ("scoring_reward", C.c_float),
    ]


class FootballBuffers(C.Structure):
    _fields_ = [
        ("pos_shaping", C.c_void_p), ("obs", C.c_void_p), ("rew", C.c_void_p), ("terms", C.c_void_p),
        ("touching", C.c_void_p), ("done", C.c_void_p), ("agent_ft", C.c_void_p), ("limit", StepLimit),
    ]


class NavigationDesc(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32), ("agent0", C.c_int32), ("goal_of", C.c_int32 * ENV_MAX_AGENTS),
        ("shared_rew", C.c_int32), ("collisions", C.c_int32), ("observe_all_goals", C.c_int32), ("n_rays", C.c_int32),
        ("agent_radius", C.c_float), ("goal_radius", C.c_float),
        ("pos_shaping_factor", C.c_float), ("final_reward", C.c_float), ("agent_collision_penalty", C.c_float),
        ("min_collision_distance", C.c_float), ("lidar_range", C.c_float),
    ]


class NavigationBuffers(C.Structure):
    _fields_ = [
        ("pos_shaping", C.c_void_p), ("obs", C.c_void_p), ("rew", C.c_void_p), ("agent_pos_rew", C.c_void_p),
        ("pos_rew", C.c_void_p), ("final_rew", C.c_void_p), ("collision_rew", C.c_void_p), ("done", C.c_void_p),
        ("lidar", C.c_void_p), ("lidar_max_rays", C.c_int64), ("pair_any", C.c_void_p), ("pair_index", C.c_void_p),
        ("limit", StepLimit),
    ]


EXPORTED_SYMBOLS = (
    "vmas_world_create",
    "vmas_world_destroy",
    "vmas_world_step",
    "vmas_world_step_n",
    "vmas_world_rollout",
    "vmas_world_pair_mask",
    "vmas_world_set_lidars",
    "vmas_world_cast_rays",
    "vmas_world_set_lidar_compact",
    "vmas_world_get_lidar_compact",
    "vmas_world_set_queries",
    "vmas_world_run_queries",
    "vmas_world_set_lanes_per_env",
    "vmas_world_get_lanes_per_env",
    "vmas_debug_math",  # include/vmas_debug_hip.h: test / profiling hooks, never called by the product path
    "vmas_debug_trace",
    "vmas_debug_schedule",
    "vmas_debug_force_gave_up",
    "vmas_debug_football_form",
    "vmas_debug_compact_plan",
    "vmas_debug_compact_stats",
    "vmas_debug_lazy_stats",
    "vmas_world_exact_status",
    "vmas_world_exact_form",
    "vmas_world_load_spec",
    "vmas_world_set_compact",
    "vmas_world_get_compact",
    "vmas_world_set_specialized",
    "vmas_world_get_specialized",
    "vmas_world_set_queues",
    "vmas_world_get_queues",
    "vmas_world_step_bytes_per_env",
    "vmas_last_error",
    "vmas_abi_version",
    "vmas_build_id",
    # include/vmas_env_hip.h
    "vmas_env_ingest_actions",
    "vmas_host_word_create",
    "vmas_host_word_destroy",
    "vmas_env_validate_actions",
    "vmas_env_validate_begin",
    "vmas_env_validate_end",
    "vmas_host_word_gate",
    "vmas_world_step_env_gated",
    "vmas_world_gated_refused",
    "vmas_balance_post_step",
    "vmas_transport_post_step",
    "vmas_navigation_post_step",
    "vmas_football_post_step",
    "vmas_env_reset_where",
    "vmas_world_step_env",
    "vmas_world_rollout_env",
    "vmas_world_reserve_epilogue",
    "vmas_world_step_env_check",
)

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen ``libvmas_hip.so`` and declare the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VmasHipLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or "
            "vectorizedmultiagentsimulator_amd/csrc/build.sh) - there is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.vmas_world_create.argtypes = [C.POINTER(WorldDesc), i32, i32, C.POINTER(vp)]
    lib.vmas_world_create.restype = C.c_int
    lib.vmas_world_destroy.argtypes = [vp]
    lib.vmas_world_destroy.restype = None
    lib.vmas_world_step.argtypes = [vp, vp, vp, i64, C.POINTER(StepArgs), vp]
    lib.vmas_world_step.restype = C.c_int
    lib.vmas_world_step_n.argtypes = [vp, vp, vp, i64, i64, i32, C.POINTER(StepArgs), vp]
    lib.vmas_world_step_n.restype = C.c_int
    lib.vmas_world_rollout.argtypes = [vp, vp, vp, i64, i64, i32, C.POINTER(StepArgs), vp]
    lib.vmas_world_rollout.restype = C.c_int
    lib.vmas_world_pair_mask.argtypes = [vp, vp, i64, vp, vp]
    lib.vmas_world_pair_mask.restype = C.c_int
    lib.vmas_world_set_lidars.argtypes = [vp, C.POINTER(LidarDesc), i32]
    lib.vmas_world_set_lidars.restype = C.c_int
    lib.vmas_world_cast_rays.argtypes = [vp, vp, i64, vp, vp]
    lib.vmas_world_cast_rays.restype = C.c_int
    lib.vmas_world_set_queries.argtypes = [vp, C.POINTER(Query), i32]
    lib.vmas_world_set_queries.restype = C.c_int
    lib.vmas_world_run_queries.argtypes = [vp, vp, i64, vp, vp]
    lib.vmas_world_run_queries.restype = C.c_int
    lib.vmas_world_set_lanes_per_env.argtypes = [vp, i32]
    lib.vmas_world_set_lanes_per_env.restype = C.c_int
    lib.vmas_world_get_lanes_per_env.argtypes = [vp]
    lib.vmas_world_get_lanes_per_env.restype = C.c_int
    lib.vmas_world_set_specialized.argtypes = [vp, i32]
    lib.vmas_world_set_specialized.restype = C.c_int
    lib.vmas_world_get_specialized.argtypes = [vp]
    lib.vmas_world_get_specialized.restype = C.c_int
    lib.vmas_world_load_spec.argtypes = [vp, C.c_char_p]
    lib.vmas_world_load_spec.restype = C.c_int
    lib.vmas_world_set_compact.argtypes = [vp, i32]
    lib.vmas_world_set_compact.restype = C.c_int
    lib.vmas_world_get_compact.argtypes = [vp]
    lib.vmas_world_get_compact.restype = C.c_int
    lib.vmas_world_exact_status.argtypes = [vp]
    lib.vmas_world_exact_status.restype = C.c_int
    if hasattr(lib, "vmas_world_exact_form"):  # (an older build beside this one, VMAS_HIP_LIB=...: A/B measurements only)
        lib.vmas_world_exact_form.argtypes = [vp]
        lib.vmas_world_exact_form.restype = C.c_int
    lib.vmas_world_set_queues.argtypes = [vp, i32]
    lib.vmas_world_set_queues.restype = C.c_int
    lib.vmas_world_get_queues.argtypes = [vp, i32]
    lib.vmas_world_get_queues.restype = C.c_int
    lib.vmas_world_step_bytes_per_env.argtypes = [vp]
    lib.vmas_world_step_bytes_per_env.restype = i64
    lib.vmas_env_ingest_actions.argtypes = [C.POINTER(IngestArgs), i32, vp, vp, i64, vp, vp]
    lib.vmas_env_ingest_actions.restype = C.c_int
    lib.vmas_host_word_create.argtypes = [i32, C.POINTER(vp), C.POINTER(vp)]
    lib.vmas_host_word_create.restype = C.c_int
    lib.vmas_host_word_destroy.argtypes = [vp]
    lib.vmas_host_word_destroy.restype = None
    lib.vmas_env_validate_actions.argtypes = [C.POINTER(IngestArgs), i32, vp, vp, i64, vp, vp, vp]
    lib.vmas_env_validate_actions.restype = C.c_int
    lib.vmas_env_validate_begin.argtypes = [C.POINTER(IngestArgs), i32, vp, vp, i64, vp, vp, vp]
    lib.vmas_env_validate_begin.restype = C.c_int
    lib.vmas_env_validate_end.argtypes = [vp, i32, vp]
    lib.vmas_env_validate_end.restype = C.c_int
    lib.vmas_host_word_gate.argtypes = [vp]
    lib.vmas_host_word_gate.restype = vp
    for fn, d, b in ((lib.vmas_balance_post_step, BalanceDesc, BalanceBuffers),
                     (lib.vmas_transport_post_step, TransportDesc, TransportBuffers),
                     (lib.vmas_navigation_post_step, NavigationDesc, NavigationBuffers),
                     (lib.vmas_football_post_step, FootballDesc, FootballBuffers)):
        fn.argtypes = [C.POINTER(d), C.POINTER(b), i32, vp, i64, vp]
        fn.restype = C.c_int
    lib.vmas_world_step_env.argtypes = [vp, vp, vp, i64, C.POINTER(StepArgs), C.POINTER(IngestArgs), vp, i32, vp, vp, vp]
    lib.vmas_world_step_env.restype = C.c_int
    lib.vmas_world_step_env_gated.argtypes = lib.vmas_world_step_env.argtypes
    lib.vmas_world_step_env_gated.restype = C.c_int
    if hasattr(lib, "vmas_world_gated_refused"):  # (an A/B against a library of an earlier round: scripts/gpu_run.sh lazy-cost)
        lib.vmas_world_gated_refused.argtypes = [vp]
        lib.vmas_world_gated_refused.restype = C.c_int
    lib.vmas_env_reset_where.argtypes = [C.POINTER(ResetArgs), i32, i32, i32, vp, vp, vp, i64, vp]
    lib.vmas_env_reset_where.restype = C.c_int
    lib.vmas_world_rollout_env.argtypes = [vp, vp, vp, i64, C.POINTER(StepArgs), C.POINTER(IngestArgs), vp, i32, vp, vp, i32, vp]
    lib.vmas_world_rollout_env.restype = C.c_int
    lib.vmas_world_reserve_epilogue.argtypes = [vp, i32, i32]
    lib.vmas_world_reserve_epilogue.restype = C.c_int
    lib.vmas_world_step_env_check.argtypes = [vp, i32, vp]
    lib.vmas_world_step_env_check.restype = C.c_int
    lib.vmas_last_error.argtypes = []
    lib.vmas_last_error.restype = C.c_char_p
    lib.vmas_abi_version.argtypes = []
    lib.vmas_abi_version.restype = C.c_int
    lib.vmas_build_id.argtypes = []
    lib.vmas_build_id.restype = C.c_char_p
    if lib.vmas_abi_version() != ABI_VERSION:
        raise VmasHipLibraryMissing(
            f"{LIB_PATH} has ABI version {lib.vmas_abi_version()}, expected {ABI_VERSION}: rebuild it"
        )
    _lib = lib
    return lib


def last_error() -> str:
    return (load_library().vmas_last_error() or b"").decode()
