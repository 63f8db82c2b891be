"""ctypes binding of libeggsim.so (include/eggsim.h).

This is the Python twin of the LuaJIT `ffi.cdef` block in
lua/egg_fluid_simulation/simulation_handler.lua: every call the Lua wrapper makes
goes through the same C entry point here.  There is no fallback: if the shared
library is missing or no HIP device is usable, construction raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EGGSIM_LIB lets a developer load a diagnostic build (e.g. libeggsim_prof.so); default is the product library
LIB_PATH = os.environ.get("EGGSIM_LIB") or os.path.join(_HERE, "libeggsim.so")

EGG_OK = 0
EGG_WARN_UNKNOWN_ID = 1
EGG_WARN_FEW_PARTICLES = 2
EGG_ERR_UNKNOWN_ID = -1
EGG_ERR_INVALID_ARGUMENT = -2
EGG_ERR_NO_DEVICE = -3
EGG_ERR_DEVICE = -4
EGG_ERR_UNSUPPORTED = -5
EGG_ERR_INTERNAL = -6

WHITE, YOLK = 0, 1
DEFAULT_COUNT = -1  # EGG_DEFAULT_COUNT: the caller gave nil for a particle count

FIELDS = ["x", "y", "vx", "vy", "last_x", "last_y", "radius", "inv_mass", "mass_t", "batch_id"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}

OPT_CLAIM_MARGIN_CELLS = 0
OPT_TILE_TARGET_PARTICLES = 1
OPT_TIMING = 2
OPT_FORCE_SINGLE_TILE = 3
OPT_THREADS_PER_PARTICLE = 4
OPT_SPIN_SLEEP = 5
OPT_BUDGET_PARTICLES_WHITE = 6
OPT_BUDGET_PARTICLES_YOLK = 7
OPT_FORCE_GLOBAL_STATE = 8
OPT_FUSE_TYPES = 9
OPT_PACKED = 10
OPT_GROUP_PARTICLES = 11
OPT_LEVEL_WALK = 12
PK_VARIANT_LEVELS_INORDER, PK_VARIANT_LEVELS_OOO, PK_VARIANT_EXEC, PK_VARIANT_EXEC_CHAIN, PK_VARIANT_SORT_LDS, PK_VARIANT_SORT_DIRECT = 1, 2, 4, 8, 16, 32
PK_VARIANT_PASS_FUSED = 64

CONFIG_FIELDS = ["damping", "follow_strength", "cohesion_strength",
                 "cohesion_interaction_distance_factor", "collision_strength",
                 "collision_overlap_factor", "min_mass", "max_mass", "min_radius", "max_radius",
                 "max_collision_fraction", "mass_distribution_variance", "eps"]


class EggConfig(C.Structure):
    _fields_ = [(k, C.c_double) for k in CONFIG_FIELDS]


class EggBatchInfo(C.Structure):
    _fields_ = [("key", C.c_int64), ("target_x", C.c_double), ("target_y", C.c_double), ("white_radius", C.c_double),
                ("yolk_radius", C.c_double), ("n_white", C.c_int64), ("n_yolk", C.c_int64)]


ENVIRONMENT_FIELDS = ("min_x", "min_y", "max_x", "max_y", "centroid_x", "centroid_y", "max_radius", "max_velocity",
                      "last_centroid_x", "last_centroid_y")


class EggEnvironment(C.Structure):  # egg_environment
    _fields_ = [(k, C.c_double) for k in ENVIRONMENT_FIELDS]


class EggStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("pair_solves", C.c_int64), ("follow_solves", C.c_int64),
                ("kernel_launches", C.c_int64), ("retiles", C.c_int64), ("redo_steps", C.c_int64),
                ("n_tiles", C.c_int64 * 2), ("max_tile_particles", C.c_int64 * 2),
                ("last_step_kernel_ms", C.c_double), ("single_tile", C.c_int64 * 2),
                ("kernel_ms", C.c_double * 2), ("kernel_ms_sum", C.c_double * 2), ("timed_steps", C.c_int64),
                ("max_pass_visits", C.c_int64 * 2), ("budget", C.c_double * 2), ("fused_launch", C.c_int64),
                ("packed", C.c_int64 * 2), ("pk_kernel_ms", (C.c_double * 10) * 2), ("pk_kernel_launches", (C.c_int64 * 10) * 2),
                ("host_ms", C.c_double * 3), ("max_levels", C.c_int64 * 2), ("pk_variants", C.c_int64 * 2)]


class EggRenderConfig(C.Structure):  # egg_render_config
    _fields_ = [("color", C.c_float * 4), ("outline_color", C.c_float * 4), ("outline_thickness", C.c_double),
                ("highlight_strength", C.c_double), ("shadow_strength", C.c_double), ("texture_scale", C.c_double),
                ("motion_blur", C.c_double)]


class EggRenderParams(C.Structure):  # egg_render_params
    _fields_ = [("screen_w", C.c_int32), ("screen_h", C.c_int32), ("origin_x", C.c_double), ("origin_y", C.c_double),
                ("interpolation_alpha", C.c_double), ("threshold", C.c_double), ("smoothness", C.c_double),
                ("use_instancing", C.c_int32), ("canvas_w", C.c_int32 * 2), ("canvas_h", C.c_int32 * 2),
                ("clear", C.c_float * 4)]


PK_KINDS = ["egg_pk_begin_kernel", "egg_pk_mid_kernel", "egg_pk_lists_fresh_kernel", "egg_pk_lists_stale_kernel",
            "egg_pk_levels_kernel", "egg_pk_sort_kernel", "egg_pk_exec_kernel", "egg_pk_end_kernel", "egg_pk_reduce_kernel",
            "egg_pk_pass_kernel"]


# every symbol include/eggsim.h declares, with its signature
_SIGNATURES = {
    "egg_default_config": (C.c_int, [C.c_int, C.POINTER(EggConfig)]),
    "egg_create": (C.c_int, [C.POINTER(EggConfig), C.POINTER(EggConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "egg_destroy": (None, [C.c_void_p]),
    "egg_last_error": (C.c_char_p, [C.c_void_p]),
    "egg_set_config": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EggConfig)]),
    "egg_get_config": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EggConfig)]),
    "egg_add": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64,
                          C.POINTER(C.c_int64)]),
    "egg_add_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                               C.c_int64, C.c_int64, C.c_void_p]),
    "egg_add_many_keyed": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                     C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "egg_export_batch": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(EggBatchInfo), C.c_void_p, C.c_void_p]),
    "egg_import_batch": (C.c_int, [C.c_void_p, C.POINTER(EggBatchInfo), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "egg_group_create": (C.c_int, [C.POINTER(EggConfig), C.POINTER(EggConfig), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_void_p)]),
    "egg_group_destroy": (None, [C.c_void_p]),
    "egg_group_last_error": (C.c_char_p, [C.c_void_p]),
    "egg_group_n_devices": (C.c_int32, [C.c_void_p]),
    "egg_group_handle": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "egg_group_set_halo": (C.c_int, [C.c_void_p, C.c_double]),
    "egg_group_add": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "egg_group_remove": (C.c_int, [C.c_void_p, C.c_int64]),
    "egg_group_set_target": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double]),
    "egg_group_get_position": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egg_group_update": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "egg_group_step": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "egg_group_owner": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "egg_group_get_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "egg_remove": (C.c_int, [C.c_void_p, C.c_int64]),
    "egg_set_target": (C.c_int, [C.c_void_p, C.c_int64, C.c_double, C.c_double]),
    "egg_set_targets_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egg_get_target": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egg_update": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "egg_step": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "egg_prepare_step": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "egg_step_begin": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32]),
    "egg_step_end": (C.c_int, [C.c_void_p, C.c_int32]),
    "egg_step_peek_visits": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64 * 2), C.POINTER(C.c_double * 2)]),
    "egg_synchronize": (C.c_int, [C.c_void_p]),
    "egg_get_position": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egg_get_positions_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egg_get_bounds_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egg_get_claims_many": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egg_get_n_particles": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "egg_list_ids": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "egg_get_elapsed": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egg_download_particles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "egg_selftest_arith": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]),
    "egg_get_stats": (C.c_int, [C.c_void_p, C.POINTER(EggStats)]),
    "egg_get_environment": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EggEnvironment)]),
    "egg_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "egg_default_render_config": (C.c_int, [C.c_int, C.POINTER(EggRenderConfig)]),
    "egg_set_render_config": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EggRenderConfig)]),
    "egg_get_render_config": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EggRenderConfig)]),
    "egg_set_render_flags": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "egg_set_add_color": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "egg_set_color": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "egg_default_render_params": (C.c_int, [C.POINTER(EggRenderParams)]),
    "egg_render": (C.c_int, [C.c_void_p, C.POINTER(EggRenderParams), C.c_void_p]),
    "egg_render_canvas": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egg_render_particle_texture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
}

EXPORTED_SYMBOLS = sorted(_SIGNATURES)

_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """Load libeggsim.so.  Raises LibraryMissing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                "libeggsim.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C egg_fluid_simulation_amd/csrc` (needs hipcc; the solver has no CPU path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
