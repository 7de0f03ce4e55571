"""ctypes binding of libxwb.so (include/xwb.h).

There is no fallback: if the HIP library is missing it is built with hipcc, and
if that fails, or no gfx950 device is usable at create time, an exception is
raised.  Nothing here (or anywhere in xworld_amd/) imports the test oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libxwb.so")
if os.environ.get("XWB_LIB_AB"):                 # lab: another build of the same ABI, for A/B runs on ONE box (box-to-box spread is larger than most effects)
    LIB_PATH = os.path.abspath(os.environ["XWB_LIB_AB"])

XWB_ABI_VERSION = 5
XWB_SIMPLE_GAME, XWB_SIMPLE_RACE, XWB_XWORLD2D = 0, 1, 2
XWB_MAP_NAV, XWB_MAP_WALLS = 0, 1
XWB_TASKMODE_LANG_ACQ, XWB_TASKMODE_ONE_CHANNEL = 0, 1
ALIVE, MAX_STEP, DEAD, SUCCESS, LOST_LIFE = 0, 1, 2, 4, 8
XWB_QUEUE_SYNC_AUTO, XWB_QUEUE_SYNC_EVENTS, XWB_QUEUE_SYNC_EPOCHS = 0, 1, 2
DEBUG_FLAGS = {"no_pregen": 1, "no_lazy": 2, "ego_no_cache": 4, "ego_no_span": 8, "ego_no_flat": 16, "no_fused": 32}
STEP_PATHS = ["none", "classic", "lazy", "pregen", "ego_span", "ego_per_env", "lazy_fused"]
SYNC_REASONS = ["probe_ok", "config", "env", "tool", "probe_failed", "probe_error", "not_used", "not_probed"]


class XwbConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("game", C.c_int32), ("num_envs", C.c_int32), ("device", C.c_int32),
        ("env_gid0", C.c_uint32), ("seed", C.c_uint32), ("policy_seed", C.c_uint32),
        ("context", C.c_int32), ("max_steps", C.c_int32),
        ("array_size", C.c_int32),
        ("track_type", C.c_int32), ("race_full_manouver", C.c_int32), ("random", C.c_int32),
        ("difficulty_hard", C.c_int32),
        ("track_width", C.c_double), ("track_length", C.c_double), ("track_radius", C.c_double),
        ("reward_scale", C.c_double),
        ("map_kind", C.c_int32), ("max_dim", C.c_int32), ("dim", C.c_int32), ("num_goals", C.c_int32),
        ("num_blocks", C.c_int32), ("max_steps_factor", C.c_int32), ("task_mode", C.c_int32),
        ("n_tasks", C.c_int32), ("tasks", C.c_int32 * 8),
        ("color", C.c_int32), ("visible_radius", C.c_int32), ("obs_format", C.c_int32), ("n_icons", C.c_int32),
        ("icons64", C.c_void_p), ("icon_type", C.c_void_p), ("icon_name", C.c_void_p), ("icon_colored", C.c_void_p),
        ("curriculum", C.c_double), ("start_level", C.c_int32),
        ("task_schedule", C.c_int32), ("task_weights", C.c_double * 8), ("no_wall_shadow", C.c_int32),
        ("rng_mode", C.c_int32), ("simulator_seed", C.c_int32), ("thread_base", C.c_int32),
        ("n_tasks2", C.c_int32), ("tasks2", C.c_int32 * 8), ("task_schedule2", C.c_int32), ("task_weights2", C.c_double * 8),
        ("task_groups_exclusive", C.c_int32),
        ("task_group_weight", C.c_double), ("task_group_weight2", C.c_double),
        ("queue_sync", C.c_int32),
        ("debug_flags", C.c_int32), ("debug_ego_per", C.c_int32), ("debug_ego_pad", C.c_int32), ("debug_render_shape", C.c_int32),
    ]


class XwbEnvState(C.Structure):
    _fields_ = [
        ("reward", C.c_float), ("game_over", C.c_int32), ("lives", C.c_int32), ("num_steps", C.c_int64),
        ("last_action", C.c_int32), ("last_action_success", C.c_int32),
        ("sg_pos", C.c_int32),
        ("race_x", C.c_float), ("race_y", C.c_float), ("race_angle", C.c_float),
        ("xw_agent_x", C.c_int32), ("xw_agent_y", C.c_int32), ("xw_event", C.c_int32), ("xw_stage", C.c_int32),
        ("xw_target_name", C.c_int32), ("xw_steps_in_task", C.c_int32),
        ("episode", C.c_uint32), ("xw_task", C.c_int32), ("xw_target", C.c_int32), ("xw_agent_dir", C.c_int32), ("xw_level", C.c_int32), ("xw_check_counter", C.c_int32), ("xw_sentence_names", C.c_uint32),
        ("xw_task2", C.c_int32), ("xw_stage2", C.c_int32), ("xw_event2", C.c_int32), ("xw_target2", C.c_int32),
        ("xw_steps_in_task2", C.c_int32),
        ("xw_group_first", C.c_int32), ("xw_group_ran", C.c_int32),
    ]


class XwbTaskPerformance(C.Structure):
    _fields_ = [("successes", C.c_int64), ("failures", C.c_int64), ("success_steps", C.c_int64), ("time_ups", C.c_int64)]


TASK_CLASSES = ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection",
                "XWorld3DNavTargetAvoid", "XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]

# every symbol include/xwb.h declares: (name, restype, argtypes)
_vp = C.c_void_p
_SIGS = [
    ("xwb_default_config", C.c_int, [C.c_int32, C.POINTER(XwbConfig)]),
    ("xwb_create", C.c_int, [C.POINTER(XwbConfig), C.POINTER(_vp)]),
    ("xwb_destroy", C.c_int, [_vp]),
    ("xwb_reset", C.c_int, [_vp, _vp]),
    ("xwb_reset_done", C.c_int, [_vp, _vp]),
    ("xwb_reset_masked", C.c_int, [_vp, _vp, _vp]),
    ("xwb_reset_env", C.c_int, [_vp, C.c_int32, _vp]),
    ("xwb_step", C.c_int, [_vp, _vp, C.c_int32, _vp]),
    ("xwb_step_host", C.c_int, [_vp, _vp, C.c_int32, _vp]),
    ("xwb_step_n", C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    ("xwb_step_autoreset", C.c_int, [_vp, _vp, C.c_int32, _vp]),
    ("xwb_check_errors", C.c_int, [_vp, _vp, C.POINTER(C.c_int32)]),
    ("xwb_queue_sync_mode", C.c_int, [_vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("xwb_queue_sync_forget", C.c_int, [_vp, _vp]),
    ("xwb_step_path", C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("xwb_obs_dev", C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    ("xwb_bind_results", C.c_int, [_vp, _vp]),
    ("xwb_bind_results_ring", C.c_int, [_vp, _vp, C.c_int64]),
    ("xwb_bind_obs", C.c_int, [_vp, _vp]),
    ("xwb_reward_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_game_over_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_get_obs", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("xwb_get_reward", C.c_int, [_vp, _vp, _vp]),
    ("xwb_get_done", C.c_int, [_vp, _vp, _vp]),
    ("xwb_actions_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_num_steps_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_success_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_episode_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_xw_grid_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_minstd_state_dev", C.c_int, [_vp, C.POINTER(_vp)]),
    ("xwb_ego_render_path", C.c_int, [_vp, C.POINTER(C.c_int32)]),
    ("xwb_set_names", C.c_int, [_vp, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int32]),
    ("xwb_sentence", C.c_int, [_vp, C.c_int32, _vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_language_sentence", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_uint32, C.c_uint32, C.c_int32,
                                        C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_language_sentence_2d", C.c_int, [C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_minstd_seed_thread", C.c_uint32, [C.c_int32, C.c_int32]),
    ("xwb_minstd_rand_ind", C.c_int32, [C.POINTER(C.c_uint32), C.c_int32]),
    ("xwb_minstd_rand_range", C.c_float, [C.POINTER(C.c_uint32), C.c_float]),
    ("xwb_done_count", C.c_int, [_vp, _vp, C.POINTER(C.c_int32)]),
    ("xwb_get_num_actions", C.c_int, [_vp, C.POINTER(C.c_int32)]),
    ("xwb_get_screen_out_dimensions", C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("xwb_get_world_dimensions", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("xwb_num_envs", C.c_int, [_vp, C.POINTER(C.c_int32)]),
    ("xwb_get_env_state", C.c_int, [_vp, C.c_int32, _vp, C.POINTER(XwbEnvState)]),
    ("xwb_get_env_obs", C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_size_t]),
    ("xwb_get_env_grid", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("xwb_xw_load_map", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("xwb_get_extra_info", C.c_int, [_vp, C.c_int32, _vp, C.c_char_p, C.c_size_t]),
    ("xwb_state_bytes", C.c_int, [_vp, C.c_int32, C.POINTER(C.c_size_t)]),
    ("xwb_save_state", C.c_int, [_vp, C.c_int32, _vp, C.c_size_t]),
    ("xwb_load_state", C.c_int, [_vp, _vp, C.c_size_t]),
    ("xwb_xw_set_agent_dir", C.c_int, [_vp, C.c_int32, C.c_int32]),
    ("xwb_xw_set_goal_pose", C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    ("xwb_xw_refresh_obs", C.c_int, [_vp, C.c_int32]),
    ("xwb_xw_load_map_task", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("xwb_race_set_car", C.c_int, [_vp, C.c_int32, C.c_float, C.c_float, C.c_float]),
    ("xwb_get_state_packet", C.c_int, [_vp, C.c_int32, C.c_float, _vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_get_task_performance", C.c_int, [_vp, _vp, C.POINTER(XwbTaskPerformance), C.POINTER(C.c_int64)]),
    ("xwb_task_performance_report", C.c_int, [_vp, _vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_decode_game_over_code", C.c_int, [C.c_int32, C.c_char_p, C.c_size_t]),
    ("xwb_xw_get_tile_table", C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("xwb_comm_version", C.c_int, [C.POINTER(C.c_int32)]),
    ("xwb_comm_unique_id", C.c_int, [_vp]),
    ("xwb_comm_init_rank", C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    ("xwb_comm_adopt", C.c_int, [_vp, C.c_int32, C.POINTER(_vp)]),
    ("xwb_comm_destroy", C.c_int, [_vp]),
    ("xwb_comm_info", C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("xwb_comm_group_start", C.c_int, [_vp]),
    ("xwb_comm_group_end", C.c_int, [_vp]),
    ("xwb_gather_results", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, _vp]),
    ("xwb_gather_screens_begin", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, _vp]),
    ("xwb_gather_screens_end", C.c_int, [_vp, _vp]),
    ("xwb_comm_release_sim", C.c_int, [_vp, _vp]),
    ("xwb_gather_results_beside", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, _vp, C.POINTER(C.c_int32)]),
    ("xwb_gather_grids_begin", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, _vp]),
    ("xwb_comm_mark", C.c_int, [_vp, C.c_int32]),
    ("xwb_comm_wait", C.c_int, [_vp, C.c_int32, _vp]),
    ("xwb_xw_pack_grids", C.c_int, [_vp, _vp, _vp, _vp]),
    ("xwb_xw_set_draw", C.c_int, [_vp, C.c_int32]),
    ("xwb_xw_render_grids", C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, _vp]),
    ("xwb_run", C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    ("xwb_last_error", C.c_char_p, []),
    ("xwb_version", C.c_char_p, []),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGS]
# include/xwb_testing.h: test and measurement hooks, outside the drop-in boundary (version node XWB_TESTING)
_TESTING_SIGS = [
    ("xwb_debug_stall_handoff", C.c_int, [_vp, _vp, C.c_int64]),
    ("xwb_profile_begin", C.c_int, [_vp]),
    ("xwb_profile_end", C.c_int, [_vp, _vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("xwb_profile_stop", C.c_int, [_vp]),
]
TESTING_SYMBOLS = [s[0] for s in _TESTING_SIGS]

_lib = None


class XwbError(RuntimeError):
    pass


def load(build_if_missing=True):
    """dlopen libxwb.so (building it in-tree with hipcc first if absent). Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; load it first so that libxwb.so binds to the HIP runtime
    # already in the process (two HIP runtimes in one process cannot both own the device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise XwbError("libxwb.so is missing (run `python -m xworld_amd.build`); there is no CPU fallback")
        from . import build as _build
        _build.build()
    L = C.CDLL(LIB_PATH)
    for name, res, args in _SIGS + _TESTING_SIGS:
        f = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise XwbError("xwb error %d: %s" % (rc, load().xwb_last_error().decode()))


def decode_game_over_code(code):
    buf = C.create_string_buffer(64)
    check(load().xwb_decode_game_over_code(int(code), buf, 64))
    return buf.value.decode()
