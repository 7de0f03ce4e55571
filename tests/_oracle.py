"""ctypes bindings for oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (xworld_amd/) never imports it.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# XWB_ORACLE_LIB: another build of the same sources (tools/sanitize.sh: oracle/_asan/liboracle.so, `make -C oracle asan`)
LIB_PATH = os.environ.get("XWB_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


class MinStd(C.Structure):
    _fields_ = [("x", C.c_uint32)]


class Stream(C.Structure):
    _fields_ = [("key", C.c_uint32 * 2), ("ctr", C.c_uint32 * 4), ("buf", C.c_uint32 * 4), ("have", C.c_int)]


class RaceCfg(C.Structure):
    _fields_ = [("track_type", C.c_int), ("track_width", C.c_double), ("track_length", C.c_double),
                ("track_radius", C.c_double), ("race_full_manouver", C.c_int), ("random", C.c_int),
                ("difficulty_hard", C.c_int), ("reward_scale", C.c_double), ("max_steps", C.c_int),
                ("context", C.c_int), ("simulator_seed", C.c_int), ("nth_thread", C.c_int)]


class IconInfo(C.Structure):
    _fields_ = [("type", C.c_int), ("name_id", C.c_int), ("colored", C.c_int)]


class XwCfg(C.Structure):
    _fields_ = [("map_kind", C.c_int), ("max_dim", C.c_int), ("dim", C.c_int), ("num_goals", C.c_int),
                ("num_blocks", C.c_int), ("max_steps", C.c_int), ("max_steps_factor", C.c_int),
                ("task_mode", C.c_int), ("color", C.c_int), ("context", C.c_int), ("seed", C.c_uint32),
                ("visible_radius", C.c_int), ("n_tasks", C.c_int), ("tasks", C.c_int * 8),
                ("curriculum", C.c_double), ("start_level", C.c_int),
                ("task_schedule", C.c_int), ("task_weights", C.c_double * 8), ("no_wall_shadow", C.c_int),
                ("simulator_seed", C.c_int), ("thread_base", C.c_int),
                ("n_tasks2", C.c_int), ("tasks2", C.c_int * 8), ("task_schedule2", C.c_int), ("task_weights2", C.c_double * 8),
                ("task_groups_exclusive", C.c_int), ("group_weight", C.c_double * 2)]


class Entity(C.Structure):
    _fields_ = [("type", C.c_int), ("x", C.c_int), ("y", C.c_int), ("icon", C.c_int),
                ("name_id", C.c_int), ("serial", C.c_int)]


class PacketField(C.Structure):
    _fields_ = [("key", C.c_char_p),
                ("has_reals", C.c_int), ("reals", f32p), ("n_reals", C.c_size_t),
                ("has_pixels", C.c_int), ("pixels", u8p), ("n_pixels", C.c_size_t),
                ("has_id", C.c_int), ("id", i32p), ("n_id", C.c_size_t),
                ("has_str", C.c_int), ("str", C.c_char_p)]


class RolloutStats(C.Structure):
    _fields_ = [("reward_sum", C.c_double), ("resets", C.c_uint64), ("task_perf", (C.c_int64 * 4) * 9)]


class RolloutOut(C.Structure):
    _fields_ = [("rewards", f32p), ("codes", u8p), ("obs_ck", C.POINTER(C.c_uint64))]


_lib = None


def build():
    # one builder at a time: several test processes starting together must not load a half-linked library
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()                      # `make` is a no-op when liboracle.so is up to date
    except Exception:
        if not os.path.exists(LIB_PATH):
            raise
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_set_trig_libm", None, C.c_int)
    sig("orc_get_trig_libm", C.c_int)
    sig("orc_trig_cos", C.c_double, C.c_double)
    sig("orc_trig_sin", C.c_double, C.c_double)
    sig("orc_xwb_sincos", None, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("orc_minstd_seed", None, C.POINTER(MinStd), C.c_uint64)
    sig("orc_minstd_next", C.c_uint32, C.POINTER(MinStd))
    sig("orc_minstd_rand_ind", C.c_int, C.POINTER(MinStd), C.c_int)
    sig("orc_minstd_rand_range", C.c_float, C.POINTER(MinStd), C.c_float)
    sig("orc_std_hash_string", C.c_uint64, C.c_char_p, C.c_size_t)
    sig("orc_minstd_seed_thread", None, C.POINTER(MinStd), C.c_int, C.c_int)
    sig("orc_philox4x32", None, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
    sig("orc_stream_init", None, C.POINTER(Stream), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
    sig("orc_stream_u32", C.c_uint32, C.POINTER(Stream))
    sig("orc_stream_below", C.c_uint32, C.POINTER(Stream), C.c_uint32)
    sig("orc_stream_unit", C.c_float, C.POINTER(Stream))
    sig("orc_policy_action", C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int)

    sig("orc_sg_create", vp, C.c_int, C.c_int, C.c_int)
    sig("orc_sg_destroy", None, vp)
    sig("orc_sg_reset_game", None, vp)
    sig("orc_sg_take_actions", C.c_float, vp, C.c_int, C.c_int)
    sig("orc_sg_game_over", C.c_int, vp)
    sig("orc_sg_get_lives", C.c_int, vp)
    sig("orc_sg_num_steps", C.c_int64, vp)
    sig("orc_sg_pos", C.c_int, vp)
    sig("orc_sg_get_screen", None, vp, u8p)
    sig("orc_sg_get_state_screen", None, vp, u8p)

    sig("orc_race_default_cfg", None, C.POINTER(RaceCfg))
    sig("orc_race_create", vp, C.POINTER(RaceCfg))
    sig("orc_race_destroy", None, vp)
    sig("orc_race_reset_game", None, vp)
    sig("orc_race_reset_game_with", None, vp, C.c_float, C.c_float, C.c_float, C.c_float)
    sig("orc_race_take_actions", C.c_float, vp, C.c_int, C.c_int)
    sig("orc_race_game_over", C.c_int, vp)
    sig("orc_race_get_lives", C.c_int, vp)
    sig("orc_race_num_actions", C.c_int, vp)
    sig("orc_race_num_steps", C.c_int64, vp)
    sig("orc_race_get_car", None, vp, f32p, f32p, f32p)
    sig("orc_race_set_car", None, vp, C.c_float, C.c_float, C.c_float)
    sig("orc_race_get_screen", None, vp, f32p)
    sig("orc_race_get_state_screen", None, vp, f32p)

    sig("orc_xw_create", vp, C.POINTER(XwCfg), C.c_int, C.POINTER(IconInfo), u8p)
    sig("orc_xw_destroy", None, vp)
    sig("orc_xw_reset_game", None, vp, C.c_uint32, C.c_uint32)
    sig("orc_xw_load_map", None, vp, C.c_int, C.POINTER(Entity), C.c_int, C.c_int, C.c_uint32, C.c_uint32)
    sig("orc_xw_load_map_ex", None, vp, C.c_int, C.POINTER(Entity), C.c_int, i32p, C.c_int, C.c_uint32, C.c_uint32)
    sig("orc_xw_task_kind", C.c_int, vp)
    sig("orc_xw_group_state", None, vp, C.c_int, *([C.POINTER(C.c_int)] * 6))
    sig("orc_xw_group_first", C.c_int, vp)
    sig("orc_xw_between_cell", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_set_pose", None, vp, C.c_int, C.c_double, C.c_double, C.c_double)
    sig("orc_xw_get_pose", None, vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("orc_xw_agent_yaw", C.c_double, vp)
    sig("orc_xw_sentence_names", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_curriculum_state", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_curriculum_configure", C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_record_result", None, vp, C.c_int, C.c_int)
    sig("orc_xw_direction_target", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_num_actions", C.c_int, vp)
    sig("orc_xw_stage_poses", None, vp, C.POINTER(C.c_double), C.c_int)
    sig("orc_xw_agent_masking", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), u8p)
    sig("orc_xw_refresh_screen", None, vp)
    sig("orc_xw_agent_view", None, vp, u8p)
    sig("orc_xw_entity_image", None, vp, C.c_int, u8p)
    sig("orc_cv_get_rotation_matrix_2d", None, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double))
    sig("orc_cv_warp_affine_8uc3", None, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int, C.POINTER(C.c_double), u8p)
    sig("orc_xw_load_map_forced", None, vp, C.c_int, C.POINTER(Entity), C.c_int, i32p, C.c_int, C.c_uint32, C.c_uint32)
    sig("orc_xw_forced_left", C.c_int, vp)
    sig("orc_xw_target2d", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_get_target_cells", None, vp, u8p)
    sig("orc_xw_take_actions", C.c_float, vp, C.c_int, C.c_int)
    for n in ("game_over", "get_lives", "num_actions", "last_action_success", "event", "stage",
              "target_name", "steps_in_task", "n_entities"):
        sig("orc_xw_" + n, C.c_int, vp)
    sig("orc_xw_num_steps", C.c_int64, vp)
    sig("orc_xw_get_entities", None, vp, C.POINTER(Entity))
    sig("orc_xw_agent_xy", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_get_grid", None, vp, i32p)
    sig("orc_xw_screen_dims", None, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int))
    sig("orc_xw_get_screen", None, vp, u8p)
    sig("orc_xw_get_state_screen", None, vp, u8p)
    sig("orc_maze_generate", None, C.POINTER(Stream), C.c_int, C.c_char_p)
    sig("orc_bfs_reachable", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p)
    sig("orc_cv_resize_linear_8u", None, u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int)
    sig("orc_cv_bgr2gray_8u", None, u8p, C.c_int, u8p)
    sig("orc_packet_encode", C.c_size_t, C.POINTER(PacketField), C.c_int, u8p, C.c_size_t)
    sig("orc_packet_decode", C.c_int, u8p, C.c_size_t, C.POINTER(PacketField), C.c_int)
    sig("orc_decode_game_over_code", C.c_int, C.c_int, C.c_char_p, C.c_int)
    sig("orc_obs_checksum", C.c_uint64, C.c_void_p, C.c_size_t)
    sig("orc_sg_rollout", C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
        C.POINTER(RolloutStats), C.POINTER(RolloutOut))
    sig("orc_race_rollout", C.c_uint64, C.c_int, C.POINTER(RaceCfg), C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
        C.POINTER(RolloutStats), C.POINTER(RolloutOut))
    sig("orc_xw_rollout", C.c_uint64, C.c_int, C.POINTER(XwCfg), C.c_int, C.POINTER(IconInfo), u8p,
        C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(RolloutStats), C.POINTER(RolloutOut))
    _lib = L
    return L


def ptr(a, typ):
    return a.ctypes.data_as(typ)


# ----------------------------------------------------------------- wrappers --
class SimpleGame:
    def __init__(self, array_size, max_steps=0, context=1):
        self.L = lib()
        self.n = array_size
        self.context = context
        self.h = self.L.orc_sg_create(array_size, max_steps, context)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_sg_destroy(self.h)
            self.h = None

    def reset_game(self):
        self.L.orc_sg_reset_game(self.h)

    def take_actions(self, a, act_rep=1):
        return self.L.orc_sg_take_actions(self.h, int(a), act_rep)

    def game_over(self):
        return self.L.orc_sg_game_over(self.h)

    def get_lives(self):
        return self.L.orc_sg_get_lives(self.h)

    def num_steps(self):
        return self.L.orc_sg_num_steps(self.h)

    def pos(self):
        return self.L.orc_sg_pos(self.h)

    def screen(self):
        out = np.zeros(self.n, np.uint8)
        self.L.orc_sg_get_screen(self.h, ptr(out, u8p))
        return out

    def state_screen(self):
        out = np.zeros(self.n * self.context, np.uint8)
        self.L.orc_sg_get_state_screen(self.h, ptr(out, u8p))
        return out


def race_cfg(**kw):
    c = RaceCfg()
    lib().orc_race_default_cfg(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class SimpleRace:
    def __init__(self, **kw):
        self.L = lib()
        self.cfg = race_cfg(**kw)
        self.h = self.L.orc_race_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_race_destroy(self.h)
            self.h = None

    def reset_game(self):
        self.L.orc_race_reset_game(self.h)

    def reset_game_with(self, u0, u1, u2, u3):
        self.L.orc_race_reset_game_with(self.h, u0, u1, u2, u3)

    def take_actions(self, a, act_rep=1):
        return self.L.orc_race_take_actions(self.h, int(a), act_rep)

    def game_over(self):
        return self.L.orc_race_game_over(self.h)

    def num_actions(self):
        return self.L.orc_race_num_actions(self.h)

    def num_steps(self):
        return self.L.orc_race_num_steps(self.h)

    def car(self):
        x, y, a = C.c_float(), C.c_float(), C.c_float()
        self.L.orc_race_get_car(self.h, C.byref(x), C.byref(y), C.byref(a))
        return np.array([x.value, y.value, a.value], np.float32)

    def set_car(self, x, y, a):
        self.L.orc_race_set_car(self.h, x, y, a)

    def screen(self):
        out = np.zeros(4, np.float32)
        self.L.orc_race_get_screen(self.h, ptr(out, f32p))
        return out

    def state_screen(self):
        out = np.zeros(4 * self.cfg.context, np.float32)
        self.L.orc_race_get_state_screen(self.h, ptr(out, f32p))
        return out


TYPE_ID = {"goal": 0, "block": 1, "agent": 2}


class Palette:
    """The icon subset a map class can place (xworld_env.py:236-255 set_goal_subtrees)."""

    def __init__(self, subtrees, max_icons_per_subtree=None):
        assets = os.path.join(ROOT, "xworld_amd", "assets")
        with open(os.path.join(assets, "icons.json")) as f:
            meta = json.load(f)
        icons = np.load(os.path.join(assets, "icons64.npz"))["icons"]
        keep = []
        per = {}
        for i, m in enumerate(meta):
            if m["type"] == "goal":
                if m["subtree"] not in subtrees:
                    continue
                if max_icons_per_subtree is not None:
                    # keep whole names only
                    names = per.setdefault(m["subtree"], [])
                    if m["name"] not in names:
                        if len(names) >= max_icons_per_subtree:
                            continue
                        names.append(m["name"])
            keep.append(i)
        self.meta = [meta[i] for i in keep]
        self.icons64 = np.ascontiguousarray(icons[keep])
        self.names = {}
        for t in TYPE_ID:
            self.names[t] = sorted({m["name"] for m in self.meta if m["type"] == t})
        self.info = (IconInfo * len(self.meta))()
        self.type_arr = np.zeros(len(self.meta), np.int32)
        self.name_arr = np.zeros(len(self.meta), np.int32)
        for i, m in enumerate(self.meta):
            self.info[i].type = TYPE_ID[m["type"]]
            self.info[i].name_id = self.names[m["type"]].index(m["name"])
            self.info[i].colored = int(m.get("color", "na") != "na")
            self.type_arr[i] = self.info[i].type
            self.name_arr[i] = self.info[i].name_id

    def __len__(self):
        return len(self.meta)


NAV_SUBTREES = ("animal", "fruit", "furniture", "vegetable")    # XWorldNav.py:17
WALLS_SUBTREES = ("animal", "fruit", "shape")                   # XWorldWalls.py:16


TASK_ID = {"XWorld3DNavTarget": 0, "XWorld3DNavTargetNear": 1, "XWorld3DNavTargetBetween": 2,
           "XWorld3DNavTargetDirection": 3, "XWorld3DNavTargetAvoid": 4,
           "XWorldNavTarget": 5, "XWorldNavNear": 6, "XWorldNavColorTarget": 7, "XWorldNavBetween": 8}


def xw_cfg(**kw):
    c = XwCfg(map_kind=0, max_dim=8, dim=8, num_goals=4, num_blocks=16, max_steps=0,
              max_steps_factor=10, task_mode=0, color=0, context=1, seed=0xC0FFEE, visible_radius=0)
    tasks = kw.pop("tasks", None)
    weights = kw.pop("task_weights", None)
    tasks2 = kw.pop("tasks2", None)
    weights2 = kw.pop("task_weights2", None)
    gw = kw.pop("group_weights", None)                   # the conf's per-group "weight" keys (exclusive scheduling)
    if gw is not None:
        c.group_weight[0], c.group_weight[1] = float(gw[0]), float(gw[1])
    if tasks2 is not None:                               # a second task group, after the first in conf order
        c.n_tasks2 = len(tasks2)
        for i, t in enumerate(tasks2):
            c.tasks2[i] = TASK_ID.get(t, t)
    if weights2 is not None:
        c.task_schedule2 = 1
        for i, x in enumerate(weights2):
            c.task_weights2[i] = float(x)
    if weights is not None:
        c.task_schedule = 1
        for i, x in enumerate(weights):
            c.task_weights[i] = float(x)
    for k, v in kw.items():
        setattr(c, k, v)
    if tasks is not None:
        c.n_tasks = len(tasks)
        for i, t in enumerate(tasks):
            c.tasks[i] = TASK_ID.get(t, t)
    return c


class XWorld:
    def __init__(self, palette, render=True, **kw):
        self.L = lib()
        self.cfg = xw_cfg(**kw)
        self.pal = palette
        self.render = render
        self.h = self.L.orc_xw_create(C.byref(self.cfg), len(palette), palette.info,
                                      ptr(palette.icons64, u8p) if render else None)
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_xw_screen_dims(self.h, C.byref(h), C.byref(w), C.byref(c))
        self.dims = (h.value, w.value, c.value)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_xw_destroy(self.h)
            self.h = None

    def reset_game(self, env_gid, episode):
        self.L.orc_xw_reset_game(self.h, env_gid, episode)

    def load_map(self, ents, dim, target_pick=-1, env_gid=0, episode=0):
        arr = (Entity * len(ents))()
        for i, e in enumerate(ents):
            arr[i] = Entity(*e)
        self.L.orc_xw_load_map(self.h, len(ents), arr, dim, target_pick, env_gid, episode)

    def load_map_ex(self, ents, dim, decisions, env_gid=0, episode=0):
        arr = (Entity * len(ents))()
        for i, e in enumerate(ents):
            arr[i] = Entity(*e)
        d = np.asarray(decisions, np.int32)
        self.L.orc_xw_load_map_ex(self.h, len(ents), arr, dim, ptr(d, i32p), len(d), env_gid, episode)

    def load_map_forced(self, ents, dim, decisions, env_gid=0, episode=0):
        """As load_map_ex, but the decisions stay installed for idle stages that run at step time."""
        arr = (Entity * len(ents))()
        for i, e in enumerate(ents):
            arr[i] = Entity(*e)
        self._forced = np.asarray(decisions, np.int32).copy()          # must outlive the episode
        self.L.orc_xw_load_map_forced(self.h, len(ents), arr, dim, ptr(self._forced, i32p), len(self._forced),
                                      env_gid, episode)

    def forced_left(self):
        return self.L.orc_xw_forced_left(self.h)

    def target2d(self):
        x, y = C.c_int(), C.c_int()
        self.L.orc_xw_target2d(self.h, C.byref(x), C.byref(y))
        return x.value, y.value

    def set_pose(self, ent, yaw, scale=1.0, offset=0.0):
        self.L.orc_xw_set_pose(self.h, int(ent), float(yaw), float(scale), float(offset))

    def get_pose(self, ent):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.orc_xw_get_pose(self.h, int(ent), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def stage_poses(self, poses):
        """[[yaw, scale, offset], ...] in entity order, applied by the next load_map* before the idle stage."""
        self._poses = np.ascontiguousarray(poses, np.float64)
        self.L.orc_xw_stage_poses(self.h, self._poses.ctypes.data_as(C.POINTER(C.c_double)), len(self._poses))

    def num_actions(self):
        return self.L.orc_xw_num_actions(self.h)

    def curriculum_state(self):
        """(level, check counter) -- XWorldEnv.current_level, curriculum_check_counter"""
        a, b = C.c_int(), C.c_int()
        self.L.orc_xw_curriculum_state(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def curriculum_configure(self):
        """the level logic of XWorldNav._configure alone: (level, dim, num_goals, num_blocks)"""
        d, g, b = C.c_int(), C.c_int(), C.c_int()
        lv = self.L.orc_xw_curriculum_configure(self.h, C.byref(d), C.byref(g), C.byref(b))
        return lv, d.value, g.value, b.value

    def record_result(self, kind, result):
        self.L.orc_xw_record_result(self.h, TASK_ID.get(kind, kind), int(result))

    def sentence_names(self):
        a, b = C.c_int(), C.c_int()
        self.L.orc_xw_sentence_names(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def direction_target(self):
        x, y, wd = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_xw_direction_target(self.h, C.byref(x), C.byref(y), C.byref(wd))
        return x.value, y.value, wd.value

    def agent_yaw(self):
        return self.L.orc_xw_agent_yaw(self.h)

    def agent_masking(self):
        r = self.cfg.visible_radius
        x, y = C.c_int(), C.c_int()
        sh = np.zeros(r * r, np.uint8)
        self.L.orc_xw_agent_masking(self.h, C.byref(x), C.byref(y), ptr(sh, u8p))
        return x.value, y.value, sh.reshape(r, r)

    def refresh_screen(self):
        self.L.orc_xw_refresh_screen(self.h)

    def agent_view(self):
        """XMap::to_image for the agent: [r*64, r*64, 3] B,G,R before XWorldSimulator's resizes"""
        s = self.cfg.visible_radius * 64
        out = np.zeros((s, s, 3), np.uint8)
        self.L.orc_xw_agent_view(self.h, ptr(out, u8p))
        return out

    def entity_image(self, ent):
        """XItem::get_item_image of entity `ent`: [64, 64, 3] B,G,R"""
        out = np.zeros((64, 64, 3), np.uint8)
        self.L.orc_xw_entity_image(self.h, int(ent), ptr(out, u8p))
        return out

    def group_state(self, g):
        """(task kind, stage, steps in task, the event its task recorded in the last call, 2-D target x, y) of task group g"""
        v = [C.c_int() for _ in range(6)]
        self.L.orc_xw_group_state(self.h, g, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def group_first(self):
        """exclusive scheduling: conf index of the group that heads the teacher's list after the last sort"""
        return self.L.orc_xw_group_first(self.h)

    def task_kind(self):
        return self.L.orc_xw_task_kind(self.h)

    def between_cell(self):
        x, y = C.c_int(), C.c_int()
        self.L.orc_xw_between_cell(self.h, C.byref(x), C.byref(y))
        return x.value, y.value

    def target_cells(self):
        d = self.cfg.max_dim
        out = np.zeros(d * d, np.uint8)
        self.L.orc_xw_get_target_cells(self.h, ptr(out, u8p))
        return out.reshape(d, d)

    def take_actions(self, a, act_rep=1):
        return self.L.orc_xw_take_actions(self.h, int(a), act_rep)

    def game_over(self):
        return self.L.orc_xw_game_over(self.h)

    def get_lives(self):
        return self.L.orc_xw_get_lives(self.h)

    def num_steps(self):
        return self.L.orc_xw_num_steps(self.h)

    def last_action_success(self):
        return self.L.orc_xw_last_action_success(self.h)

    def event(self):
        return self.L.orc_xw_event(self.h)

    def stage(self):
        return self.L.orc_xw_stage(self.h)

    def target_name(self):
        return self.L.orc_xw_target_name(self.h)

    def steps_in_task(self):
        return self.L.orc_xw_steps_in_task(self.h)

    def entities(self):
        n = self.L.orc_xw_n_entities(self.h)
        arr = (Entity * n)()
        self.L.orc_xw_get_entities(self.h, arr)
        return [(e.type, e.x, e.y, e.icon, e.name_id, e.serial) for e in arr]

    def agent_xy(self):
        x, y = C.c_int(), C.c_int()
        self.L.orc_xw_agent_xy(self.h, C.byref(x), C.byref(y))
        return x.value, y.value

    def grid(self):
        d = self.cfg.max_dim
        out = np.zeros(d * d, np.int32)
        self.L.orc_xw_get_grid(self.h, ptr(out, i32p))
        return out.reshape(d, d)

    def screen(self):
        h, w, c = self.dims
        out = np.zeros(h * w * c, np.uint8)
        self.L.orc_xw_get_screen(self.h, ptr(out, u8p))
        return out.reshape(c, h, w)

    def state_screen(self):
        h, w, c = self.dims
        out = np.zeros(h * w * c * self.cfg.context, np.uint8)
        self.L.orc_xw_get_state_screen(self.h, ptr(out, u8p))
        return out.reshape(self.cfg.context * c, h, w)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return list(o)


def policy_action(policy_seed, env_gid, step, num_actions):
    return lib().orc_policy_action(policy_seed, env_gid, step, num_actions)


def decode_game_over_code(code):
    buf = C.create_string_buffer(64)
    lib().orc_decode_game_over_code(code, buf, 64)
    return buf.value.decode()


class Rollout:
    """Outputs of an oracle batch rollout: arrays shaped [steps, n_envs]."""

    def __init__(self, n_envs, steps, want_obs=True):
        self.rewards = np.zeros((steps, n_envs), np.float32)
        self.codes = np.zeros((steps, n_envs), np.uint8)
        self.obs_ck = np.zeros((steps, n_envs), np.uint64) if want_obs else None
        self.stats = RolloutStats()
        self.out = RolloutOut(ptr(self.rewards, f32p), ptr(self.codes, u8p),
                              ptr(self.obs_ck, C.POINTER(C.c_uint64)) if want_obs else None)


def sg_rollout(n_envs, array_size, steps, policy_seed, env_gid0=0, context=1):
    r = Rollout(n_envs, steps)
    lib().orc_sg_rollout(n_envs, array_size, context, steps, policy_seed, env_gid0, C.byref(r.stats), C.byref(r.out))
    return r


def race_rollout(n_envs, cfg, seed, steps, policy_seed, env_gid0=0):
    r = Rollout(n_envs, steps)
    lib().orc_race_rollout(n_envs, C.byref(cfg), seed, steps, policy_seed, env_gid0, C.byref(r.stats), C.byref(r.out))
    return r


def xw_rollout(n_envs, cfg, palette, steps, policy_seed, env_gid0=0, render=False):
    r = Rollout(n_envs, steps, want_obs=render)
    lib().orc_xw_rollout(n_envs, C.byref(cfg), len(palette), palette.info, ptr(palette.icons64, u8p), steps,
                         policy_seed, env_gid0, 1 if render else 0, C.byref(r.stats), C.byref(r.out))
    return r


def obs_checksum_np(obs2d):
    """numpy version of orc_obs_checksum for rows of a [n, bytes] uint8 view."""
    n = obs2d.shape[1]
    w = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    return (obs2d.astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64)
