"""BatchedSimulator: num_envs reference `SimulatorInterface`s as one object on one MI355X.

Host-side mirror of simulator::SimulatorInterface (simulator_interface.h:40-89) over the C ABI
of libxwb.so.  Option names and defaults are those of python/py_simulator.cpp:97-136.
torch is used for device memory views and streams only.
"""
import ctypes as C
import os

import numpy as np

from . import assets, lib
from .lib import XWB_SIMPLE_GAME, XWB_SIMPLE_RACE, XWB_XWORLD2D

GAMES = {"simple_game": XWB_SIMPLE_GAME, "simple_race": XWB_SIMPLE_RACE, "xworld": XWB_XWORLD2D}


class _DevArray:
    """Zero-copy view of library-owned device memory through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _require(opts, key):
    # extract_py_dict_val(required=True), py_simulator.cpp:40-57 -> RuntimeError
    if key not in opts:
        raise RuntimeError("Key '%s' is required but not provided." % key)
    return opts[key]


class BatchedSimulator:
    def __init__(self, name, opts=None, num_envs=1, device=0, env_gid0=0, seed=0xC0FFEE, policy_seed=0x5EED):
        opts = dict(opts or {})
        if name not in GAMES:
            raise RuntimeError("Unrecognized game type: " + name)        # py_simulator.cpp:184-186
        self.L = lib.load()
        self.name = name
        cfg = lib.XwbConfig()
        lib.check(self.L.xwb_default_config(GAMES[name], C.byref(cfg)))
        cfg.num_envs = int(num_envs)
        cfg.device = int(device)
        cfg.env_gid0 = int(env_gid0)
        cfg.seed = int(seed) & 0xFFFFFFFF
        cfg.policy_seed = int(policy_seed) & 0xFFFFFFFF
        cfg.max_steps = int(opts.get("max_steps", 0))
        cfg.context = int(opts.get("context", 1))
        # FLAGS_simulator_seed (simulator_util.cpp:26): rng = "minstd" replays the reference's thread-local engines
        rng = opts.get("rng", "philox")
        if rng not in ("philox", "minstd"):
            raise RuntimeError("rng must be 'philox' or 'minstd'")
        cfg.rng_mode = 1 if rng == "minstd" else 0
        cfg.simulator_seed = int(opts.get("simulator_seed", 0))
        cfg.thread_base = int(opts.get("thread_base", 0))
        # how the batch's two internal queues hand over (include/xwb.h XWB_QUEUE_SYNC_*; not a reference option)
        qs = opts.get("queue_sync", "auto")
        if qs not in ("auto", "events", "epochs"):
            raise RuntimeError("queue_sync must be 'auto', 'events' or 'epochs'")
        cfg.queue_sync = ("auto", "events", "epochs").index(qs)
        # xwb_config "Debug configuration": A/B switches of the library's own paths, e.g. debug=["no_pregen"]
        dbg = opts.get("debug", ())
        for switch in ([dbg] if isinstance(dbg, str) else dbg):
            if switch not in lib.DEBUG_FLAGS:
                raise RuntimeError("unknown debug switch %r (one of %s)" % (switch, sorted(lib.DEBUG_FLAGS)))
            cfg.debug_flags |= lib.DEBUG_FLAGS[switch]
        cfg.debug_ego_per = int(opts.get("debug_ego_per", 0))
        cfg.debug_ego_pad = int(opts.get("debug_ego_pad", 0))
        cfg.debug_render_shape = int(opts.get("debug_render_shape", 0))
        self.palette = None
        self._keep = []
        if name == "simple_game":
            cfg.array_size = int(_require(opts, "array_size"))            # py_simulator.cpp:99-102
        elif name == "simple_race":                                       # py_simulator.cpp:106-122
            tt = opts.get("track_type", "straight")
            if tt not in ("straight", "circle"):
                raise RuntimeError("track_type must be 'straight' or 'circle'")
            cfg.track_type = 1 if tt == "circle" else 0
            cfg.track_width = float(np.float32(_require(opts, "track_width")))
            cfg.track_length = float(np.float32(_require(opts, "track_length")))
            cfg.track_radius = float(np.float32(_require(opts, "track_radius")))
            cfg.race_full_manouver = int(bool(opts.get("race_full_manouver", False)))
            cfg.random = int(bool(opts.get("random", False)))
            diff = opts.get("difficulty", "easy")
            cfg.difficulty_hard = 0 if diff == "easy" else 1
            cfg.reward_scale = float(opts.get("reward_scale", 1.0))
        else:                                                             # py_simulator.cpp:126-136
            conf_path = _require(opts, "xwd_conf_path")
            conf = assets.read_conf(conf_path)
            map_name = opts.get("map", conf["map"])
            if map_name not in assets.MAP_CLASSES:
                raise RuntimeError("Error loading map: " + str(map_name))  # xworld.cpp:104-105
            mc = assets.MAP_CLASSES[map_name]
            cfg.map_kind = mc["map_kind"]
            cfg.max_dim = int(opts.get("max_dim", mc["max_dim"]))
            cfg.dim = int(opts.get("dim", cfg.max_dim))
            cfg.num_goals = int(opts.get("num_goals", mc["num_goals"]))
            cfg.num_blocks = int(opts.get("num_blocks", mc["num_blocks"]))
            cfg.max_steps_factor = int(opts.get("max_steps_factor", 10))
            mode = opts.get("task_mode", "one_channel")                   # py_simulator.cpp:130-131
            if mode not in ("lang_acquisition", "one_channel"):
                raise RuntimeError("unsupported task mode: " + str(mode))  # xworld_simulator.cpp:194-196
            cfg.task_mode = 0 if mode == "lang_acquisition" else 1
            cfg.color = int(bool(opts.get("color", False)))
            fmt = opts.get("obs_format", "uint8")
            if fmt not in ("uint8", "float32"):
                raise RuntimeError("obs_format must be 'uint8' or 'float32'")
            cfg.obs_format = 1 if fmt == "float32" else 0
            # the conf's task groups in conf order (unbuilt groups are skipped with a warning); "tasks" [+ "tasks2"] override
            # them with lists of task class names / ids, "task_weights" [+ "task_weights2"] = schedule "weighted"
            # (teaching_task.cpp:204-213)
            if opts.get("tasks") is None:
                groups = assets.conf_groups(conf, opts.get("task_group"))
                if "task_weights" in opts:
                    groups[0] = (groups[0][0], groups[0][1], opts["task_weights"])
            else:
                groups = [("tasks", [assets.TASK_IDS.get(t, t) for t in opts["tasks"]], opts.get("task_weights"))]
                if opts.get("tasks2") is not None:
                    groups.append(("tasks2", [assets.TASK_IDS.get(t, t) for t in opts["tasks2"]], opts.get("task_weights2")))
            for gi, (gname, tasks, weights) in enumerate(groups):
                if weights is not None and len(weights) != len(tasks):
                    raise RuntimeError("task_weights needs one weight per task")
                if gi == 0:
                    cfg.n_tasks = len(tasks)
                    for i, t in enumerate(tasks):
                        cfg.tasks[i] = int(t)
                    if weights is not None:
                        cfg.task_schedule = 1
                        for i, x in enumerate(weights):
                            cfg.task_weights[i] = float(x)
                else:
                    cfg.n_tasks2 = len(tasks)
                    for i, t in enumerate(tasks):
                        cfg.tasks2[i] = int(t)
                    if weights is not None:
                        cfg.task_schedule2 = 1
                        for i, x in enumerate(weights):
                            cfg.task_weights2[i] = float(x)
            # FLAGS_task_groups_exclusive (py_simulator.cpp:132, default true); lang_acquisition turns it off as the
            # reference does (simulator_interface.cpp:46-48)
            cfg.task_groups_exclusive = int(bool(opts.get("task_groups_exclusive", True))) if mode != "lang_acquisition" else 0
            # the groups' "weight" keys (teacher.cpp:83-91), read by the exclusive scheduler's group sort only
            gw = opts.get("task_group_weights")
            if gw is None:
                gw = assets.conf_group_weights(conf, [g[0] for g in groups]) if opts.get("tasks") is None else [0.0] * len(groups)
            if len(gw) != len(groups):
                raise RuntimeError("task_group_weights needs one weight per task group")
            cfg.task_group_weight = float(gw[0])
            cfg.task_group_weight2 = float(gw[1]) if len(gw) > 1 else 0.0
            self.tasks = list(groups[0][1])
            self.task_groups = [(g[0], list(g[1])) for g in groups]
            cfg.visible_radius = int(opts.get("visible_radius", 0))         # py_simulator.cpp:133
            cfg.no_wall_shadow = 0 if opts.get("wall_shadow", True) else 1   # FLAGS_wall_shadow (xmap.cpp:19), C++ gflag only
            # py_simulator.cpp:127: FLAGS_curriculum.  XWorldNav.py:27-55: != 0 -> every env walks through the six levels
            # (dims 3..8) as its success rate passes the value; XWorldWalls never reads the flag
            cfg.curriculum = float(opts.get("curriculum", 0.0))
            cfg.start_level = int(opts.get("start_level", 0))
            self.palette = assets.Palette(mc["subtrees"], opts.get("assets_dir", assets.ASSETS))
            cfg.n_icons = len(self.palette)
            cfg.icons64 = self.palette.icons64.ctypes.data
            cfg.icon_type = self.palette.icon_type.ctypes.data
            cfg.icon_name = self.palette.icon_name.ctypes.data
            cfg.icon_colored = self.palette.icon_colored.ctypes.data
        self.cfg = cfg
        self._host_actions = None
        self.obs_is_float = name == "simple_race" or (name == "xworld" and cfg.obs_format == 1)
        h = C.c_void_p()
        lib.check(self.L.xwb_create(C.byref(cfg), C.byref(h)))
        self.h = h
        if self.palette is not None:
            # the strings behind the name ids: the library builds the teacher's sentences for C / C++ / TCP callers too
            enc = lambda xs: (C.c_char_p * len(xs))(*[x.encode() for x in xs])
            goals = self.palette.names["goal"]
            lib.check(self.L.xwb_set_names(self.h, enc(goals), len(goals), enc([m["name"] for m in self.palette.meta]),
                                           enc([m.get("color", "na") for m in self.palette.meta]), len(self.palette)))
        self.num_envs = int(num_envs)
        self.device = int(device)
        n = C.c_int32()
        lib.check(self.L.xwb_get_num_actions(self.h, C.byref(n)))
        self.num_actions = n.value
        hh, ww, cc = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib.check(self.L.xwb_get_screen_out_dimensions(self.h, C.byref(hh), C.byref(ww), C.byref(cc)))
        self.screen_dims = (hh.value, ww.value, cc.value)
        p, b = C.c_void_p(), C.c_size_t()
        lib.check(self.L.xwb_obs_dev(self.h, C.byref(p), C.byref(b)))
        self.obs_bytes_per_env = b.value
        self._obs_ptr = p.value
        self._views = {}

    # ------------------------------------------------------------------ life cycle
    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.xwb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self, stream):
        """The stream handle for the C ABI.  The step verbs never probe a stream by themselves (a stream nobody probed hands over
        through events: the slower path, ~12 us per xworld step, without being told), so every call on an explicit stream asks
        xwb_queue_sync_mode first: the library probes a handle it has not seen ONCE (that synchronises the stream), answers from
        its table afterwards (a handful of compares), skips the probe while THIS stream -- the handle passed, not torch's current
        one -- is being captured into a graph, and forgets handles whose streams died, so a new stream that reuses a handle is
        probed again.  (No cache on this side: the library's table is the only one that knows all of that.)"""
        if stream is None:
            return None
        h = int(getattr(stream, "cuda_stream", stream))
        if h and self.cfg.game == lib.XWB_XWORLD2D:
            m, r = C.c_int32(), C.c_int32()
            lib.check(self.L.xwb_queue_sync_mode(self.h, C.c_void_p(h), C.byref(m), C.byref(r)))
        return C.c_void_p(h)

    # ------------------------------------------------------------------ batched verbs
    def reset(self, stream=None):
        lib.check(self.L.xwb_reset(self.h, self._stream(stream)))

    def reset_done(self, stream=None):
        lib.check(self.L.xwb_reset_done(self.h, self._stream(stream)))

    def reset_masked(self, mask, stream=None):
        lib.check(self.L.xwb_reset_masked(self.h, C.c_void_p(mask.data_ptr()), self._stream(stream)))

    def reset_env(self, env, stream=None):
        """SimulatorInterface::reset_game of one env slot."""
        lib.check(self.L.xwb_reset_env(self.h, int(env), self._stream(stream)))

    def step(self, actions=None, act_rep=1, stream=None):
        ptr = None if actions is None else C.c_void_p(actions.data_ptr())
        lib.check(self.L.xwb_step(self.h, ptr, int(act_rep), self._stream(stream)))

    def step_host(self, actions, act_rep=1, stream=None):
        """xwb_step_host: the action ids in HOST memory -- a numpy int32 array or a CPU torch tensor (a pinned one is read by
        the step kernel in place: keep it unchanged until `stream` has passed the call)."""
        # the kernel reads int32 action ids straight from this memory: anything else would be read as garbage without an error
        if hasattr(actions, "data_ptr"):
            import torch
            if actions.is_cuda or actions.dtype != torch.int32 or actions.numel() != self.num_envs or not actions.is_contiguous():
                raise ValueError("step_host: a contiguous CPU int32 tensor of num_envs = %d action ids is required (got %s %s, cuda=%s)"
                                 % (self.num_envs, tuple(actions.shape), actions.dtype, actions.is_cuda))
            ptr = actions.data_ptr()
        else:
            import numpy as np
            if not isinstance(actions, np.ndarray) or actions.dtype != np.int32 or actions.size != self.num_envs or not actions.flags["C_CONTIGUOUS"]:
                raise ValueError("step_host: a C-contiguous numpy int32 array of num_envs = %d action ids is required" % self.num_envs)
            ptr = actions.ctypes.data
        self._host_actions = actions                         # kept alive until the next call (a pinned buffer is read in place)
        lib.check(self.L.xwb_step_host(self.h, C.c_void_p(ptr), int(act_rep), self._stream(stream)))

    def step_n(self, n_steps, act_rep=1, stream=None):
        """n_steps x step_autoreset under the built-in random policy; one launch for the simple games."""
        lib.check(self.L.xwb_step_n(self.h, int(n_steps), int(act_rep), self._stream(stream)))

    def run(self, iterations, act_rep=1, autoreset=False, stream=None):
        """xwb_run: `iterations` x (step; reset_done) -- or x step_autoreset -- under the built-in policy, one call into the library"""
        lib.check(self.L.xwb_run(self.h, int(iterations), int(act_rep), 1 if autoreset else 0, self._stream(stream)))

    def step_autoreset(self, actions=None, act_rep=1, stream=None):
        ptr = None if actions is None else C.c_void_p(actions.data_ptr())
        lib.check(self.L.xwb_step_autoreset(self.h, ptr, int(act_rep), self._stream(stream)))

    def check_errors(self, stream=None):
        n = C.c_int32()
        lib.check(self.L.xwb_check_errors(self.h, self._stream(stream), C.byref(n)))
        return n.value

    def queue_sync_mode(self, stream=None):
        """("events" | "epochs", reason) for calls made on `stream` (xwb_queue_sync_mode; probes the stream once)."""
        m, r = C.c_int32(), C.c_int32()
        lib.check(self.L.xwb_queue_sync_mode(self.h, self._stream(stream), C.byref(m), C.byref(r)))
        return ("events" if m.value == lib.XWB_QUEUE_SYNC_EVENTS else "epochs"), lib.SYNC_REASONS[r.value]

    def step_path(self):
        """xwb_step_path: {"path", "queue_sync", "shadow_breaks"} of the last step call"""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        lib.check(self.L.xwb_step_path(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"path": lib.STEP_PATHS[a.value], "queue_sync": ("auto", "events", "epochs")[b.value], "shadow_breaks": c.value}

    def done_count(self, stream=None):
        n = C.c_int32()
        lib.check(self.L.xwb_done_count(self.h, self._stream(stream), C.byref(n)))
        return n.value

    # ------------------------------------------------------------------ device views (torch)
    def _view(self, key, getter, shape, typestr):
        import torch
        if key not in self._views:
            p = C.c_void_p()
            lib.check(getter(self.h, C.byref(p)))
            self._views[key] = torch.as_tensor(_DevArray(p.value, shape, typestr, self),
                                               device="cuda:%d" % self.device)
        return self._views[key]

    @property
    def reward(self):
        return self._view("reward", self.L.xwb_reward_dev, (self.num_envs,), "<f4")

    @property
    def episode(self):
        """resets so far per env (the RNG's episode index), as int32"""
        return self._view("episode", self.L.xwb_episode_dev, (self.num_envs,), "<i4")

    @property
    def game_over_codes(self):
        return self._view("done", self.L.xwb_game_over_dev, (self.num_envs,), "|u1")

    @property
    def minstd_state(self):
        """rng = "minstd": the per-env minstd_rand0 states, uint32 as an int32 view"""
        return self._view("minstd", self.L.xwb_minstd_state_dev, (self.num_envs,), "<i4")

    @property
    def ego_render_path(self):
        """egocentric xworld: "span" (cells -> evaluated pixels -> gather) or "per_env" (one workgroup per env)"""
        v = C.c_int32(0)
        lib.check(self.L.xwb_ego_render_path(self.h, C.byref(v)))
        return "span" if v.value else "per_env"

    @property
    def actions(self):
        return self._view("actions", self.L.xwb_actions_dev, (self.num_envs,), "<i4")

    @property
    def num_steps(self):
        return self._view("num_steps", self.L.xwb_num_steps_dev, (self.num_envs,), "<i4")

    @property
    def success(self):
        return self._view("success", self.L.xwb_success_dev, (self.num_envs,), "|u1")

    @property
    def grid(self):
        """xworld: [num_envs, max_dim, max_dim] cell codes, int16 view: code & 0x7fff = palette icon + 1 (0 = empty);
        the sign bit marks the goals in the task's target set."""
        d = self.cfg.max_dim
        return self._view("grid", self.L.xwb_xw_grid_dev, (self.num_envs, d, d), "<i2")

    @property
    def obs(self):
        """[num_envs, context*c, h, w]; uint8 (simple_game, xworld: planar B,G,R) or float32 (simple_race; xworld
        created with obs_format="float32": pixel * 1/255, the scaling of py_simulator's get_state)."""
        import torch
        if "obs" not in self._views:
            h, w, c = self.screen_dims
            ctx = self.cfg.context
            if self.obs_is_float:
                shape, ts = (self.num_envs, ctx * c, h, w), "<f4"
            else:
                shape, ts = (self.num_envs, ctx * c, h, w), "|u1"
            self._views["obs"] = torch.as_tensor(_DevArray(self._obs_ptr, shape, ts, self),
                                                 device="cuda:%d" % self.device)
        return self._views["obs"]

    def bind_results(self, tensor):
        """float32 [num_envs, 2] device tensor (or None): every step also writes (reward, game_over code) there."""
        if tensor is None:
            lib.check(self.L.xwb_bind_results(self.h, None))
        else:
            assert tensor.is_contiguous() and tensor.dtype.itemsize == 4 and tensor.numel() == 2 * self.num_envs
            lib.check(self.L.xwb_bind_results(self.h, C.c_void_p(tensor.data_ptr())))
        self._results = tensor

    def bind_results_ring(self, tensor):
        """float32 [slots, num_envs, 2] device tensor: the k-th step call after the bind writes (reward, game_over code)
        of every env into slot k % slots -- a per-step record of a rollout without any per-step host call."""
        assert tensor.is_contiguous() and tensor.dtype.itemsize == 4 and tensor.dim() == 3
        assert tensor.shape[1] == self.num_envs and tensor.shape[2] == 2
        lib.check(self.L.xwb_bind_results_ring(self.h, C.c_void_p(tensor.data_ptr()), int(tensor.shape[0])))
        self._results = tensor

    def bind_obs(self, tensor):
        """Redirect the observation output into caller-owned device memory (e.g. a shard of a gathered tensor)."""
        assert tensor.is_contiguous() and tensor.numel() * tensor.element_size() == self.num_envs * self.obs_bytes_per_env
        lib.check(self.L.xwb_bind_obs(self.h, C.c_void_p(tensor.data_ptr())))
        h, w, c = self.screen_dims
        self._obs_ptr = tensor.data_ptr()
        self._views["obs"] = tensor.view(self.num_envs, self.cfg.context * c, h, w)
        self._bound = tensor

    # ------------------------------------------------------------------ per-env host access
    def set_draw(self, on):
        """xwb_xw_set_draw: False = the verbs store no pixels (frames are drawn elsewhere from pack_grids); obs is then stale."""
        lib.check(self.L.xwb_xw_set_draw(self.h, int(bool(on))))

    def pack_grids(self, grids, flags=None, stream=None):
        """xwb_xw_pack_grids: the draw state of every env -- the cell codes its current frame shows into `grids` (int16 / uint16
        [num_envs, max_dim * max_dim] device tensor) and the context-ring flag of its last draw into `flags` (uint8 [num_envs];
        optional when context == 1)."""
        lib.check(self.L.xwb_xw_pack_grids(self.h, C.c_void_p(grids.data_ptr()), C.c_void_p(flags.data_ptr()) if flags is not None else None,
                                           self._stream(stream)))

    def render_grids(self, grids, flags, out, n_envs=None, stream=None):
        """xwb_xw_render_grids: draws n_envs frames (default: the rows of `grids`) from cell codes with this batch's tile table
        and frame format into `out` [n_envs, ...] -- the root of a sharded batch draws every shard's frames with it."""
        n = int(grids.shape[0] if n_envs is None else n_envs)
        lib.check(self.L.xwb_xw_render_grids(self.h, C.c_void_p(grids.data_ptr()), C.c_void_p(flags.data_ptr()) if flags is not None else None,
                                             n, C.c_void_p(out.data_ptr()), self._stream(stream)))

    def env_state(self, env=0, stream=None):
        st = lib.XwbEnvState()
        lib.check(self.L.xwb_get_env_state(self.h, int(env), self._stream(stream), C.byref(st)))
        return st

    def env_obs(self, env=0, stream=None):
        dt = np.float32 if self.obs_is_float else np.uint8
        out = np.empty(self.obs_bytes_per_env // np.dtype(dt).itemsize, dt)
        lib.check(self.L.xwb_get_env_obs(self.h, int(env), self._stream(stream), out.ctypes.data, self.obs_bytes_per_env))
        return out

    def env_grid(self, env=0, stream=None, raw=False):
        """Cell codes (palette icon + 1, 0 empty); raw=True keeps bit 15 = the goal is in the task's target set."""
        d = self.cfg.max_dim
        out = np.empty(d * d, np.uint16)
        lib.check(self.L.xwb_get_env_grid(self.h, int(env), self._stream(stream), out.ctypes.data))
        out = out.reshape(d, d)
        return out if raw else out & np.uint16(0x7fff)

    def env_target_cells(self, env=0, stream=None):
        """(x, y) cells (max_dim coordinates, row-major order) of the goals in the task's target set."""
        g = self.env_grid(env, stream, raw=True)
        ys, xs = np.nonzero(g & np.uint16(0x8000))
        return [(int(x), int(y)) for x, y in zip(xs, ys)]

    def load_map(self, env, grid, agent_x, agent_y, target_name=None, dim=None, task=None, target=-1):
        """Replay a map.  task=None: XWorld3DNavTarget with `target_name`; else grid codes carry bit 15 on the
        target goals and `target` is the middle cell (y * max_dim + x) for XWorld3DNavTargetBetween."""
        g = np.ascontiguousarray(grid, np.uint16)
        dim = int(self.cfg.dim if dim is None else dim)
        if task is None:
            lib.check(self.L.xwb_xw_load_map(self.h, int(env), g.ctypes.data, int(agent_x), int(agent_y),
                                             int(target_name), dim))
        else:
            lib.check(self.L.xwb_xw_load_map_task(self.h, int(env), g.ctypes.data, int(agent_x), int(agent_y), dim,
                                                  int(assets.TASK_IDS.get(task, task)), int(target)))

    def set_agent_dir(self, env, d):
        """egocentric heading: 0 right, 1 down, 2 left, 3 up"""
        lib.check(self.L.xwb_xw_set_agent_dir(self.h, int(env), int(d)))

    def set_goal_pose(self, env, x, y, yaw, scale=1.0, offset=0.0):
        lib.check(self.L.xwb_xw_set_goal_pose(self.h, int(env), int(x), int(y), float(yaw), float(scale), float(offset)))

    def refresh_obs(self, env):
        lib.check(self.L.xwb_xw_refresh_obs(self.h, int(env)))

    def sentence(self, env=0, stream=None):
        """The teacher's sentence of one env after the last call (language.py); "" where the reference shows "-".  With two
        task groups the first one (conf order) that speaks wins: Task::teacher_speak only records into an empty buffer
        (teaching_task.cpp:118-127)."""
        st = self.env_state(env, stream)
        if st.xw_group_ran == 1:                    # exclusive scheduling: only the group the last teach() ran can have spoken
            return self._group_sentence(env, stream, st, st.xw_task2, st.xw_stage2, st.xw_event2, st.xw_target2, st.xw_steps_in_task2)
        out = self._group_sentence(env, stream, st, st.xw_task, st.xw_stage, st.xw_event, st.xw_target, st.xw_steps_in_task)
        if not out and self.cfg.n_tasks2 > 0 and st.xw_group_ran < 0:
            out = self._group_sentence(env, stream, st, st.xw_task2, st.xw_stage2, st.xw_event2, st.xw_target2, st.xw_steps_in_task2)
        return out

    def sentence_c(self, env=0, stream=None):
        """The same sentence built inside libxwb.so (xwb_sentence: what state packets and the TCP endpoint carry)."""
        need = C.c_size_t()
        lib.check(self.L.xwb_sentence(self.h, int(env), self._stream(stream), None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        lib.check(self.L.xwb_sentence(self.h, int(env), self._stream(stream), buf, need.value, C.byref(need)))
        return buf.value.decode()

    def _group_sentence(self, env, stream, st, task, stage, event, target, steps_in_task):
        from . import language
        if task in (5, 7):
            # 2-D-native Target / ColorTarget: they speak on the teach() call that picked the target, and "Time up ." on
            # the one_channel step that runs out of time (xworld_task.py:205-211): back to idle with the target still recorded
            if stage == 0 and event == 0 and target >= 0 and st.num_steps > 0 and self.cfg.task_mode == 1:
                return language.sentence_2d_timeup(task)
            if stage != 1 or steps_in_task != 0 or target < 0:
                return ""
            d = self.cfg.max_dim
            icon = int(self.env_grid(env, stream)[target // d, target % d]) - 1
            if icon < 0:                                   # (two groups: the 3-D stage may have moved the goal away since)
                return ""
            m = self.palette.meta[icon]
            return language.sentence_2d(task, m["name"], m.get("color", "na"), self.cfg.seed,
                                        self.cfg.env_gid0 + int(env), st.episode, int(st.num_steps))
        sn = st.xw_sentence_names
        return language.sentence(task, stage, event, self.palette.names["goal"], sn & 0xffff, sn >> 16,
                                 (target >> 8) & 7 if task == 3 and target >= 0 else 0,
                                 self.cfg.seed, self.cfg.env_gid0 + int(env), st.episode)

    def task_performance(self, stream=None):
        """Teacher::report_task_performance's numbers for the whole batch since it was created: ({task class: (successes,
        failures, success_steps, time_ups)} for the classes that occurred, games reset)."""
        arr = (lib.XwbTaskPerformance * 9)()
        resets = C.c_int64()
        lib.check(self.L.xwb_get_task_performance(self.h, self._stream(stream), arr, C.byref(resets)))
        out = {lib.TASK_CLASSES[k]: (p.successes, p.failures, p.success_steps, p.time_ups)
               for k, p in enumerate(arr) if p.successes + p.failures}
        return out, resets.value

    def task_performance_report(self, stream=None):
        """the text SimulatorInterface::teacher_report_task_performance logs (teacher.cpp:175-200)"""
        need = C.c_size_t()
        lib.check(self.L.xwb_task_performance_report(self.h, self._stream(stream), None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        lib.check(self.L.xwb_task_performance_report(self.h, self._stream(stream), buf, need.value, C.byref(need)))
        return buf.value.decode()

    def save_state(self, include_obs=True):
        """The batch's whole dynamic state as one numpy uint8 blob (checkpoint); load_state() resumes bit for bit."""
        n = C.c_size_t()
        lib.check(self.L.xwb_state_bytes(self.h, int(include_obs), C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        lib.check(self.L.xwb_save_state(self.h, int(include_obs), buf.ctypes.data, n.value))
        return buf

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        lib.check(self.L.xwb_load_state(self.h, blob.ctypes.data, blob.size))

    def race_set_car(self, env, x, y, angle):
        lib.check(self.L.xwb_race_set_car(self.h, int(env), float(x), float(y), float(angle)))

    def state_packet(self, env=0, reward=0.0, stream=None):
        """SimulatorInterface::get_state(reward) of one env in the reference's StatePacket wire layout."""
        need = C.c_size_t()
        lib.check(self.L.xwb_get_state_packet(self.h, int(env), float(reward), self._stream(stream), None, 0, C.byref(need)))
        buf = (C.c_uint8 * need.value)()
        lib.check(self.L.xwb_get_state_packet(self.h, int(env), float(reward), self._stream(stream), buf, need.value,
                                              C.byref(need)))
        return bytes(buf)

    def tile_table(self):
        need = C.c_size_t()
        lib.check(self.L.xwb_xw_get_tile_table(self.h, None, 0, C.byref(need)))
        out = np.empty(need.value, np.uint8)
        lib.check(self.L.xwb_xw_get_tile_table(self.h, out.ctypes.data, need.value, C.byref(need)))
        c = self.screen_dims[2]
        return out.reshape(len(self.palette), c, 12, 12)

    # ------------------------------------------------------------------ profiling hooks
    def profile_begin(self):
        lib.check(self.L.xwb_profile_begin(self.h))

    def profile_end(self, kernel, stream=None):
        us, n = C.c_double(), C.c_int64()
        lib.check(self.L.xwb_profile_end(self.h, self._stream(stream), kernel.encode(), C.byref(us), C.byref(n)))
        return us.value, n.value

    def profile_stop(self):
        lib.check(self.L.xwb_profile_stop(self.h))
