#!/usr/bin/env python3
"""Generates tests/golden/*.json by IMPORTING the reference's Python modules (build container only).

    python tests/golden/make_golden.py            # needs /root/reference, numpy, oracle/liboracle.so

Nothing of the reference's source is copied: the fixtures are inputs and the outputs the reference
modules produced for them.  What is imported from /root/reference:
    python/maze2d.py                      spanning_tree_maze_generator, bfs
    python/py_util.py
    games/xworld/maps/xworld_env.py, XWorldNav.py, XWorldWalls.py
    games/xworld3d/tasks/xworld3d_task.py, XWorld3DNavTarget.py, XWorld3DNavTargetNear.py,
    XWorld3DNavTargetBetween.py, XWorld3DNavTargetDirection.py, XWorld3DNavTargetAvoid.py

Harness shims (this file, clearly not reference code):
  * `py_gflags`: in the reference this module is provided by the embedding C++ program
    (python/py_init.cpp:37-58) and simply returns gflags values; here get_flag() reads a dict holding the
    flag values of the configuration under test (visible_radius 0, max_steps_factor 10, curriculum 0,
    task_mode lang_acquisition).
  * `context_free_grammar`: the teacher's sentence generator (language side, out of scope; the real module is
    Python-2 only): a no-op CFG so that XWorld3DTask can be constructed.
  * Python 2 -> 3: XWorldEnv.set_dims uses `/` on ints (xworld_env.py:129-130) and
    get_all_possible_names returns dict.keys() that is later shuffled (:292); both are patched to their
    Python-2 meaning (integer division, list).
  * the ~30 lines of C++ glue between the simulator and the Python task (teaching_task.cpp:64-116: push the
    entities / events into the env, call the stage, read event + reward; xmap.cpp:76-101: 4-way move into an
    empty in-bounds cell, contact list; xworld_simulator.cpp:124-137: "collision:<ids>\n") are restated
    in `Harness` below.

Fixtures written:
  maze.json      reference maze generator driven by the xwb-rng-v1 shuffle decisions -> maze rows
  bfs.json       reference bfs() reachability on random obstacle maps
  maps_nav.json  XWorldNav.reset() maps (python random seeded), entity lists as palette indices,
                 + reference _reachable() per goal
  maps_walls.json  same for XWorldWalls
  maps_levels.json XWorldNav at curriculum levels 0..4 (3x3 .. 7x7 inside the 8x8 world): placed entities and the padded C++ view
  teacher.json   XWorld3DNavTarget idle/navigation_reward run over random action strings on those maps:
                 per step action, reward, event, stage, agent cell, action success
  tasks_ego.json the five tasks with FLAGS_visible_radius = 3: entity poses (yaw, scale, offset) as the reference's
                 map generator drew them, six first-person actions, the agent's yaw after every step
  sentences.json the reference's CFG with each task's grammar: sentences for random bindings + the choices behind them
  tasks2d.json   the 2-D-native group of confs/walls.json (XWorldNavTarget / Near / ColorTarget / Between, rule D14b)
                 as a one-task group, both task modes, every teach() call of a 70-step episode
  groups.json    two task groups (an XWorld3DNav* and an XWorldNav* task) run non-exclusively in both conf orders: every
                 teach() call of 60-step episodes, per group and summed
  groups_exclusive.json  the same two groups under EXCLUSIVE scheduling (task_mode one_channel): the per-teach() group sort,
                 the one group that runs, mid-episode 3-D idle stages with the entity list they leave
  tasks.json     all five tasks of the XWorld3DNav group (Target, Near, Between, Direction, Avoid): the idle stage
                 driven by logged decisions (see DecisionRandom), the map after the teacher's rearrangement,
                 the target cells, and a random-action trace as in teacher.json
"""
import ctypes as C
import json
import math
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("XWORLD_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

FLAGS = {"visible_radius": 0, "max_steps_factor": 10, "task_mode": "lang_acquisition", "curriculum": 0.0}

py_gflags = types.ModuleType("py_gflags")
py_gflags.get_flag = lambda name: FLAGS[name]
py_gflags.log_info = lambda *a: None
py_gflags.log_fatal = lambda *a: None
sys.modules["py_gflags"] = py_gflags


class _CFG(object):
    def __init__(self, *a, **k):
        pass

    def bind(self, *a):
        pass

    def generate(self, *a):
        return ""

    def generate_all(self, *a):
        return []

    def set_production_rule(self, *a):
        pass

    def total_possible_sentences(self):
        return 0

    def show(self):
        pass


cfg_mod = types.ModuleType("context_free_grammar")
cfg_mod.CFG = _CFG
sys.modules["context_free_grammar"] = cfg_mod

for sub in ("python", "games/xworld/maps", "games/xworld3d/tasks"):
    sys.path.insert(0, os.path.join(REF, sub))

import maze2d                                   # noqa: E402  (reference)
import xworld_env                               # noqa: E402  (reference)
from XWorldNav import XWorldNav                 # noqa: E402  (reference)
from XWorldWalls import XWorldWalls             # noqa: E402  (reference)
from XWorld3DNavTarget import XWorld3DNavTarget  # noqa: E402  (reference)

import _oracle as O                             # noqa: E402  (only for the xwb-rng-v1 stream + palettes)


def _set_dims(self, h, w):
    # xworld_env.py:118-134 with Python-2 integer division
    assert h >= 1 and w >= 1
    assert h <= self.max_height and w <= self.max_width
    self.height = h
    self.width = w
    self.offset_h = (self.max_height - h) // 2
    self.offset_w = (self.max_width - w) // 2
    self.pad_blocks = self._XWorldEnv__padding_walls()
    existing_entities = [e.loc for e in self.entities]
    self.available_grids = list(set(self._XWorldEnv__generate_all_grids(h, w)) - set(existing_entities))
    self.changed = True


xworld_env.XWorldEnv.set_dims = _set_dims
_orig_names = xworld_env.XWorldEnv.get_all_possible_names
xworld_env.XWorldEnv.get_all_possible_names = lambda self, t: sorted(_orig_names(self, t))

ITEM_PATH = os.path.join(REF, "games", "xworld", "images")


# ------------------------------------------------------------------ maze ----
class StreamRandom(object):
    """random.shuffle replacement drawing from the xwb-rng-v1 stream: Fisher-Yates i = n-1..1, j = below(i+1)."""

    def __init__(self, seed, gid, episode):
        self.s = O.Stream()
        O.lib().orc_stream_init(C.byref(self.s), seed, gid, episode, 0)

    def shuffle(self, lst):
        for i in range(len(lst) - 1, 0, -1):
            j = O.lib().orc_stream_below(C.byref(self.s), i + 1)
            lst[i], lst[j] = lst[j], lst[i]


def gen_maze():
    cases = []
    real_random = maze2d.random
    try:
        for X in (3, 4, 5, 6, 7, 8, 9, 11, 12, 15, 16):
            for k in range(6):
                seed, gid, ep = 1000 + X, 7 * k + 1, k
                maze2d.random = StreamRandom(seed, gid, ep)
                maze = maze2d.spanning_tree_maze_generator(X, X)
                cases.append({"X": X, "seed": seed, "gid": gid, "episode": ep,
                              "maze": ["".join(r) for r in maze]})
    finally:
        maze2d.random = real_random
    return cases


def gen_bfs():
    rnd = random.Random(12345)
    cases = []
    for k in range(300):
        X = rnd.randint(2, 11)
        Y = rnd.randint(2, 11)
        cells = [(x, y, 0) for x in range(X) for y in range(Y)]
        rnd.shuffle(cells)
        start, end = cells[0], cells[1]
        n_obst = rnd.randint(0, max(0, (X * Y) // 2))
        obst = cells[2:2 + n_obst]
        random.seed(k)
        path = maze2d.bfs(start, end, X, Y, obst)
        cases.append({"X": X, "Y": Y, "start": start[:2], "end": end[:2],
                      "obstacles": [o[:2] for o in obst], "reachable": path is not None,
                      "path_len": None if path is None else len(path)})
    return cases


# ------------------------------------------------------------------ maps ----
def entity_records(env, pal):
    """cpp_get_entities() -> (type, x, y, palette icon, name id, serial) in the order C++ receives them."""
    path_to_icon = {m["path"]: i for i, m in enumerate(pal.meta)}
    out = []
    for e in env.cpp_get_entities():
        rel = os.path.relpath(e["asset_path"], ITEM_PATH)
        t = O.TYPE_ID[e["type"]]
        icon = path_to_icon[rel]
        assert pal.meta[icon]["name"] == e["name"], (rel, e["name"])
        serial = int(e["id"].split("_")[-1])
        out.append([t, int(e["loc"][0]), int(e["loc"][1]), icon, int(pal.name_arr[icon]), serial])
    return out


def gen_maps(cls, pal, n_maps, seed0):
    maps = []
    env = cls(ITEM_PATH)
    for k in range(n_maps):
        random.seed(seed0 + k)
        env.reset()
        h, w = env.get_dims()
        ents = entity_records(env, pal)
        task = XWorld3DNavTarget(env)
        agent = [e for e in env.get_entities() if e.type == "agent"][0]
        reach = [bool(task._reachable(agent.loc, g.loc)) for g in env.get_goals()]
        maps.append({"py_seed": seed0 + k, "dim": h, "max_dim": env.get_max_dims()[0], "entities": ents,
                     "goal_reachable": reach})
    return maps


def gen_maps_levels(pal, per_level, seed0):
    """XWorldNav at the curriculum levels 0..4 (XWorldNav.py:27-39: dims 3..7 inside the 8x8 world, goals / blocks per level):
    the entities in env coordinates (what the generator placed) and the C++ view cpp_get_entities() returns -- shifted
    by the padding offset and followed by the padding wall (xworld_env.py:354-365, 454-493)."""
    out = []
    FLAGS["curriculum"] = 1e9                      # level logic on, never advancing
    try:
        for level in range(5):
            env = XWorldNav(ITEM_PATH, start_level=level)
            for k in range(per_level):
                random.seed(seed0 + 100 * level + k)
                env.reset()
                h, w = env.get_dims()
                task = XWorld3DNavTarget(env)
                agent = [e for e in env.get_entities() if e.type == "agent"][0]
                reach = [bool(task._reachable(agent.loc, g.loc)) for g in env.get_goals()]
                # cpp_get_entities() shifts the entities' own locs by the offset (update_entities_from_cpp undoes it in
                # the real flow), so it comes last here
                cpp = entity_records(env, pal)
                n_actual = len(env.get_entities())
                actual = [[t, x - env.offset_w, y - env.offset_h, icon, name, serial] for t, x, y, icon, name, serial in cpp[:n_actual]]
                out.append({"py_seed": seed0 + 100 * level + k, "level": level, "dim": h, "max_dim": env.get_max_dims()[0],
                            "num_goals": len(env.get_goals()), "num_blocks": len(env.get_blocks()),
                            "entities": actual, "cpp_entities": cpp, "goal_reachable": reach})
    finally:
        FLAGS["curriculum"] = 0.0
    return out


# --------------------------------------------------------------- teacher ----
class Harness(object):
    """The C++ side of one XWorld2D env as far as a Python task can see it."""

    def __init__(self, env):
        self.env = env
        self.ents = [dict(e) for e in env.cpp_get_entities()]
        for e in self.ents:
            e["loc"] = tuple(e["loc"])
        self.H, self.W = env.get_max_dims()
        self.agent = [e for e in self.ents if e["type"] == "agent"][0]
        self.game_events = ""
        self.success = False

    def cell(self, x, y):
        return [e for e in self.ents if e["loc"][0] == x and e["loc"][1] == y]

    def act(self, a):
        # XAgent::act (xitem.cpp:89-101) + XMap::move_item (xmap.cpp:76-101) + record_collision_events
        x, y = self.agent["loc"][0], self.agent["loc"][1]
        tx, ty = [(x, y - 1), (x, y + 1), (x - 1, y), (x + 1, y)][a]
        contact = []
        ok = False
        if 0 <= tx < self.W and 0 <= ty < self.H:
            items = self.cell(tx, ty)
            contact = [i["id"] for i in items if i["id"] != self.agent["id"]]
            ok = len(items) == 0
        if ok:
            self.agent["loc"] = (tx, ty, 0)
        if contact:
            self.game_events += "collision:" + "|".join(contact) + "\n"
        self.success = ok

    def py_stage(self, task, stage):
        # Task::py_stage, teaching_task.cpp:64-116
        self.env.update_entities_from_cpp([dict(e) for e in self.ents])
        self.env.update_agent_sentence_from_cpp("")
        self.env.update_agent_action_success_from_cpp(self.success)
        ev, self.game_events = self.game_events, ""
        self.env.update_game_event_from_cpp(ev)
        ret = getattr(task, stage)()
        assert not self.env.env_changed() or stage == "idle" or True
        event = task.get_event()
        return ret[0], float(ret[1]), event


def gen_teacher(cls, pal, n_maps, seed0, steps):
    runs = []
    env = cls(ITEM_PATH)
    rnd = random.Random(999)
    import XWorld3DNavTarget as nav_mod
    for k in range(n_maps):
        random.seed(seed0 + k)
        env.reset()
        env.env_changed()
        ents = entity_records(env, pal)
        h = Harness(env)
        task = XWorld3DNavTarget(env)
        task.reset()
        picked = {}
        real_choice = nav_mod.random.choice

        def choice(seq, _p=picked, _r=real_choice):
            v = _r(seq)
            _p["index"] = list(seq).index(v)
            _p["n"] = len(seq)
            return v
        nav_mod.random.choice = choice
        try:
            stage, reward, event = h.py_stage(task, "idle")
        finally:
            nav_mod.random.choice = real_choice
        assert stage == "navigation_reward" and reward == 0.0 and event == ""
        # per step: [action, reward, event, stage, agent x, agent y, success]; a few steps past the end
        trace = []
        after_end = 0
        for t in range(steps):
            a = rnd.randrange(4)
            h.act(a)
            stage, reward, event = h.py_stage(task, stage)
            trace.append([a, reward, event, stage, h.agent["loc"][0], h.agent["loc"][1], int(bool(h.success))])
            if stage == "terminal":
                after_end += 1
                if after_end > 3:
                    break
        runs.append({"py_seed": seed0 + k, "dim": env.get_dims()[0], "max_dim": env.get_max_dims()[0],
                     "entities": ents, "target_pick": picked["index"], "n_candidates": picked["n"],
                     "target_name": int(pal.names["goal"].index(task.target[0].name)), "trace": trace})
    return runs


# ------------------------------------------------------ the five nav tasks ----
class DecisionRandom(object):
    """Stands in for the `random` module inside ONE task module.  Every call takes its decision(s) as
    below(n) values from a seeded generator and logs them, so the oracle / product can be driven through the
    same decisions ("xwb-taskgen-v1"):
      shuffle(lst): the first two positions of a Fisher-Yates shuffle -- below(n), then below(n-1) -- because
                    the tasks only ever use lst[:2] or lst[0] of a shuffled list;
      choice(seq):  seq[below(n)]; a list of bare (x, y, 0) cells is first put in row-major order, since its
                    order comes from the env's randomly shuffled available_grids list."""

    def __init__(self, seed):
        self.rnd = random.Random(seed)
        self.log = []

    def below(self, n):
        v = self.rnd.randrange(n)
        self.log.append(v)
        return v

    def shuffle(self, lst):
        n = len(lst)
        if n == 0:
            return
        lst.insert(0, lst.pop(self.below(n)))
        if n >= 2:
            lst.insert(1, lst.pop(1 + self.below(n - 1)))

    def choice(self, seq):
        seq = list(seq)
        if seq and isinstance(seq[0], tuple) and len(seq[0]) == 3 and not isinstance(seq[0][0], tuple):
            seq = sorted(seq, key=lambda c: (c[1], c[0]))
        return seq[self.below(len(seq))]


class EgoHarness(Harness):
    """Harness for FLAGS_visible_radius > 0: XAgent::act with the six first-person actions (xitem.cpp:103-155)
    and XMap::move_item, for which a turn is a failed move onto the agent's own cell (xmap.cpp:76-101)."""

    @staticmethod
    def facing(yaw):                                   # XItem::get_item_facing_dir, xitem.cpp:65-78
        eps = 1e-4
        if abs(yaw) < eps:
            return 0                                   # right
        if abs(yaw - math.pi / 2) < eps:
            return 1                                   # down
        if abs(yaw - math.pi) < eps:
            return 2                                   # left
        return 3                                       # up

    def act(self, a):
        x, y = self.agent["loc"][0], self.agent["loc"][1]
        d = self.facing(self.agent["yaw"])
        fwd = [(1, 0), (0, 1), (-1, 0), (0, -1)][d]
        left = [(0, -1), (1, 0), (0, 1), (-1, 0)][d]
        tx, ty = x, y
        if a == 0:
            tx, ty = x + fwd[0], y + fwd[1]
        elif a == 1:
            tx, ty = x - fwd[0], y - fwd[1]
        elif a == 2:
            tx, ty = x + left[0], y + left[1]
        elif a == 3:
            tx, ty = x - left[0], y - left[1]
        elif a == 4:
            self.agent["yaw"] -= math.pi / 2
            if self.agent["yaw"] < -math.pi / 2 - 1e-4:
                self.agent["yaw"] += 2 * math.pi
        else:
            self.agent["yaw"] += math.pi / 2
            if self.agent["yaw"] > math.pi + 1e-4:
                self.agent["yaw"] -= 2 * math.pi
        contact = []
        ok = False
        if 0 <= tx < self.W and 0 <= ty < self.H:
            items = self.cell(tx, ty)
            contact = [i["id"] for i in items if i["id"] != self.agent["id"]]
            ok = len(items) == 0
        if ok:
            self.agent["loc"] = (tx, ty, 0)
        if contact:
            self.game_events += "collision:" + "|".join(contact) + "\n"
        self.success = ok


def gen_tasks(pal, n_maps, seed0, steps, ego=0):
    import importlib
    FLAGS["visible_radius"] = ego
    try:
        return _gen_tasks(pal, n_maps, seed0, steps, ego, importlib)
    finally:
        FLAGS["visible_radius"] = 0


def _gen_tasks(pal, n_maps, seed0, steps, ego, importlib):
    names = ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween",
             "XWorld3DNavTargetDirection", "XWorld3DNavTargetAvoid"]
    out = {}
    env = XWorldNav(ITEM_PATH)
    rnd = random.Random(4242)
    for name in names:
        mod = importlib.import_module(name)
        cls = getattr(mod, name)
        runs = []
        k = 0
        while len(runs) < n_maps and k < 4 * n_maps:
            k += 1
            random.seed(seed0 + k)
            env.reset()
            env.env_changed()
            before = entity_records(env, pal)
            poses = [[float(e["yaw"]), float(e["scale"]), float(e["offset"])] for e in env.cpp_get_entities()]
            h = EgoHarness(env) if ego else Harness(env)
            task = cls(env)
            task.reset()
            fake = DecisionRandom(seed0 * 7 + k)
            real = mod.random
            mod.random = fake
            try:
                h.env.update_entities_from_cpp([dict(e) for e in h.ents])
                h.env.update_agent_sentence_from_cpp("")
                h.env.update_agent_action_success_from_cpp(False)
                h.env.update_game_event_from_cpp("")
                try:
                    ret = task.idle()
                except AssertionError as err:                # "map too crowded?" -- the reference process would die
                    continue
            finally:
                mod.random = real
            assert ret[0] == "navigation_reward" and ret[1] == 0.0
            task.get_event()
            if env.env_changed():                            # Task::py_stage -> game_->update_environment()
                h.ents = [dict(e) for e in env.cpp_get_entities()]
                for e in h.ents:
                    e["loc"] = tuple(int(v) for v in e["loc"])
                h.agent = [e for e in h.ents if e["type"] == "agent"][0]
            after = entity_records(env, pal)
            goals = env.get_goals()
            rec = {"py_seed": seed0 + k, "dim": env.get_dims()[0], "max_dim": env.get_max_dims()[0],
                   "entities_before": before, "entities_after": after, "decisions": list(fake.log)}
            if ego:
                rec["poses"] = poses
            if name == "XWorld3DNavTargetBetween":
                l1, l2 = task.target
                rec["between"] = [int((l1[0] + l2[0]) // 2), int((l1[1] + l2[1]) // 2)]
                rec["target_cells"] = []
            elif name == "XWorld3DNavTargetDirection":
                referent, direction = task.target
                agent = [e for e in env.get_entities() if e.type == "agent"][0]
                fn = getattr(task, "_XWorld3DNavTargetDirection__compute_triple_direction")
                cells = []
                for g in goals:
                    near = task._get_distance(g.loc, referent.loc) < 1.0 + 1e-3
                    if near and fn(g, referent, agent.loc, agent.yaw) == direction:
                        cells.append([int(g.loc[0]), int(g.loc[1])])
                rec["target_cells"] = cells
                rec["direction"] = direction
            else:
                rec["target_cells"] = [[int(t.loc[0]), int(t.loc[1])] for t in task.target]
            stage = "navigation_reward"
            trace, after_end = [], 0
            for t in range(steps):
                a = rnd.randrange(6 if ego else 4)
                h.act(a)
                stage, reward, event = h.py_stage(task, stage)
                trace.append([a, reward, event, stage, int(h.agent["loc"][0]), int(h.agent["loc"][1]), int(bool(h.success))]
                             + ([float(h.agent["yaw"])] if ego else []))
                if stage == "terminal":
                    after_end += 1
                    if after_end > 2:
                        break
            rec["trace"] = trace
            runs.append(rec)
        out[name] = runs
    return out


# ------------------------------------- the 2-D-native group (rule D14b) ----
class _ItDict(dict):
    """dict with the Python-2 iteritems() that XWorldTask._get_surrounding_empty_grids calls."""

    def iteritems(self):
        return iter(self.items())


class _Py2Int(int):
    """int whose `*` stays a _Py2Int and whose `/` floors, as Python 2's int / int does
    (xworld_task.py:205: `self.steps_in_cur_task >= h*w / 2`)."""

    def __mul__(self, o):
        return _Py2Int(int(self) * int(o))

    def __truediv__(self, o):
        return _Py2Int(int(self) // int(o))


def gen_tasks2d(pals, n_maps, seed0, steps):
    """games/xworld/tasks/XWorldNav{Target,Near,ColorTarget,Between}.py (confs/walls.json group "XWorldNav")
    run as a one-task TaskGroup (teaching_task.cpp:204-222: when the task is idle, Task::reset then its stage).
    Per teach() call: was the task idle, the decisions its idle stage drew, reward, event, stage, target cell."""
    import importlib
    sys.path.insert(0, os.path.join(REF, "games", "xworld", "tasks"))
    names = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]
    out = {}
    rnd = random.Random(777)
    for mode in ("lang_acquisition", "one_channel"):
        FLAGS["task_mode"] = mode
        for key, env_cls in (("nav", XWorldNav), ("walls", XWorldWalls)):
            pal = pals[key]
            env = env_cls(ITEM_PATH)
            for name in names:
                mod = importlib.import_module(name)
                cls = getattr(mod, name)
                runs = []
                for k in range(n_maps):
                    random.seed(seed0 + k)
                    env.reset()
                    env.env_changed()
                    ents = entity_records(env, pal)
                    h = Harness(env)
                    mh, mw = env.get_max_dims()
                    env.get_max_dims = lambda _h=mh, _w=mw: (_Py2Int(_h), _Py2Int(_w))
                    task = cls(env)
                    task.directions = _ItDict(task.directions)
                    fake = DecisionRandom(seed0 * 11 + k)
                    real = mod.random
                    mod.random = fake
                    state = {"stage": "idle"}

                    def teach():
                        was_idle = state["stage"] == "idle"
                        if was_idle:
                            task.reset()
                        n0 = len(fake.log)
                        stage, reward, event = h.py_stage(task, state["stage"])
                        state["stage"] = stage
                        tgt = task.target
                        tx, ty = (int(tgt[0]) + env.offset_w, int(tgt[1]) + env.offset_h) if tgt[0] >= 0 else (-1, -1)
                        return [int(was_idle), list(fake.log[n0:]), reward, event, stage, tx, ty]
                    try:
                        first = teach()
                        trace = []
                        for t in range(steps):
                            a = rnd.randrange(4)
                            h.act(a)
                            rec = teach()
                            trace.append([a, int(h.agent["loc"][0]), int(h.agent["loc"][1]), int(bool(h.success))] + rec)
                    finally:
                        mod.random = real
                        del env.get_max_dims
                    runs.append({"py_seed": seed0 + k, "dim": env.get_dims()[0], "max_dim": mh, "entities": ents,
                                 "reset_teach": first, "trace": trace})
                out["%s/%s/%s" % (mode, key, name)] = runs
    FLAGS["task_mode"] = "lang_acquisition"
    return out



# ------------------------------------------- two task groups, non-exclusive (D13) ----
def gen_groups(pal, n_maps, seed0, steps):
    """Teacher::teach with TWO task groups and task_groups_exclusive = false (teacher.cpp:207-230; lang_acquisition forces it,
    simulator_interface.cpp:46-48): a one-task XWorld3DNav* group and a one-task XWorldNav* (2-D-native) group, in both conf
    orders.  The ~15 lines of teacher.cpp / teaching_task.cpp restated here: per teach() every group in conf order -- if its
    task is idle: Task::reset, then the current stage through Task::py_stage (Harness.py_stage: the FIRST group's call consumes
    the step's collision events, get_events_of_game clears them; every call overwrites the event buffer, "" included);
    rewards add up.  Logged per teach() and group: was it idle, the decisions its stage drew, reward, event, next stage; plus
    the 2-D task's target cell, the summed reward and the event left in the buffer."""
    import importlib
    sys.path.insert(0, os.path.join(REF, "games", "xworld", "tasks"))
    pairs = [("XWorld3DNavTarget", "XWorldNavTarget"), ("XWorld3DNavTargetNear", "XWorldNavColorTarget"),
             ("XWorld3DNavTargetBetween", "XWorldNavTarget"), ("XWorld3DNavTargetDirection", "XWorldNavNear"),
             ("XWorld3DNavTargetAvoid", "XWorldNavColorTarget")]
    out = {}
    rnd = random.Random(1357)
    env = XWorldNav(ITEM_PATH)
    for n3, n2 in pairs:
        m3, m2 = importlib.import_module(n3), importlib.import_module(n2)
        for order in ("3d_first", "2d_first"):
            runs = []
            k = 0
            while len(runs) < n_maps and k < 6 * n_maps:
                k += 1
                random.seed(seed0 + k)
                env.reset()
                env.env_changed()
                before = entity_records(env, pal)
                h = Harness(env)
                t3, t2 = getattr(m3, n3)(env), getattr(m2, n2)(env)
                t2.directions = _ItDict(t2.directions)
                fake = DecisionRandom(seed0 * 13 + k)
                real3, real2 = m3.random, m2.random
                m3.random = m2.random = fake
                groups = [("3d", t3), ("2d", t2)] if order == "3d_first" else [("2d", t2), ("3d", t3)]
                stage = {"3d": "idle", "2d": "idle"}

                def teach():
                    recs, total, buffer_event = [], 0.0, ""
                    for fam, task in groups:
                        was_idle = stage[fam] == "idle"
                        if was_idle:
                            task.reset()
                        n0 = len(fake.log)
                        # Task::py_stage, teaching_task.cpp:64-116 (Harness.py_stage reads env_changed() itself, which
                        # clears the flag: restated here so that update_environment() below still sees it)
                        env.update_entities_from_cpp([dict(e) for e in h.ents])
                        env.update_agent_sentence_from_cpp("")
                        env.update_agent_action_success_from_cpp(h.success)
                        ev, h.game_events = h.game_events, ""
                        env.update_game_event_from_cpp(ev)
                        ret = getattr(task, stage[fam])()
                        st, reward = ret[0], float(ret[1])
                        stage[fam] = st
                        changed = env.env_changed()
                        event = task.get_event()
                        if changed:                              # Task::py_stage -> game_->update_environment()
                            h.ents = [dict(e) for e in env.cpp_get_entities()]
                            for e in h.ents:
                                e["loc"] = tuple(int(v) for v in e["loc"])
                            h.agent = [e for e in h.ents if e["type"] == "agent"][0]
                        total += reward
                        buffer_event = event
                        recs.append([fam, int(was_idle), list(fake.log[n0:]), reward, event, st])
                    tgt = t2.target
                    tx, ty = (int(tgt[0]) + env.offset_w, int(tgt[1]) + env.offset_h) if tgt[0] >= 0 else (-1, -1)
                    return {"groups": recs, "reward": total, "event": buffer_event, "target2d": [tx, ty]}
                try:
                    try:
                        first = teach()
                    except AssertionError:                       # "map too crowded?": the reference process would die
                        continue
                    after = entity_records(env, pal)
                    trace = []
                    for t in range(steps):
                        a = rnd.randrange(4)
                        h.act(a)
                        rec = teach()
                        rec.update(action=a, agent=[int(h.agent["loc"][0]), int(h.agent["loc"][1])], success=int(bool(h.success)))
                        trace.append(rec)
                finally:
                    m3.random, m2.random = real3, real2
                runs.append({"py_seed": seed0 + k, "dim": env.get_dims()[0], "max_dim": env.get_max_dims()[0],
                             "entities_before": before, "entities_after": after, "reset_teach": first, "trace": trace})
            out["%s+%s/%s" % (n3, n2, order)] = runs
    return out


# ------------------------------------------- two task groups, EXCLUSIVE scheduling (D13) ----
def gen_groups_exclusive(pal, n_maps, seed0, steps):
    """Teacher::teach with TWO task groups and task_groups_exclusive = true (teacher.cpp:207-220; py_simulator's default,
    in force whenever task_mode is not lang_acquisition): a one-task XWorld3DNav* group and a one-task XWorldNav*
    (2-D-native) group, both conf orders, several group weights, task_mode one_channel.  The lines of teacher.cpp restated
    here: per teach() nondeterministic_sort_task_groups (:143-163) re-sorts the group list IN PLACE by weighted sampling
    without replacement -- one util::simple_importance_sampling draw per position, the last one over a single weight --, then
    the LAST non-idle group of that order runs its stage, else the first (:209-220; the loop has no break).  The sampling
    decisions (index into the remaining groups) are drawn here with the same rule (uniform value in [0, total), first
    accumulated weight >= it) and logged; a group's stage runs as in gen_groups.  A 3-D group that is picked while idle in
    mid-episode runs its map-rearranging idle stage at step time: the entity list after every such call is recorded."""
    import importlib
    sys.path.insert(0, os.path.join(REF, "games", "xworld", "tasks"))
    pairs = [("XWorld3DNavTarget", "XWorldNavTarget"), ("XWorld3DNavTargetNear", "XWorldNavColorTarget"),
             ("XWorld3DNavTargetBetween", "XWorldNavTarget"), ("XWorld3DNavTargetDirection", "XWorldNavColorTarget"),
             ("XWorld3DNavTargetAvoid", "XWorldNavNear")]
    weights = {"3d_first": [(1.0, 1.0), (0.5, 2.0)], "2d_first": [(1.0, 1.0), (3.0, 1.0)]}    # (first, second) in conf order
    out = {}
    rnd = random.Random(2468)
    FLAGS["task_mode"] = "one_channel"
    env = XWorldNav(ITEM_PATH)
    mh, mw = env.get_max_dims()
    try:
        for n3, n2 in pairs:
            m3, m2 = importlib.import_module(n3), importlib.import_module(n2)
            for order in ("3d_first", "2d_first"):
                for wts in weights[order]:
                    runs = []
                    k = 0
                    while len(runs) < n_maps and k < 8 * n_maps:
                        k += 1
                        random.seed(seed0 + k + 100 * len(out))
                        env.reset()
                        env.env_changed()
                        before = entity_records(env, pal)
                        h = Harness(env)
                        env.get_max_dims = lambda _h=mh, _w=mw: (_Py2Int(_h), _Py2Int(_w))
                        t3, t2 = getattr(m3, n3)(env), getattr(m2, n2)(env)
                        t2.directions = _ItDict(t2.directions)
                        fake = DecisionRandom(seed0 * 17 + k + 1000 * len(out))
                        grnd = random.Random(seed0 * 19 + k + 1000 * len(out))
                        real3, real2 = m3.random, m2.random
                        m3.random = m2.random = fake
                        task = {"3d": t3, "2d": t2}
                        # Teacher::task_groups_ / task_group_weights_: conf order, then whatever the sorts leave
                        glist = ["3d", "2d"] if order == "3d_first" else ["2d", "3d"]
                        gw = list(wts)
                        stage = {"3d": "idle", "2d": "idle"}

                        def teach():
                            draws = []
                            for i in range(len(glist)):                 # nondeterministic_sort_task_groups
                                acc, tot = [], 0.0
                                for x in gw[i:]:
                                    tot += x
                                    acc.append(tot)
                                val = grnd.random() * tot
                                idx = [j for j, a in enumerate(acc) if val <= a][0]
                                draws.append(idx)
                                glist[i], glist[i + idx] = glist[i + idx], glist[i]
                                gw[i], gw[i + idx] = gw[i + idx], gw[i]
                            busy = None
                            for fam in glist:                           # no break: the last busy group wins
                                if stage[fam] != "idle":
                                    busy = fam
                            fam = busy if busy is not None else glist[0]
                            was_idle = stage[fam] == "idle"
                            if was_idle:
                                task[fam].reset()
                            n0 = len(fake.log)
                            env.update_entities_from_cpp([dict(e) for e in h.ents])
                            env.update_agent_sentence_from_cpp("")
                            env.update_agent_action_success_from_cpp(h.success)
                            ev, h.game_events = h.game_events, ""
                            env.update_game_event_from_cpp(ev)
                            ret = getattr(task[fam], stage[fam])()
                            st, reward = ret[0], float(ret[1])
                            stage[fam] = st
                            changed = env.env_changed()
                            event = task[fam].get_event()
                            # compact record (GX_FIELDS): the sort's draws, is the 3-D group first afterwards, did the 3-D group
                            # run, was it idle, its idle stage's decisions, reward, event, stage, both groups' stages, the
                            # 2-D task's target cell, the entity list when the stage changed the map (else None)
                            rec = [draws[0], draws[1], int(glist[0] == "3d"), int(fam == "3d"), int(was_idle), list(fake.log[n0:]),
                                   reward, event, st, stage["3d"], stage["2d"]]
                            ents_after = None
                            if changed:                                  # Task::py_stage -> game_->update_environment()
                                h.ents = [dict(e) for e in env.cpp_get_entities()]
                                for e in h.ents:
                                    e["loc"] = tuple(int(v) for v in e["loc"])
                                h.agent = [e for e in h.ents if e["type"] == "agent"][0]
                                ents_after = entity_records(env, pal)
                            tgt = t2.target
                            rec += [int(tgt[0]) + env.offset_w, int(tgt[1]) + env.offset_h] if tgt[0] >= 0 else [-1, -1]
                            rec.append(ents_after)
                            return rec
                        ok = True
                        try:
                            first = teach()
                            trace = []
                            for t in range(steps):
                                a = rnd.randrange(4)
                                h.act(a)
                                rec = teach()
                                trace.append(rec + [a, int(h.agent["loc"][0]), int(h.agent["loc"][1]), int(bool(h.success))])
                        except AssertionError:                           # "map too crowded?": the reference process would die
                            ok = False
                        finally:
                            m3.random, m2.random = real3, real2
                            del env.get_max_dims
                        if ok:
                            runs.append({"py_seed": seed0 + k + 100 * len(out), "dim": env.get_dims()[0], "max_dim": mh, "weights": list(wts),
                                         "entities_before": before, "reset_teach": first, "trace": trace})
                    out["%s+%s/%s/%g:%g" % (n3, n2, order, wts[0], wts[1])] = runs
    finally:
        FLAGS["task_mode"] = "lang_acquisition"
    return out


# ------------------------------------------------------------ teacher sentences ----
def gen_sentences(n_per_task, seed0):
    """The reference's context_free_grammar.CFG (the real module, not the no-op stand-in above) fed with each task's own
    _define_grammar(): sentences for random bindings, with the random.choice decisions that produced them."""
    import importlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("real_cfg", os.path.join(REF, "python", "context_free_grammar.py"))
    real = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(real)
    real.CFG._CFG__unbind_all = lambda self: [rhs.unbind() for rhs in self.productions.values()]   # py2 iteritems
    env = XWorldNav(ITEM_PATH)
    random.seed(seed0)
    env.reset()
    names = sorted(_orig_names(env, "goal"))
    rnd = random.Random(seed0)
    out = {"goal_names": names, "tasks": {}}
    sys.path.insert(0, os.path.join(REF, "games", "xworld", "tasks"))
    colors = sorted(set(env.get_all_colors()))
    out["colors"] = colors
    for name in ["XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection",
                 "XWorld3DNavTargetAvoid", "XWorldNavTarget", "XWorldNavColorTarget"]:
        mod = importlib.import_module(name)
        task = getattr(mod, name)(env)
        grammar, start = task._define_grammar()
        cfg = real.CFG(grammar, start)
        recs = []
        for k in range(n_per_task):
            what = ["start", "correct", "wrong", "timeup"][0 if k % 4 else rnd.randrange(1, 4)] if k % 7 == 0 else "start"
            if not name.startswith("XWorld3D") and what in ("correct", "wrong"):
                what = "finish"
            binds = {"S": what}
            if what == "start":
                if name == "XWorldNavColorTarget":
                    binds["O"] = "'%s'" % rnd.choice(names)
                    binds["C"] = "'%s'" % rnd.choice(colors)
                elif name == "XWorld3DNavTargetBetween":
                    binds["G1"] = "'%s'" % rnd.choice(names)
                    binds["G2"] = "'%s'" % rnd.choice(names)
                else:
                    binds["G"] = "'%s'" % rnd.choice(names)
                if name == "XWorld3DNavTargetDirection":
                    binds["P"] = rnd.choice(["LEFT", "RIGHT", "FRONT", "BEHIND"])
            fake = DecisionRandom(seed0 * 3 + k)
            real.random = fake
            try:
                for lhs, rhs in binds.items():
                    cfg.bind("%s -> %s" % (lhs, rhs))
                sent = cfg.generate()
            finally:
                real.random = random
            recs.append({"bind": binds, "decisions": list(fake.log), "sentence": sent})
        out["tasks"][name] = recs
    return out


# ------------------------------------------------------------ curriculum ----
def gen_curriculum(pal, n_episodes, seed0, threshold):
    """FLAGS_curriculum != 0 (XWorldNav.py:36-53, xworld_env.py:103-110, xworld3d_task.py:129-146): ONE XWorldNav env and
    one object per task class live through `n_episodes` resets, as inside a reference process.  A scripted agent (walk
    to the cell above a target goal and bump into it, or wander) makes the success rate swing; per episode: the level,
    dims and entity counts the reset chose, and every result the busy task recorded, next to the event of that step."""
    import importlib
    from collections import deque
    names = ["XWorld3DNavTarget", "XWorld3DNavTargetBetween", "XWorld3DNavTargetAvoid"]   # the ones this script can solve
    FLAGS["curriculum"] = threshold
    try:
        env = XWorldNav(ITEM_PATH)
        tasks, episodes = {}, []
        rnd = random.Random(777)
        for k in range(n_episodes):
            random.seed(seed0 + k)
            env.reset()
            env.env_changed()
            rec = {"level": int(env.dump_curriculum_progress()), "dim": int(env.get_dims()[0]),
                   "num_goals": len(env.get_goals()), "num_blocks": len(env.get_blocks()),
                   "counter": int(env.curriculum_check_counter), "records": [], "events": []}
            episodes.append(rec)
            name = names[rnd.randrange(len(names))]
            rec["task"] = name
            mod = importlib.import_module(name)
            if name not in tasks:
                tasks[name] = getattr(mod, name)(env)
            task = tasks[name]
            task.reset()
            h = Harness(env)
            fake = DecisionRandom(seed0 * 11 + k)
            real = mod.random
            mod.random = fake
            try:
                h.env.update_entities_from_cpp([dict(e) for e in h.ents])
                h.env.update_agent_sentence_from_cpp("")
                h.env.update_agent_action_success_from_cpp(False)
                h.env.update_game_event_from_cpp("")
                try:
                    ret = task.idle()
                except AssertionError:                       # "map too crowded?": no task this episode
                    continue
            finally:
                mod.random = real
            task.get_event()
            if env.env_changed():
                h.ents = [dict(e) for e in env.cpp_get_entities()]
                for e in h.ents:
                    e["loc"] = tuple(int(v) for v in e["loc"])
                h.agent = [e for e in h.ents if e["type"] == "agent"][0]
            # where the scripted agent wants to stand, and the move that ends the task from there
            if name == "XWorld3DNavTargetBetween":
                l1, l2 = task.target
                want, last = [(int((l1[0] + l2[0]) // 2) + env.offset_w, int((l1[1] + l2[1]) // 2) + env.offset_h)], None
            elif name == "XWorld3DNavTargetDirection":
                referent, direction = task.target
                agent = [e for e in env.get_entities() if e.type == "agent"][0]
                fn = getattr(task, "_XWorld3DNavTargetDirection__compute_triple_direction")
                want = [(int(g.loc[0]) + env.offset_w, int(g.loc[1]) + env.offset_h - 1) for g in env.get_goals()
                        if task._get_distance(g.loc, referent.loc) < 1.0 + 1e-3 and fn(g, referent, agent.loc, agent.yaw) == direction]
                last = 1
            else:
                want, last = [(int(t.loc[0]) + env.offset_w, int(t.loc[1]) + env.offset_h - 1) for t in task.target], 1
            skilled = rnd.random() < (0.05 if 20 <= k < 200 else 0.97)
            stage = "navigation_reward"
            total = task.num_successes + task.num_failures
            for t in range(h.H * h.W * 10 + 5):
                a = rnd.randrange(4)
                if skilled:
                    start = (h.agent["loc"][0], h.agent["loc"][1])
                    prev, todo = {start: None}, deque([start])
                    while todo:                                  # BFS over empty cells
                        c = todo.popleft()
                        for act, (dx, dy) in enumerate([(0, -1), (0, 1), (-1, 0), (1, 0)]):
                            n = (c[0] + dx, c[1] + dy)
                            if n in prev or not (0 <= n[0] < h.W and 0 <= n[1] < h.H) or h.cell(n[0], n[1]):
                                continue
                            prev[n] = (c, act)
                            todo.append(n)
                    goal = [w for w in want if w in prev]
                    if goal:
                        c = goal[0]
                        if c == start:
                            a = last if last is not None else a
                        else:
                            while prev[c][0] != start:
                                c = prev[c][0]
                            a = prev[c][1]
                h.act(a)
                stage, reward, event = h.py_stage(task, stage)
                now = task.num_successes + task.num_failures
                if now != total or event:
                    rec["events"].append([event, int(task.success_seq[-1]) if now != total else -1])
                if now != total:
                    assert now == total + 1
                    rec["records"].append(int(task.success_seq[-1]))
                    total = now
                if stage == "terminal":
                    break
        return {"threshold": threshold, "tasks": names, "episodes": episodes}
    finally:
        FLAGS["curriculum"] = 0.0


def main():
    nav_pal = O.Palette(O.NAV_SUBTREES)
    walls_pal = O.Palette(O.WALLS_SUBTREES)
    makers = {
        "maze.json": lambda: gen_maze(),
        "bfs.json": lambda: gen_bfs(),
        "maps_nav.json": lambda: gen_maps(XWorldNav, nav_pal, 60, 100),
        "maps_walls.json": lambda: gen_maps(XWorldWalls, walls_pal, 30, 500),
        "maps_levels.json": lambda: gen_maps_levels(nav_pal, 8, 700),
        "teacher.json": lambda: {"nav": gen_teacher(XWorldNav, nav_pal, 40, 2000, 700),
                                 "walls": gen_teacher(XWorldWalls, walls_pal, 24, 3000, 500)},
        "tasks.json": lambda: gen_tasks(nav_pal, 24, 9000, 660),
        "tasks_ego.json": lambda: gen_tasks(nav_pal, 16, 15000, 900, ego=3),
        "sentences.json": lambda: gen_sentences(60, 31000),
        "tasks2d.json": lambda: gen_tasks2d({"nav": nav_pal, "walls": walls_pal}, 6, 12000, 70),
        "curriculum.json": lambda: gen_curriculum(nav_pal, 920, 41000, 0.33),
        "groups.json": lambda: gen_groups(nav_pal, 6, 52000, 60),
        "groups_exclusive.json": lambda: gen_groups_exclusive(nav_pal, 3, 63000, 100),
    }
    only = sys.argv[1:]                      # optional: the fixtures to (re)generate
    out = {name: make() for name, make in makers.items() if not only or name in only}
    for name, data in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
