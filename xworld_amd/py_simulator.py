"""Drop-in for the reference's Boost.Python module `py_simulator` (python/py_simulator.cpp:310-329).

    from xworld_amd.py_simulator import Simulator
    game = Simulator.create("simple_game", {"array_size": 6})

Same class and method names and return shapes as the reference.  With the default
num_envs == 1 the scalar signatures hold exactly (`take_actions(dict, act_rep, show)` -> float,
`get_state()` -> dict of Python float lists with uint8 pixels scaled by 1/255, `game_over()` -> str).
`create(name, opts)` additionally accepts "num_envs" (and "device", "seed", "env_gid0") in opts;
a batch exposes the tensor API through `.batch` (xworld_amd.batched.BatchedSimulator).
"""
import os

import numpy as np

from . import assets

from . import lib
from .batched import BatchedSimulator


def help():
    # py_simulator.cpp:299-307
    print("Games are created by calling Simulator.create()\n"
          "Example:\n"
          "  from py_simulator import Simulator\n"
          "  simple_game = Simulator.create(\"simple_game\", {\"array_size\":6})\n"
          "For more help on how to create games, call Simulator.help()")


class Simulator:
    def __init__(self):
        raise RuntimeError("This class cannot be instantiated from Python")   # py::no_init

    @staticmethod
    def create(name, opts):
        self = object.__new__(Simulator)
        opts = dict(opts)
        kw = {k: opts.pop(k) for k in ("num_envs", "device", "seed", "env_gid0", "policy_seed") if k in opts}
        self.batch = BatchedSimulator(name, opts, **kw)
        self._env = 0
        self._last_reward = 0.0
        return self

    # --- the reference surface -------------------------------------------------
    def reset_game(self):
        self.batch.reset()

    def game_over(self):
        # "alive" | "max_step|dead|success|lost_life" (simulator.cpp:125-144)
        return lib.decode_game_over_code(self.batch.env_state(self._env).game_over)

    def get_num_actions(self):
        return self.batch.num_actions

    def get_lives(self):
        return self.batch.env_state(self._env).lives

    def get_screen_out_dimensions(self):
        h, w, c = self.batch.screen_dims
        return [h, w, c, self.batch.cfg.context]                    # py_simulator.cpp:287-297

    def take_actions(self, actions, act_rep=1, show_screen=False):
        import torch
        if len(actions) == 0:
            raise RuntimeError("You can't take an empty action")     # py_simulator.cpp:199 CHECK_GT
        a = int(actions.get("action", 0))                            # default action id 0, py_simulator.cpp:195
        if self.batch.name == "xworld" and self.batch.cfg.task_mode == 1 and "pred_sentence" not in actions:
            raise RuntimeError("The agent has to take the speak action.")   # xworld_simulator.cpp:211-212
        if a < 0 or a >= self.batch.num_actions:
            raise RuntimeError("action invalid: %d" % a)             # CHECK_LT -> abort in the reference
        t = torch.full((self.batch.num_envs,), a, dtype=torch.int32, device="cuda:%d" % self.batch.device)
        self.batch.step(t, act_rep)
        self._last_reward = float(self.batch.env_state(self._env).reward)
        return self._last_reward

    def take_action(self, actions, show_screen=False):
        return self.take_actions(actions, 1, show_screen)

    def get_state(self):
        # py_simulator.cpp:246-285: every non-"reward", non-"sentence" key -> list of floats, uint8 x 1/255
        d = {}
        obs = self.batch.env_obs(self._env)
        if obs.dtype.kind == "u":
            # `float scale = 1 / 255.0; l.append(x * scale)` with x a float: a float32 product (py_simulator.cpp:264-272)
            d["screen"] = (obs.astype(np.float32) * np.float32(1 / 255.0)).tolist()
        else:
            d["screen"] = [float(x) for x in obs]
        if self.batch.name == "xworld":
            st = self.batch.env_state(self._env)
            d["sentence"] = self.batch.sentence(self._env) or "-"     # get_teacher_sentence_for_agent: "" -> "-"
            events = {0: "", 1: "correct_goal", 2: "wrong_goal", 3: "time_up"}
            d["task"] = assets.TASK_NAMES[st.xw_task]
            d["event"] = events[st.xw_event]
            # get_extra_info, xworld_simulator.cpp:495-504: xworld_.actual_height() / actual_width() = the level's
            # dims (3 + level) under FLAGS_curriculum
            dim = 3 + st.xw_level if self.batch.cfg.curriculum != 0 and self.batch.cfg.map_kind == 0 else self.batch.cfg.dim
            d["height"] = str(dim)
            d["width"] = str(dim)
        return d

    def get_num_steps(self):
        return int(self.batch.env_state(self._env).num_steps)
