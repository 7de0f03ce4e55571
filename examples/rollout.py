#!/usr/bin/env python3
"""Random-policy rollouts, the shape of the reference's python/examples/test_*.py -- once through the
py_simulator-compatible object (one env, Python dicts), once through the batch API (device tensors).

    python examples/rollout.py [simple_game|simple_race|xworld|xworld_ego]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from xworld_amd.batched import BatchedSimulator      # noqa: E402
from xworld_amd.py_simulator import Simulator        # noqa: E402

CONF = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")
OPTS = {
    "simple_game": ("simple_game", {"array_size": 16}),
    "simple_race": ("simple_race", {"track_type": "straight", "track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}),
    "xworld": ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "color": True}),
    "xworld_ego": ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "color": True, "visible_radius": 3}),
}


def single_env(name, opts, steps=200):
    """python/examples/test_simple_game.py:15-30 / test_xworld.py:41-60"""
    game = Simulator.create(name, opts)
    game.reset_game()
    n_actions = game.get_num_actions()
    total, episodes = 0.0, 0
    for _ in range(steps):
        if game.game_over() != "alive":
            episodes += 1
            game.reset_game()
        state = game.get_state()                     # {"screen": [floats], ("sentence": str, ...)}
        if "sentence" in state and state["sentence"] != "-" and total == 0.0:
            print("teacher:", state["sentence"])
        total += game.take_actions({"action": random.randrange(n_actions)}, 1, False)
    print("%s: 1 env, %d steps, %d episodes, reward %.2f, screen %s" % (name, steps, episodes, total,
                                                                       game.get_screen_out_dimensions()))


def batch(name, opts, num_envs=4096, steps=200):
    import torch
    sim = BatchedSimulator(name, opts, num_envs=num_envs)
    total = torch.zeros(num_envs, device="cuda")
    finished = 0
    for _ in range(steps):
        actions = torch.randint(0, sim.num_actions, (num_envs,), dtype=torch.int32, device="cuda")
        sim.step(actions)                            # obs / reward / game_over_codes are device tensors (views)
        total += sim.reward
        finished += int((sim.game_over_codes != 0).sum())
        sim.reset_done()                             # `if game_over: reset_game()` for the whole batch
    print("%s: %d envs, %d steps, %d episodes finished, mean reward %.3f, obs %s %s" % (
        name, num_envs, steps, finished, float(total.mean()), tuple(sim.obs.shape), sim.obs.dtype))
    sim.close()


if __name__ == "__main__":
    which = sys.argv[1:] or list(OPTS)
    for w in which:
        name, opts = OPTS[w]
        single_env(name, dict(opts))
        batch(name, dict(opts))
