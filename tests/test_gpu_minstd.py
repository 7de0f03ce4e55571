"""rng = "minstd" on the GPU: every env owns the reference's thread-local minstd_rand0 (simulator_util.cpp:38-73, seeded from
FLAGS_simulator_seed and the thread number) and the decisions the reference takes with util::get_rand_range_val /
get_rand_ind come from it -- SimpleRace's random reset (simple_race_simulator.cpp:86-89,196-199,237-243,267-284) and the
teacher's task draw (teaching_task.cpp:204-213) -- against the oracle's engines (oracle/rng.c, pinned by the reference's
tests/test_simulator_seed.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("sim_seed,track,full", [(1, "straight", False), (2, "circle", True), (77, "straight", True)])
def test_simple_race_replays_a_seeded_reference_run(oracle, sim_seed, track, full):
    """random = true, FLAGS_simulator_seed = 1 / 2 / 77: env e is the reference's (e + 1 + thread_base)-th simulator thread;
    start positions, angles, every reward bit, code and observation of a rollout with resets."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps, base, gid0 = 96, 300, 3, 10
    opts = {"track_type": track, "track_width": 20.0, "track_length": 100.0, "track_radius": 30.0, "random": True,
            "race_full_manouver": full, "rng": "minstd", "simulator_seed": sim_seed, "thread_base": base}
    sim = BatchedSimulator("simple_race", opts, num_envs=n, env_gid0=gid0, policy_seed=4)
    envs = []
    for e in range(n):
        # the constructor resets once (SimpleRaceGame ctor, cpp:457): xwb_create did the same
        envs.append(oracle.SimpleRace(track_type=1 if track == "circle" else 0, random=1, race_full_manouver=int(full),
                                      simulator_seed=sim_seed, nth_thread=base + gid0 + e + 1))
    L = oracle.lib()
    for e in (0, 5, n - 1):                                # engine states after the constructor's four draws
        g = oracle.MinStd()
        L.orc_minstd_seed_thread(C.byref(g), sim_seed, base + gid0 + e + 1)
        for _ in range(4):
            L.orc_minstd_next(C.byref(g))
        assert int(sim.minstd_state[e]) & 0xffffffff == g.x
    sim.reset()                                             # SimulatorInterface::reset_game: four more draws, init_screen
    for g in envs:
        g.reset_game()
    resets = 0
    for t in range(steps):
        codes = sim.game_over_codes.cpu().numpy()
        sim.reset_done()
        for e, g in enumerate(envs):
            if g.game_over() != 0:
                assert codes[e] != 0
                g.reset_game()
                resets += 1
        obs = sim.obs.cpu().numpy().reshape(n, 4)
        cars = [g.state_screen() for g in envs]
        assert np.array_equal(obs.view(np.uint32), np.array(cars, np.float32).reshape(n, 4).view(np.uint32)), t
        sim.step()
        acts = sim.actions.cpu().numpy()
        rew = np.array([np.float32(g.take_actions(int(acts[e]))) for e, g in enumerate(envs)], np.float32)
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), rew.view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), np.array([g.game_over() for g in envs], np.uint8)), t
    assert resets > n
    # checkpoint carries the engines
    blob = sim.save_state()
    a = sim.minstd_state.clone()
    for _ in range(50):
        sim.step_autoreset()
    assert not torch.equal(a, sim.minstd_state)
    sim.load_state(blob)
    assert torch.equal(a, sim.minstd_state)
    sim.close()
    with pytest.raises(Exception):
        BatchedSimulator("simple_race", dict(opts, simulator_seed=0), num_envs=4)


@pytest.mark.parametrize("weights", [None, [1, 2, 3, 4, 5]])
def test_xworld_task_draw_from_the_reference_engine(oracle, weights):
    """The teacher's task draw comes from the env's minstd engine; the map (xwb-rng-v1) is the one the default mode makes."""
    _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 512, 300
    opts = {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition", "max_dim": 7,
            "num_blocks": 16}
    if weights:
        opts["task_weights"] = weights
    a = BatchedSimulator("xworld", dict(opts, rng="minstd", simulator_seed=2, thread_base=5), num_envs=n, seed=9, policy_seed=3,
                         env_gid0=20)
    b = BatchedSimulator("xworld", opts, num_envs=n, seed=9, policy_seed=3, env_gid0=20)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(map_kind=0, max_dim=7, dim=7, num_goals=4, num_blocks=16, seed=9, tasks=[0, 1, 2, 3, 4], simulator_seed=2,
               thread_base=5)
    if weights:
        cfg["task_weights"] = weights
    ow = oracle.XWorld(pal, render=False, **cfg)
    kinds_a, kinds_b = [], []
    for e in range(n):
        ow.reset_game(20 + e, 0)
        st = a.env_state(e)
        assert st.xw_task == ow.task_kind(), e
        kinds_a.append(st.xw_task)
        kinds_b.append(b.env_state(e).xw_task)
        if st.xw_task == kinds_b[-1] and st.xw_task in (0, 4):          # same task, no map rearrangement: the same map
            assert np.array_equal(a.env_grid(e), b.env_grid(e)), e
    assert kinds_a != kinds_b and len(set(kinds_a)) == 5
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=3, env_gid0=20)
    for t in range(steps):
        a.reset_done()
        a.step()
        assert np.array_equal(a.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(a.game_over_codes.cpu().numpy(), ref.codes[t]), t
    assert ref.stats.resets > 100
    a.close()
    b.close()
