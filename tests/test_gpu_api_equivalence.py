"""The batched verbs agree with each other in every mode: step + reset_done == step + reset_masked(done mask) ==
step + reset_env(each finished env) == step_autoreset (frames, rewards, codes, counters), also with caller-chosen
actions, skipped envs and a caller stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")
NAV = os.path.join(CONF, "navigation2d.json")
MODES = {
    "full": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "dim": 7, "color": True, "context": 2},
    "gray": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition"},
    "ego": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "dim": 7, "color": True, "visible_radius": 3, "context": 2},
    "curriculum": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "tasks": ["XWorld3DNavTarget"], "curriculum": 0.05, "color": True},
    "walls2d": {"xwd_conf_path": os.path.join(CONF, "walls.json"), "task_mode": "one_channel", "max_steps": 40, "color": True},
    "f32": {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "dim": 7, "color": True, "obs_format": "float32"},
}


def _same(torch, a, b, what):
    assert torch.equal(a.reward, b.reward), what
    assert torch.equal(a.game_over_codes, b.game_over_codes), what
    assert torch.equal(a.num_steps, b.num_steps), what
    assert torch.equal(a.obs, b.obs), what


@pytest.mark.parametrize("mode", sorted(MODES))
def test_reset_verbs_agree(mode):
    import torch
    assert torch.cuda.is_available()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 768, 260 if mode != "curriculum" else 1400
    sims = [BatchedSimulator("xworld", MODES[mode], num_envs=n, seed=9, policy_seed=4) for _ in range(4)]
    done_list, masked, single, fused = sims
    side = torch.cuda.Stream()
    own = torch.cuda.Stream()                                # the reference verbs of this test run on a caller stream too
    rng = np.random.default_rng(1)
    n_act = done_list.num_actions
    for t in range(steps):
        acts = None
        if t % 3 == 1:                                   # caller-chosen actions, a few envs sit the step out
            a = rng.integers(0, n_act, n).astype(np.int32)
            a[rng.integers(0, n, 7)] = -1
            acts = torch.from_numpy(a).cuda()
        with torch.cuda.stream(own):
            done_list.step(acts, stream=own)
            codes_after_step = done_list.game_over_codes.clone()     # queued behind the render, read before reset_done clears
            done_list.reset_done(stream=own)
        own.synchronize()
        masked.step(acts)
        assert torch.equal(codes_after_step, masked.game_over_codes), (mode, t, "codes survive until reset_done")
        masked.reset_masked(masked.game_over_codes != 0)
        if t % 4 == 0:                                   # the slow verb, now and then on every finished env
            single.step(acts)
            for e in torch.nonzero(single.game_over_codes != 0).flatten().tolist():
                single.reset_env(e)
        else:
            single.step(acts)
            single.reset_done()
        with torch.cuda.stream(side):                    # the fused verb on a caller stream
            fused.step_autoreset(acts, stream=side)
        side.synchronize()
        _same(torch, done_list, masked, (mode, t, "masked"))
        _same(torch, done_list, single, (mode, t, "reset_env"))
        # step_autoreset never shows the terminal frame and reports the step's codes / rewards all the same
        assert torch.equal(done_list.obs, fused.obs), (mode, t, "autoreset obs")
        assert torch.equal(done_list.num_steps, fused.num_steps), (mode, t)
    assert int(done_list.episode.max()) >= 2
    for s in sims:
        s.close()


@pytest.mark.parametrize("game,opts,n", [("simple_game", {"array_size": 16}, 3000), ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}, 3000),
                                         ("xworld", MODES["gray"], 1024), ("xworld", MODES["ego"], 512)])
@pytest.mark.parametrize("autoreset", [False, True])
def test_run_equals_the_separate_calls(game, opts, n, autoreset):
    """xwb_run(k) = k x (xwb_step; xwb_reset_done) -- or k x xwb_step_autoreset -- issued from C (the reference example loop,
    examples/test_simple_race.cpp:26-53): same state, frames, results ring and policy draws as the separate calls."""
    import torch
    assert torch.cuda.is_available()
    from xworld_amd.batched import BatchedSimulator
    a = BatchedSimulator(game, opts, num_envs=n, seed=5, policy_seed=6)
    b = BatchedSimulator(game, opts, num_envs=n, seed=5, policy_seed=6)
    ra = torch.zeros((64, n, 2), dtype=torch.float32, device="cuda")
    rb = torch.zeros_like(ra)
    a.bind_results_ring(ra); b.bind_results_ring(rb)
    done = 0
    for k in (1, 7, 20, 3, 33):
        a.run(k, autoreset=autoreset)
        for _ in range(k):
            if autoreset:
                b.step_autoreset()
            else:
                b.step()
                done += int((b.game_over_codes != 0).sum())
                b.reset_done()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes), k
        assert torch.equal(a.num_steps, b.num_steps) and torch.equal(a.episode, b.episode) and torch.equal(a.actions, b.actions), k
        assert torch.equal(ra, rb), k
    assert autoreset or done > 0
    with pytest.raises(Exception, match="iterations"):
        a.run(0)
    assert a.check_errors() == 0
    a.close(); b.close()
