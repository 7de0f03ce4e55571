"""GPU parity: SimpleGame / SimpleRace HIP kernels (through the C ABI) vs the CPU oracle.

Discrete state, reward bits and game_over codes must be bit-exact.  SimpleRace float state: bit-exact as well, against
an oracle whose cos / sin are the host's libm -- no arithmetic shared with the kernels (include/xwb_trig.h) -- and, through
the `trig` fixture, against the oracle switched to the kernels' own definition; mismatches are counted and reported.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = -7046029254386353131      # 0x9E3779B97F4A7C15 as int64


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.mark.parametrize("array_size,context", [(16, 1), (64, 1), (8, 2), (6, 1), (7, 3), (1, 1), (2, 1), (3, 1)])
def test_simple_game_trajectories(oracle, array_size, context):
    _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 300, 260
    sim = BatchedSimulator("simple_game", {"array_size": array_size, "context": context}, num_envs=n,
                           policy_seed=123, env_gid0=5)
    envs = [oracle.SimpleGame(array_size, 0, context) for _ in range(n)]
    for g in envs:
        g.reset_game()
    for t in range(steps):
        sim.reset_done()
        for g in envs:
            if g.game_over() != 0:
                g.reset_game()
        obs = sim.obs.cpu().numpy().reshape(n, -1)
        for e in (0, 1, n // 2, n - 1):
            assert np.array_equal(obs[e], envs[e].state_screen()), (t, e)
        sim.step()                       # built-in random policy
        acts = sim.actions.cpu().numpy()
        rew = sim.reward.cpu().numpy()
        codes = sim.game_over_codes.cpu().numpy()
        nst = sim.num_steps.cpu().numpy()
        for e, g in enumerate(envs):
            assert acts[e] == oracle.policy_action(123, 5 + e, t, 2)
            r = np.float32(g.take_actions(int(acts[e])))
            assert r.view(np.uint32) == rew[e:e + 1].view(np.uint32)[0], (t, e, r, rew[e])
            assert codes[e] == g.game_over() and nst[e] == g.num_steps(), (t, e)
    obs = sim.obs.cpu().numpy().reshape(n, -1)
    for e, g in enumerate(envs):
        assert np.array_equal(obs[e], g.state_screen())
        st = sim.env_state(e)
        assert st.sg_pos == g.pos() and st.lives == g.get_lives()
    sim.close()


def test_simple_game_kat_reference_test(oracle):
    """tests/test_simple_game_simulator.cpp:21-47 through the batched product."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("simple_game", {"array_size": 8}, num_envs=3)
    pos = 4
    a = torch.ones(3, dtype=torch.int32, device="cuda")
    for _ in range(3):
        obs = sim.obs.cpu().numpy().reshape(3, 8)
        for e in range(3):
            assert obs[e].sum() == 1 and obs[e][pos] == 1
        sim.step(a)
        pos += 1
        r = sim.reward.cpu().numpy()
        assert np.allclose(r, 2.0 if pos == 7 else -0.1, atol=1e-6)
    sim.close()


def test_simple_game_act_rep_and_explicit_actions(oracle):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 64
    sim = BatchedSimulator("simple_game", {"array_size": 16}, num_envs=n)
    envs = [oracle.SimpleGame(16) for _ in range(n)]
    for g in envs:
        g.reset_game()
    rng = np.random.default_rng(0)
    for t in range(40):
        acts = rng.integers(0, 2, n).astype(np.int32)
        rep = int(rng.integers(1, 4))
        sim.step(torch.from_numpy(acts).cuda(), act_rep=rep)
        rew = sim.reward.cpu().numpy()
        codes = sim.game_over_codes.cpu().numpy()
        for e, g in enumerate(envs):
            r = np.float32(g.take_actions(int(acts[e]), rep))
            assert r == rew[e] and codes[e] == g.game_over()
    # out-of-range action: flagged, env untouched (the reference aborts: CHECK_LT)
    bad = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    before = sim.num_steps.clone()
    sim.step(bad)
    assert sim.check_errors() == n
    assert torch.equal(before, sim.num_steps)
    sim.close()


def test_simple_game_full_size_c2(oracle):
    """BASELINE config C2: 65 536 envs, array_size 64 -- whole-batch rollout vs the oracle batch driver."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps, A = 65536, 48, 64
    ref = oracle.sg_rollout(n, A, steps, policy_seed=77)
    sim = BatchedSimulator("simple_game", {"array_size": A}, num_envs=n, policy_seed=77)
    w = torch.arange(1, A + 1, dtype=torch.int64, device="cuda") * GOLD
    for t in range(steps):
        sim.reset_done()
        ck = (sim.obs.view(n, A).to(torch.int64) * w[None, :]).sum(1)
        assert np.array_equal(ck.cpu().numpy().view(np.uint64), ref.obs_ck[t]), t
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    sim.close()


def test_simple_game_autoreset_equals_step_then_reset(oracle):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 4096
    a = BatchedSimulator("simple_game", {"array_size": 16}, num_envs=n, policy_seed=9)
    b = BatchedSimulator("simple_game", {"array_size": 16}, num_envs=n, policy_seed=9)
    for t in range(120):
        a.step_autoreset()
        b.step()
        rb, cb = b.reward.clone(), b.game_over_codes.clone()
        b.reset_done()
        assert torch.equal(a.reward, rb) and torch.equal(a.game_over_codes, cb)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.num_steps, b.num_steps)
    a.close()
    b.close()


RACE_CASES = [
    dict(),
    dict(race_full_manouver=True),
    dict(difficulty="hard"),
    dict(track_type="circle"),
    dict(random=True),
    dict(track_type="circle", random=True, race_full_manouver=True),
    dict(context=3),
]


def _race_opts(case):
    o = {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}
    o.update(case)
    return o


def _race_oracle_cfg(oracle, case):
    return oracle.race_cfg(track_type=1 if case.get("track_type") == "circle" else 0,
                           race_full_manouver=int(case.get("race_full_manouver", False)),
                           random=int(case.get("random", False)),
                           difficulty_hard=0 if case.get("difficulty", "easy") == "easy" else 1,
                           context=case.get("context", 1))


@pytest.mark.parametrize("case", RACE_CASES)
def test_simple_race_rollout(oracle, case, trig):
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 2048, 150
    ctx = case.get("context", 1)
    ref = oracle.race_rollout(n, _race_oracle_cfg(oracle, case), seed=4242, steps=steps, policy_seed=31, env_gid0=100)
    sim = BatchedSimulator("simple_race", _race_opts(case), num_envs=n, seed=4242, policy_seed=31, env_gid0=100)
    nb = 16 * ctx
    w = torch.arange(1, nb + 1, dtype=torch.int64, device="cuda") * GOLD
    bad_r = bad_c = bad_o = 0
    for t in range(steps):
        sim.reset_done()
        ob = sim.obs.contiguous().view(torch.uint8).view(n, nb).to(torch.int64)
        ck = (ob * w[None, :]).sum(1).cpu().numpy().view(np.uint64)
        bad_o += int((ck != ref.obs_ck[t]).sum())
        sim.step()
        bad_r += int((sim.reward.cpu().numpy().view(np.uint32) != ref.rewards[t].view(np.uint32)).sum())
        bad_c += int((sim.game_over_codes.cpu().numpy() != ref.codes[t]).sum())
    print("simple_race", case, "mismatching env-steps: reward", bad_r, "code", bad_c, "obs", bad_o, "of", n * steps)
    assert bad_c == 0 and bad_r == 0 and bad_o == 0
    sim.close()


def test_simple_race_kat_survey_reward_unpinned_by_reference(oracle):
    """SURVEY.md 8(a) known answers (straight defaults, legal actions {4,7}) through the product."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("simple_race", _race_opts({}), num_envs=2)
    exp_r = [0.920154452, 0.969097912, 0.95105648, 1.0, 0.920154452, 0.71933651, 0.830473959, 0.629656076]
    got = []
    for a in [0, 1, 1, 0, 0, 0, 1, 0]:
        sim.step(torch.full((2,), a, dtype=torch.int32, device="cuda"))
        got.append(float(sim.reward[0]))
    assert np.array_equal(np.float32(got), np.float32(exp_r))
    sim.reset()
    total, k = np.float32(0), 0
    while int(sim.game_over_codes[1]) == 0:
        sim.step(torch.full((2,), k % 2, dtype=torch.int32, device="cuda"))
        total = np.float32(total + np.float32(float(sim.reward[1])))
        k += 1
    assert k == 65 and int(sim.game_over_codes[1]) == 2
    assert np.float32(total) == np.float32(29.732481)
    last = sim.env_obs(1).view(np.uint32)
    assert [hex(x) for x in last] == ["0x3f737871", "0xbe9e3778", "0xbf8287cd", "0x3eef7571"]
    sim.close()


@pytest.mark.parametrize("name,k", [("straight", 3), ("circle", 16)])
def test_simple_race_reference_frames_through_the_product(oracle, name, k):
    """The reference's own rendered SimpleRace frames (tests/test_oracle_race_doc_images.py: doc/simple_race_{1,2}.png carry the
    get_screen values of their state as text): the car is put one unit step behind a state that prints exactly the image's
    numbers, the HIP kernel drives it there (action 1 of the full manoeuvre set: forward, no turn), and its observation must
    equal the oracle's bit for bit AND print the image's five numbers."""
    torch = _torch()
    from test_oracle_race_doc_images import FRAMES, PI, printed_state, solve
    from xworld_amd.batched import BatchedSimulator
    hits, _ = solve(oracle, name)
    assert list(hits) == [k]
    pts = np.array(hits[k])
    x, y = pts.mean(axis=0)                                           # the middle of the region: away from its rounding edges
    ang = np.float32((PI / 2 + k * PI / 10) % (2 * PI))
    f = FRAMES[name]
    opts = dict(track_width=f["opts"]["track_width"], track_length=f["opts"].get("track_length", 100.0),
                track_radius=f["opts"].get("track_radius", 30.0), track_type="circle" if f["opts"]["track_type"] else "straight",
                race_full_manouver=True)
    sim = BatchedSimulator("simple_race", opts, num_envs=4)
    g = oracle.SimpleRace(race_full_manouver=1, **f["opts"])
    g.reset_game()
    x0, y0 = np.float32(x - np.cos(float(ang))), np.float32(y - np.sin(float(ang)))
    sim.race_set_car(2, float(x0), float(y0), float(ang))
    g.set_car(float(x0), float(y0), float(ang))
    acts = torch.full((4,), -1, dtype=torch.int32, device="cuda")
    acts[2] = 1
    sim.step(acts)
    r = np.float32(g.take_actions(1))
    obs = sim.env_obs(2).view(np.float32)
    assert np.array_equal(obs.view(np.uint32), g.state_screen().view(np.uint32)) and np.float32(float(sim.reward[2])) == r
    assert printed_state(obs) == f["printed"], (printed_state(obs), f["printed"])
    sim.close()


@pytest.mark.parametrize("name", ["circle", "straight"])
def test_simple_race_walk_into_the_reference_frame(oracle, name):
    """tests/test_oracle_race_doc_images.py WALKS: from reset, the default action set drives the HIP kernel into the state the
    reference's frame shows; its observation prints the image's numbers and every step equals the oracle's bit for bit."""
    torch = _torch()
    from test_oracle_race_doc_images import FRAMES, WALKS, printed_state
    from xworld_amd.batched import BatchedSimulator
    f = FRAMES[name]
    opts = dict(track_width=f["opts"]["track_width"], track_length=f["opts"].get("track_length", 100.0),
                track_radius=f["opts"].get("track_radius", 30.0), track_type="circle" if f["opts"]["track_type"] else "straight")
    sim = BatchedSimulator("simple_race", opts, num_envs=3)
    g = oracle.SimpleRace(**f["opts"])
    g.reset_game()
    for a in WALKS[name]:
        sim.step(torch.full((3,), a, dtype=torch.int32, device="cuda"))
        r = np.float32(g.take_actions(a))
        assert np.float32(float(sim.reward[1])) == r
        assert np.array_equal(sim.env_obs(1).view(np.uint32), g.state_screen().view(np.uint32))
    assert printed_state(sim.env_obs(1).view(np.float32)) == f["printed"]
    sim.close()


def test_simple_race_full_size_c3(oracle, trig):
    """BASELINE config C3: 65 536 envs, straight track, 2.6 M env-steps: every reward bit, code and observation, against
    the libm oracle (independent of the kernels' include/xwb_trig.h) and against the oracle on xwb_trig.h.  Mismatches are
    counted over the whole run and reported, not just the first one."""
    _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 65536, 40
    ref = oracle.race_rollout(n, _race_oracle_cfg(oracle, {}), seed=1, steps=steps, policy_seed=5)
    sim = BatchedSimulator("simple_race", _race_opts({}), num_envs=n, seed=1, policy_seed=5)
    bad_o = bad_r = bad_c = 0
    for t in range(steps):
        sim.reset_done()
        obs = sim.obs.cpu().numpy().reshape(n, 4).view(np.uint8)
        bad_o += int((oracle.obs_checksum_np(obs) != ref.obs_ck[t]).sum())       # the frame each policy step sees
        sim.step()
        bad_r += int((sim.reward.cpu().numpy().view(np.uint32) != ref.rewards[t].view(np.uint32)).sum())
        bad_c += int((sim.game_over_codes.cpu().numpy() != ref.codes[t]).sum())
    sim.close()
    assert (bad_o, bad_r, bad_c) == (0, 0, 0), "trig=%s: %d observations, %d rewards, %d codes of %d env-steps differ" % (
        trig, bad_o, bad_r, bad_c, n * steps)


def test_action_skip_leaves_env_untouched(oracle):
    """XWB_ACTION_SKIP (-1): a per-slot view steps one env; the others keep state, reward, code and screen."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 256
    sim = BatchedSimulator("simple_game", {"array_size": 16, "context": 2}, num_envs=n)
    g = oracle.SimpleGame(16, context=2)
    g.reset_game()
    before = sim.obs.clone()
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    for a in (1, 1, 0, 1):
        acts[7] = a
        sim.step(acts)
        r = np.float32(g.take_actions(a))
        assert np.float32(float(sim.reward[7])) == r and sim.check_errors() == 0
    obs = sim.obs.cpu().numpy().reshape(n, -1)
    assert np.array_equal(obs[7], g.state_screen())
    keep = torch.ones(n, dtype=torch.bool, device="cuda")
    keep[7] = False
    assert torch.equal(sim.obs[keep], before[keep]) and int(sim.num_steps[keep].sum()) == 0
    sim.close()


@pytest.mark.parametrize("game,opts", [("simple_game", {"array_size": 16, "context": 2}), ("simple_game", {"array_size": 64}),
                                        ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0, "random": True})])
def test_step_n_equals_n_autoreset_steps(oracle, game, opts):
    """xwb_step_n: n steps inside one launch leave exactly the state n separate step_autoreset calls leave."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 5000
    a = BatchedSimulator(game, opts, num_envs=n, policy_seed=21, seed=9)
    b = BatchedSimulator(game, opts, num_envs=n, policy_seed=21, seed=9)
    for k in (1, 7, 40):
        a.step_n(k)
        for _ in range(k):
            b.step_autoreset()
        assert torch.equal(a.reward, b.reward) and torch.equal(a.game_over_codes, b.game_over_codes)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.num_steps, b.num_steps) and torch.equal(a.actions, b.actions)
    a.step()                                                  # and the policy step counter moved with it
    b.step()
    assert torch.equal(a.actions, b.actions) and torch.equal(a.obs, b.obs)
    a.close()
    b.close()
