"""GPU parity for the five tasks of the XWorld3DNav group (confs/navigation2d.json): the reset kernel's idle
stages (task sampling, goal / agent rearrangement, target set) and the step kernel's reward rule, through the
C ABI, against the CPU oracle and against traces of the reference's own Python tasks (tests/golden/tasks.json).
"""
import json
import os

import numpy as np
import pytest

from test_gpu_xworld import MAPS, _torch
from test_oracle_tasks import EVENTS, GOLD, KINDS, STAGES, runs_of

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")


def _make(oracle, key, n, tasks, seed=0xC0FFEE, policy_seed=0x5EED, gid0=0, **opts):
    from xworld_amd.batched import BatchedSimulator
    conf, popts, ocfg = MAPS[key]
    o = {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "tasks": list(tasks)}
    o.update(popts)
    o.update(opts)
    sim = BatchedSimulator("xworld", o, num_envs=n, seed=seed, policy_seed=policy_seed, env_gid0=gid0)
    pal = oracle.Palette(oracle.NAV_SUBTREES if ocfg["map_kind"] == 0 else oracle.WALLS_SUBTREES)
    cfg = dict(ocfg)
    cfg.update(seed=seed, tasks=list(tasks), max_steps=int(opts.get("max_steps", 0)))
    return sim, pal, cfg


CASES = [(k, KINDS) for k in MAPS] + [("nav8", [t]) for t in KINDS] + [("walls7", [KINDS[1], KINDS[3]]),
                                                                         ("nav11", [KINDS[2]])]


@pytest.mark.parametrize("key,tasks", CASES, ids=lambda v: v if isinstance(v, str) else "+".join(t[11:] for t in v))
def test_reset_idle_stage(oracle, key, tasks):
    """xwb-taskgen-v1: task drawn, map after the rearrangement, agent cell, target set, middle cell."""
    _torch()
    n = 512
    sim, pal, cfg = _make(oracle, key, n, tasks, seed=1234, gid0=77)
    ow = oracle.XWorld(pal, render=False, **cfg)
    md = cfg["max_dim"]
    kinds = np.zeros(5, int)
    with_target = 0
    for episode in range(2):
        if episode:
            sim.reset()
        for e in range(n):
            ow.reset_game(77 + e, episode)
            st = sim.env_state(e)
            assert st.xw_task == ow.task_kind(), (episode, e)
            raw = sim.env_grid(e, raw=True)
            assert np.array_equal((raw & 0x7fff).astype(np.int32), ow.grid()), (episode, e, st.xw_task)
            assert (st.xw_agent_x, st.xw_agent_y) == ow.agent_xy(), (episode, e, st.xw_task)
            assert np.array_equal((raw >> 15).astype(np.uint8), ow.target_cells()), (episode, e, st.xw_task)
            assert st.xw_target_name == ow.target_name() and st.xw_stage == ow.stage() == 1
            if st.xw_task == 2 and st.xw_target >= 0:
                assert (st.xw_target % md, st.xw_target // md) == tuple(ow.between_cell())
            kinds[st.xw_task] += 1
            with_target += int((raw >> 15).any()) or int(st.xw_task == 2 and st.xw_target >= 0)
    assert all(kinds[oracle.TASK_ID[t]] > 0 for t in tasks) and kinds.sum() == 2 * n
    # Direction: the target set is empty unless the direction word drawn at idle time also holds along the agent's
    # fixed yaw (reference behaviour of the full-observation game, pinned by tests/golden/tasks.json)
    assert with_target > (0.25 if KINDS[3] in tasks else 0.8) * 2 * n, with_target
    sim.close()


@pytest.mark.parametrize("key,tasks", [("nav8", KINDS), ("nav7", KINDS), ("walls7", KINDS), ("nav8_dim5", KINDS),
                                       ("nav8", [KINDS[2]]), ("nav8", [KINDS[3]])],
                         ids=lambda v: v if isinstance(v, str) else str(len(v)))
def test_rollout_all_tasks(oracle, key, tasks):
    """Random-policy rollouts with resets over the whole group: reward bits and codes of every env-step."""
    _torch()
    n, steps = 1536, 240
    sim, pal, cfg = _make(oracle, key, n, tasks, seed=21, policy_seed=5, gid0=9)
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=5, env_gid0=9)
    resets = 0
    for t in range(steps):
        sim.reset_done()
        resets += sim.done_count()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    assert resets == ref.stats.resets and resets > 0
    sim.close()


def test_rollout_c4_all_tasks_screens(oracle):
    """BASELINE C4 shape (7x7 colour) with the five tasks: obs checksums, rewards and codes of every env-step."""
    torch = _torch()
    n, steps = 1024, 60                # (the oracle renders every frame on one host core: the test's whole cost)
    sim, pal, cfg = _make(oracle, "nav7", n, KINDS, seed=3, policy_seed=8, color=True)
    cfg["color"] = 1
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=8, render=True)
    nb = sim.obs_bytes_per_env
    w = torch.arange(1, nb + 1, dtype=torch.int64, device="cuda") * -7046029254386353131
    for t in range(steps):
        sim.reset_done()
        ck = (sim.obs.reshape(n, nb).to(torch.int64) * w[None, :]).sum(1)
        assert np.array_equal(ck.cpu().numpy().view(np.uint64), ref.obs_ck[t]), t
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    sim.close()


@pytest.mark.parametrize("kind", KINDS)
def test_reference_task_traces_through_product(oracle, kind):
    """The reference's Python tasks' own episodes (tests/golden/tasks.json): the map after the idle stage and the
    target set are loaded into the product; its step kernel must return the reference's rewards / events."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    runs = runs_of(kind)
    by_dim = {}
    for r in runs:
        by_dim.setdefault((r["max_dim"], r["dim"]), []).append(r)
    seen = set()
    for (md, dim), rs in by_dim.items():
        n = len(rs)
        sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "nav_target.json"), "max_dim": md, "dim": dim,
                                          "task_mode": "lang_acquisition", "tasks": [kind]}, num_envs=n)
        for e, run in enumerate(rs):
            g = np.zeros((md, md), np.uint16)
            agent = None
            for t, x, y, icon, name, serial in run["entities_after"]:
                g[y, x] = icon + 1
                if t == 2:
                    agent = (x, y)
            for x, y in run["target_cells"]:
                g[y, x] |= 0x8000
            target = -1
            if kind == "XWorld3DNavTargetBetween":
                target = run["between"][1] * md + run["between"][0]
            sim.load_map(e, g, agent[0], agent[1], dim=dim, task=kind, target=target)
        T = max(len(r["trace"]) for r in rs)
        for t in range(T):
            acts = np.full(n, -1, np.int32)
            for e, run in enumerate(rs):
                if t < len(run["trace"]):
                    acts[e] = run["trace"][t][0]
            sim.step(torch.from_numpy(acts).cuda())
            rew = sim.reward.cpu().numpy()
            for e, run in enumerate(rs):
                if t >= len(run["trace"]):
                    continue
                a, reward, event, stage, ax, ay, success = run["trace"][t]
                st = sim.env_state(e)
                assert rew[e] == np.float32(reward), (run["py_seed"], t)
                assert st.xw_event == EVENTS[event] and st.xw_stage == STAGES[stage], (run["py_seed"], t)
                assert (st.xw_agent_x, st.xw_agent_y) == (ax, ay) and st.last_action_success == success
                seen.add(event)
        sim.close()
    assert {"correct_goal", "wrong_goal"} <= seen


def test_conf_navigation2d_lists_five_tasks():
    _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "navigation2d.json")}, num_envs=4096, seed=2)
    assert sim.tasks == [0, 1, 2, 3, 4]
    kinds = np.bincount([sim.env_state(e).xw_task for e in range(0, 4096, 4)], minlength=5)
    assert kinds.min() > 140 and kinds.max() < 270, kinds
    sim.close()


# ------------------------------------------------------------------ rule D14b: the 2-D-native group of confs/walls.json
from test_oracle_tasks2d import KINDS2D, STAGES2D, load2d   # noqa: E402


@pytest.mark.parametrize("key,mode,tasks", [("nav8", "lang_acquisition", KINDS2D), ("nav8", "one_channel", KINDS2D),
                                            ("walls7", "one_channel", KINDS2D), ("nav11", "one_channel", KINDS2D),
                                            ("nav8_dim5", "one_channel", KINDS2D),
                                            ("walls7", "one_channel", [KINDS2D[2]]), ("nav7", "lang_acquisition", [KINDS2D[0]])],
                         ids=lambda v: v if isinstance(v, str) else str(len(v)))
def test_rule_d14b_reset_and_rollout(oracle, key, mode, tasks):
    """Task resampling whenever the group is idle (also at step time, stream 2), candidate tables, -0.1 / -0.3 rewards,
    the one_channel time-up back to idle; episodes end by max_steps only."""
    _torch()
    n, steps = 1024, 150
    sim, pal, cfg = _make(oracle, key, n, tasks, seed=31, policy_seed=6, gid0=3, task_mode=mode, max_steps=41)
    cfg["task_mode"] = 0 if mode == "lang_acquisition" else 1
    md = cfg["max_dim"]
    ow = oracle.XWorld(pal, render=False, **cfg)
    kinds = np.zeros(9, int)
    for e in range(0, n, 3):
        ow.reset_game(3 + e, 0)
        st = sim.env_state(e)
        assert st.xw_task == ow.task_kind() and st.xw_stage == ow.stage(), e
        tx, ty = ow.target2d()
        assert st.xw_target == (ty * md + tx if tx >= 0 else -1), e
        assert not (sim.env_grid(e, raw=True) >> 15).any()
        kinds[st.xw_task] += 1
    assert all(kinds[oracle.TASK_ID[t]] > 0 for t in tasks)
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=6, env_gid0=3)
    resets = 0
    seen = set()
    for t in range(steps):
        sim.reset_done()
        resets += sim.done_count()
        sim.step()
        r = sim.reward.cpu().numpy()
        assert np.array_equal(r.view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
        seen.update(np.unique(r).tolist())
    assert resets == ref.stats.resets == n * (steps // 41)
    assert seen - {0.0} == {float(np.float32(-0.1)), float(np.float32(-0.1 + -0.2))}, seen
    assert (0.0 in seen) == (mode == "one_channel" or len(tasks) > 1)
    sim.close()


@pytest.mark.parametrize("key", [k for k in sorted(load2d())
                                 if k.startswith("lang_acquisition") or k.endswith(("Near", "Between"))])
def test_reference_2d_task_traces_through_product(oracle, key):
    """tests/golden/tasks2d.json (the reference's Python tasks) replayed through the product.  The one_channel
    Target / ColorTarget runs re-pick a target after each time-up with decisions only the oracle can be forced
    through; they are covered by the oracle (test_oracle_tasks2d.py) + the oracle-vs-product rollouts above."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    mode, mapk, name = key.split("/")
    runs = load2d()[key]
    md, dim = runs[0]["max_dim"], runs[0]["dim"]
    conf = os.path.join(CONF, "nav_target.json" if mapk == "nav" else "walls_target.json")
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "max_dim": md, "dim": dim, "task_mode": mode,
                                      "tasks": [name]}, num_envs=len(runs))
    for e, run in enumerate(runs):
        g = np.zeros((md, md), np.uint16)
        agent = None
        for t, x, y, icon, nm, serial in run["entities"]:
            g[y, x] = icon + 1
            if t == 2:
                agent = (x, y)
        tx, ty = run["reset_teach"][5:7]
        sim.load_map(e, g, agent[0], agent[1], dim=dim, task=name, target=ty * md + tx if tx >= 0 else -1)
        assert sim.env_state(e).xw_stage == STAGES2D[run["reset_teach"][4]]
    for t in range(len(runs[0]["trace"])):
        acts = np.array([run["trace"][t][0] for run in runs], np.int32)
        sim.step(torch.from_numpy(acts).cuda())
        rew = sim.reward.cpu().numpy()
        for e, run in enumerate(runs):
            a, ax, ay, success, was_idle, decs, reward, event, stage, tx, ty = run["trace"][t]
            st = sim.env_state(e)
            assert rew[e] == np.float32(reward), (run["py_seed"], t)
            assert st.xw_stage == STAGES2D[stage] and st.xw_event == 0 and st.game_over == 0
            assert (st.xw_agent_x, st.xw_agent_y) == (ax, ay) and st.last_action_success == success
            assert st.xw_target == (ty * md + tx if tx >= 0 else -1)
    sim.close()


def test_conf_walls_json_navigation_group(tmp_path):
    _torch()
    import json
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "walls.json")}, num_envs=64, seed=2)
    assert sim.tasks == [5, 6, 7, 8]
    sim.close()
    # a conf that also lists a group this build does not run (the reference's walls.json lists the language group): the
    # built group runs, the other is skipped with a warning (tests/test_gpu_groups.py loads the reference's own file)
    with open(os.path.join(CONF, "walls.json")) as f:
        conf = json.load(f)
    conf["task_groups"]["XWorldRec"] = {"schedule": "weighted", "weight": 1, "tasks": {"XWorldRecColorToObject": 1}}
    two = tmp_path / "two_groups.json"
    two.write_text(json.dumps(conf))
    with pytest.warns(UserWarning, match="XWorldRec"):
        sim = BatchedSimulator("xworld", {"xwd_conf_path": str(two)}, num_envs=4)
    assert sim.tasks == [5, 6, 7, 8] and sim.cfg.n_tasks2 == 0
    sim.close()
    sim = BatchedSimulator("xworld", {"xwd_conf_path": str(two), "task_group": "XWorldNav"}, num_envs=4)
    assert sim.tasks == [5, 6, 7, 8]
    sim.close()
    with pytest.raises(RuntimeError):
        BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "nav_target.json"),
                                    "tasks": ["XWorldNavTarget", "XWorld3DNavTarget"]}, num_envs=4)



# ------------------------------------------------------------------------------------------ the teacher's sentences
def _sent(sim, e):
    """The sentence of env e from the Python layer (language.py) -- and the same one built inside libxwb.so (xwb_sentence)."""
    s = sim.sentence(e)
    assert sim.sentence_c(e) == s, (e, s, sim.sentence_c(e))
    return s


def test_sentences_follow_the_episode(oracle):
    """The names the idle stage binds (vs the oracle), one instruction per episode that contains them, the closing
    message on the step that ends the episode, silence afterwards, a new instruction after the reset."""
    torch = _torch()
    from xworld_amd import language
    n = 384
    sim, pal, cfg = _make(oracle, "nav8", n, KINDS, seed=77, policy_seed=1, gid0=500)
    names = sim.palette.names["goal"]
    assert names == pal.names["goal"]
    ow = oracle.XWorld(pal, render=False, **cfg)
    first = {}
    starts = set()
    for e in range(n):
        ow.reset_game(500 + e, 0)
        st = sim.env_state(e)
        a, b = ow.sentence_names()
        assert (st.xw_sentence_names & 0xffff, st.xw_sentence_names >> 16) == (a & 0xffff, b & 0xffff), (e, st.xw_task)
        s = _sent(sim, e)
        first[e] = s
        if a >= 0:
            assert names[a] in s.split() and (b < 0 or names[b] in s.split()), (e, s)
            starts.add(s.split()[0])
        else:
            assert s == ""
    assert len(starts) >= 6 and len(set(first.values())) > n // 2       # the grammar's variety shows
    closing = {1: "Well done !", 2: "Wrong !", 3: "Time up ."}
    seen = set()
    for t in range(120):
        sim.step()
        for e in range(0, n, 7):
            st = sim.env_state(e)
            s = _sent(sim, e)
            if st.xw_event:
                assert s == closing[st.xw_event]
                seen.add(st.xw_event)
            elif st.xw_stage == 2:
                assert s == ""
            else:
                assert s == first[e]                              # the instruction is repeated every step
        sim.reset_done()
        for e in range(0, n, 7):
            st = sim.env_state(e)
            if st.num_steps == 0 and st.episode > 0:
                first[e] = _sent(sim, e)
    assert {1, 2} <= seen
    # the py_simulator surface shows it under "sentence"
    from xworld_amd.py_simulator import Simulator
    g = Simulator.create("xworld", {"xwd_conf_path": os.path.join(CONF, "navigation2d.json"), "task_mode": "lang_acquisition"})
    g.reset_game()
    assert g.get_state()["sentence"] == g.batch.sentence(0) == g.batch.sentence_c(0) != ""
    # ... and so does the state packet the C / C++ / TCP callers get (reference wire layout), "-" once the teacher is silent
    L = oracle.lib()
    for e in (0, 7, 14):
        raw = sim.state_packet(env=e, reward=0.5)
        buf = np.frombuffer(raw, np.uint8).copy()
        out = (oracle.PacketField * 4)()
        assert L.orc_packet_decode(oracle.ptr(buf, oracle.u8p), len(raw), out, 4) == 3
        keys = {out[i].key: out[i] for i in range(3)}
        want = sim.sentence(e) or "-"
        assert want.encode() in raw and set(keys) == {b"reward", b"screen", b"sentence"}, (e, want)
    sim.close()


def test_sentences_of_the_2d_native_group(oracle):
    """XWorldNavTarget / ColorTarget speak on the teach() call that picks the target only; the sentence names the goal
    (and its colour) at the target cell."""
    _torch()
    n = 256
    sim, pal, cfg = _make(oracle, "nav8", n, [KINDS2D[0], KINDS2D[2]], seed=5, task_mode="lang_acquisition", max_steps=30)
    md = cfg["max_dim"]
    spoke = 0
    for e in range(n):
        st = sim.env_state(e)
        s = _sent(sim, e)
        if st.xw_stage == 1:
            icon = int(sim.env_grid(e)[st.xw_target // md, st.xw_target % md]) - 1
            m = sim.palette.meta[icon]
            assert m["name"] in s.split(), (e, s)
            if st.xw_task == 7:
                assert m["color"] in s.split() and m["color"] != "na"
            spoke += 1
        else:
            assert s == ""
    assert spoke > n // 2
    sim.step()
    assert all(_sent(sim, e) == "" for e in range(0, n, 5))      # the navigation stage says nothing
    sim.close()
    # one_channel: the task runs out of time after h * w / 2 = 32 steps and says so on that step (xworld_task.py:205-211)
    sim, pal, cfg = _make(oracle, "nav8", 64, [KINDS2D[0]], seed=5, task_mode="one_channel")
    busy = [e for e in range(64) if sim.env_state(e).xw_stage == 1]
    assert len(busy) > 32
    for t in range(32):
        assert all(_sent(sim, e) == "" for e in busy[:4]) or t == 0
        sim.step()
    for e in busy:
        st = sim.env_state(e)
        assert st.xw_stage == 0 and st.xw_steps_in_task == 0 and _sent(sim, e) == "Time up ."
    sim.step()                                                      # the next teach() call picks a new target and speaks
    assert all(_sent(sim, e) not in ("", "Time up .") for e in busy)
    sim.close()


def test_weighted_schedule(oracle):
    """Schedule "weighted" (teaching_task.cpp:204-213, util::simple_importance_sampling): tasks are drawn by the conf's
    per-task weights -- same draws as the oracle, frequencies as the weights say, rollout parity through resets."""
    _torch()
    n = 4096
    weights = [4.0, 1.0, 0.5, 0.5, 2.0]
    sim, pal, cfg = _make(oracle, "nav8", n, KINDS, seed=77, policy_seed=3, gid0=5, task_weights=weights)
    cfg["task_weights"] = weights
    ow = oracle.XWorld(pal, render=False, **cfg)
    counts = np.zeros(5)
    for e in range(n):
        ow.reset_game(5 + e, 0)
        st = sim.env_state(e)
        assert st.xw_task == ow.task_kind(), e
        counts[st.xw_task] += 1
    assert np.abs(counts / n - np.array(weights) / sum(weights)).max() < 0.03, counts
    steps = 200
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=3, env_gid0=5)
    for t in range(steps):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    sim.close()
    # the 2-D-native group draws a task at step time too
    w2 = [1.0, 3.0, 1.0, 2.0]
    kinds2 = ["XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"]
    sim, pal, cfg = _make(oracle, "walls7", 2048, kinds2, seed=8, policy_seed=1, task_weights=w2, task_mode="one_channel", max_steps=60)
    cfg.update(task_weights=w2, task_mode=1, max_steps=60)
    ref = oracle.xw_rollout(2048, oracle.xw_cfg(**cfg), pal, 150, policy_seed=1)
    for t in range(150):
        sim.reset_done()
        sim.step()
        assert np.array_equal(sim.reward.cpu().numpy().view(np.uint32), ref.rewards[t].view(np.uint32)), t
        assert np.array_equal(sim.game_over_codes.cpu().numpy(), ref.codes[t]), t
    sim.close()


def test_teacher_performance_counters(oracle):
    """Teacher::report_task_performance (teacher.cpp:175-200): successes / failures / success_steps per task class, counted on
    the device over a 10 000-step soak with resets, against the oracle's tallies of the same rollout; the report text; the
    counters travel with a checkpoint."""
    _torch()
    from xworld_amd.batched import BatchedSimulator
    n, steps = 1024, 10000
    conf = os.path.join(CONF, "navigation2d.json")
    sim = BatchedSimulator("xworld", {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16},
                           num_envs=n, seed=77, policy_seed=12)
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    cfg = dict(map_kind=0, max_dim=7, dim=7, num_goals=4, num_blocks=16, seed=77, tasks=[0, 1, 2, 3, 4])
    ref = oracle.xw_rollout(n, oracle.xw_cfg(**cfg), pal, steps, policy_seed=12)
    blob = None
    for t in range(steps):
        sim.reset_done()
        sim.step()
        if t == 4000:
            blob, at4000 = sim.save_state(), sim.task_performance()
    # the example loop resets at the top of an iteration: the games that ended on the last step are still to be reset
    got, resets = sim.task_performance()
    want = {}
    from xworld_amd.lib import TASK_CLASSES
    for k in range(9):
        row = tuple(int(x) for x in ref.stats.task_perf[k])
        if row[0] + row[1]:
            want[TASK_CLASSES[k]] = row
    assert got == want, (got, want)
    assert len(got) == 5 and all(v[0] > 100 and v[1] > 100 and v[2] > v[0] for v in got.values())
    assert resets == n + ref.stats.resets                       # the first reset of every env + the example loop's
    text = sim.task_performance_report()
    lines = text.splitlines()
    assert lines[0] == "=== XWorld3DNavTarget ===" and len(lines) == 10
    s, f, ss, _ = got["XWorld3DNavTarget"]
    assert lines[1] == "=== %d(S)/%d(F) -> %g@%g" % (s, f, s / (s + f), ss / s)
    sim.load_state(blob)
    assert sim.task_performance() == at4000
    sim.close()
    # one_channel, the 2-D-native group: failures are time-ups, no success_steps
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "walls.json"), "task_mode": "one_channel", "max_steps": 200},
                           num_envs=256, seed=3)
    for t in range(400):
        sim.reset_done()
        sim.step()
    got, _ = sim.task_performance()
    assert set(got) <= {"XWorldNavTarget", "XWorldNavColorTarget"} and got
    assert all(v[0] == 0 and v[1] == v[3] > 0 and v[2] == 0 for v in got.values()), got
    sim.close()
