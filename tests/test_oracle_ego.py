"""Egocentric mode (FLAGS_visible_radius > 0) in the oracle: six first-person actions, yaw, the teacher's reach test
along the agent's heading -- against the reference's own Python tasks (tests/golden/tasks_ego.json) -- and
XMap::image_masking / the OpenCV warp restatements against hand-derived answers.  CPU only."""
import json
import os

import numpy as np
import pytest

from test_oracle_tasks import EVENTS, GOLD, KINDS, STAGES


def ego_runs(kind):
    with open(os.path.join(GOLD, "tasks_ego.json")) as f:
        return json.load(f)[kind]


@pytest.mark.parametrize("kind", KINDS)
def test_ego_tasks_match_reference(oracle, kind):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    seen = set()
    for run in ego_runs(kind):
        d = run["max_dim"]
        w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=d, dim=run["dim"], num_goals=4, tasks=[kind], visible_radius=3)
        assert w.num_actions() == 6
        w.stage_poses(run["poses"])
        w.load_map_ex([tuple(e) for e in run["entities_before"]], run["dim"], [0] + run["decisions"])
        g = w.grid()
        exp = np.zeros_like(g)
        for t, x, y, icon, name, serial in run["entities_after"]:
            exp[y, x] = icon + 1
        assert np.array_equal(g, exp), run["py_seed"]
        agent = [e for e in run["entities_after"] if e[0] == 2][0]
        assert w.agent_xy() == (agent[1], agent[2])
        if kind != "XWorld3DNavTargetDirection":              # Direction's target set depends on the current yaw
            tc = np.zeros_like(g, dtype=np.uint8)
            for x, y in run["target_cells"]:
                tc[y, x] = 1
            assert np.array_equal(w.target_cells(), tc), run["py_seed"]
        for t, (a, reward, event, stage, ax, ay, success, yaw) in enumerate(run["trace"]):
            r = np.float32(w.take_actions(a))
            assert r == np.float32(reward), (run["py_seed"], t, r, reward)
            assert w.event() == EVENTS[event] and w.stage() == STAGES[stage], (run["py_seed"], t)
            assert w.agent_xy() == (ax, ay) and w.last_action_success() == success, (run["py_seed"], t)
            assert w.agent_yaw() == yaw, (run["py_seed"], t)
            seen.add(event)
    assert {"correct_goal", "wrong_goal"} <= seen


def _world(oracle, blocks, agent, yaw, dim=7, r=3):
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    brick = [i for i, m in enumerate(pal.meta) if m["type"] == "block"][0]
    robot = [i for i, m in enumerate(pal.meta) if m["type"] == "agent"][0]
    ents = [(1, x, y, brick, 0, k) for k, (x, y) in enumerate(blocks)] + [(2, agent[0], agent[1], robot, 0, 99)]
    w = oracle.XWorld(pal, render=False, map_kind=0, max_dim=dim, dim=dim, visible_radius=r, tasks=["XWorld3DNavTarget"])
    w.stage_poses([[1.5707963, 1, 0]] * len(blocks) + [[yaw, 1, 0]])
    w.load_map_ex(ents, dim, [0])
    return w


def test_image_masking_roi_and_shadows(oracle):
    """xmap.cpp:273-362.  The ROI is the r x r block of cells in front of the agent (agent in the middle of the near
    edge), in padded coordinates; a wall block shadows the cells behind it along its scan line, and a block right
    beside the agent blocks the scan lines that start further out."""
    # 4x4 map, agent (1,1), yaw 0 (facing right), r = 3 -> ROI origin (4, 3)
    w = _world(oracle, [], (1, 1), 0.0, dim=4)
    x, y, sh = w.agent_masking()
    assert (x, y) == (4, 3) and not sh.any()
    # facing down from (3,1): ROI cells x 2..4, y 1..3 -> origin (2+3, 1+3); a block two cells ahead shadows nothing in
    # its own cell but the cell behind it
    w = _world(oracle, [(3, 2)], (3, 1), np.pi / 2)
    x, y, sh = w.agent_masking()
    assert (x, y) == (5, 4)
    assert sh.tolist() == [[0, 0, 0], [0, 0, 0], [0, 1, 0]]
    # facing up: the scan runs from the agent's row upwards
    w = _world(oracle, [(3, 4)], (3, 5), -np.pi / 2)
    x, y, sh = w.agent_masking()
    assert (x, y) == (5, 6) and sh.tolist() == [[0, 1, 0], [0, 0, 0], [0, 0, 0]]
    # facing left with a block directly at the agent's right-hand side (the cell above it on the map is its left):
    # rays that start beyond that block are blocked for the whole line, and the block's own line is dark behind it
    w = _world(oracle, [(4, 2)], (4, 3), np.pi, r=5)
    x, y, sh = w.agent_masking()
    assert (x, y) == (4 + 5 - 2 - 2, 3 + 5 - 2)
    assert sh.tolist() == [[1, 1, 1, 1, 1], [1, 1, 1, 1, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]


def test_warp_affine_quarter_turns_and_identity(oracle):
    """cv::warpAffine restatement: the near-identity warp every default-pose item goes through is a pure copy; quarter
    turns about (w/2, h/2) are exact permutations shifted by one pixel, the vacated row / column takes the border."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    M = (C.c_double * 6)()
    border = np.array([255, 255, 255], np.uint8)

    def warp(angle, scale=1.0):
        L.orc_cv_get_rotation_matrix_2d(32.0, 32.0, angle, scale, M)
        dst = np.zeros_like(src)
        L.orc_cv_warp_affine_8uc3(src.ctypes.data_as(oracle.u8p), 64, 64, dst.ctypes.data_as(oracle.u8p), 64, 64, M,
                                  border.ctypes.data_as(oracle.u8p))
        return dst
    assert np.array_equal(warp(90 - 1.5707963 * 180 / np.pi), src)               # default yaw: 1.5e-6 degrees
    r180 = warp(180.0)
    assert np.array_equal(r180[1:, 1:], src[::-1, ::-1][:-1, :-1]) and (r180[0] == 255).all() and (r180[:, 0] == 255).all()
    r90 = warp(90.0)                                                              # counter-clockwise on screen
    exp = np.rot90(src, 1)
    assert np.array_equal(r90[1:, :], exp[:-1, :]) and (r90[0] == 255).all()
    half = warp(0.0, 0.5)                                                         # scale 0.5 about the centre
    assert (half[:15] == 255).all() and (half[:, :15] == 255).all() and (half[49:] == 255).all()
    assert np.array_equal(half[16:48:1, 16:48:1][::1, ::1], src[0:64:2, 0:64:2])
