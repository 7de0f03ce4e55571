"""The one rendered frame the reference itself holds: doc/xworld2d.png (shown by games/xworld/README.md:2), committed as
tests/golden/xworld2d_doc.png.  It is a lossless capture of XMap::to_image(agent, false, visible_radius = 5) for an agent
heading up -- the output of the reference's real libjpeg + OpenCV 3.2 pipeline -- and pins, for the oracle:

  * the JPEG decode + 64-px cv::resize of the icon atlas (brick_1, robot_1, three goal icons), xitem.cpp:38-44;
  * XItem::get_item_image (cv::getRotationMatrix2D + cv::warpAffine, xitem.cpp:47-60) at 180 degrees (the agent) and at
    three non-trivial (yaw, scale, offset) poses (the goals; poses fitted by tests/golden/fit_doc_image.py);
  * XMap::image_masking: ROI and wall shadows (xmap.cpp:273-362), the black / white fill and the near-identity view
    rotation of XMap::to_image (xmap.cpp:125-206);
and, through oracle-vs-product parity on the GPU (tests/test_gpu_doc_image.py), the same for the HIP kernels.
Still NOT pinned by any reference-held vector: cv::resize(INTER_LINEAR) 64 -> 12 / 320 -> 80 and cvtColor(BGR2GRAY)
(XWorldSimulator::down_sample_image) -- restated from OpenCV 3.2's published fixed-point algorithm.  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R, CELL = 5, 64
# what the image shows, as (row, column) of the 5 x 5 view; the agent sits at (4, 2) and looks up
BRICKS = [(3, 0), (3, 1), (3, 3), (1, 4)]
BLACK = [(r, c) for r in range(3) for c in (0, 1, 3)] + [(0, 4)]
AGENT = (4, 2)


def doc_view():
    """[319, 320, 3] B,G,R: the view region of the PNG (origin x 0, y 3; its last pixel line is cut off)"""
    from PIL import Image
    a = np.array(Image.open(os.path.join(GOLD, "xworld2d_doc.png")).convert("RGB"))
    assert a.shape == (322, 322, 3)
    return np.ascontiguousarray(a[3:, 0:R * CELL, ::-1])


def doc_fit():
    with open(os.path.join(GOLD, "doc_image.json")) as f:
        return json.load(f)


def cell_of(view, r, c):
    return view[r * CELL:(r + 1) * CELL, c * CELL:(c + 1) * CELL]


def doc_world(oracle, dim, agent_xy, color=1):
    """The map read off the image inside a dim x dim world with the agent at agent_xy heading up; goals carry the fitted
    poses.  Returns (world, entity list)."""
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    path = {m["path"]: i for i, m in enumerate(pal.meta)}
    brick, robot = path["block/brick_1.jpg"], path["agent/robot_1.jpg"]
    ax, ay = agent_xy
    cell = lambda r, c: (ax + c - AGENT[1], ay + r - AGENT[0])          # noqa: E731  view (row, col) -> world (x, y)
    ents, poses = [], []
    for k, g in enumerate(doc_fit()["cells"]):
        x, y = cell(g["row"], g["col"])
        icon = path[g["icon_path"]]
        ents.append((0, x, y, icon, int(pal.name_arr[icon]), k))
        poses.append([g["yaw"], g["scale"], g["offset"]])
    for k, (r, c) in enumerate(BRICKS):
        x, y = cell(r, c)
        ents.append((1, x, y, brick, 0, 10 + k))
        poses.append([1.5707963, 1.0, 0.0])
    ents.append((2, ax, ay, robot, 0, 99))
    poses.append([-np.pi / 2, 1.0, 0.0])
    w = oracle.XWorld(pal, render=True, map_kind=0, max_dim=dim, dim=dim, num_goals=3, visible_radius=R, color=color,
                      tasks=["XWorld3DNavTarget"])
    w.stage_poses(poses)
    w.load_map_forced(ents, dim, [0] * 8)                 # task pick + the idle stage's draws (a goal to name)
    w.refresh_screen()
    return w, ents, poses


def test_plain_cells_bit_exact(oracle):
    """(a) brick, white, black and agent cells of the reference's frame equal the product's icon atlas bit for bit:
    brick upright, robot_1 turned by XItem::get_item_image's 180 degrees (yaw = -pi/2 -> 90 + 90)."""
    import ctypes as C
    view = doc_view()
    pal = oracle.Palette(oracle.NAV_SUBTREES)
    path = {m["path"]: i for i, m in enumerate(pal.meta)}
    brick = pal.icons64[path["block/brick_1.jpg"]]
    for r, c in BRICKS:
        assert np.array_equal(cell_of(view, r, c), brick), (r, c)
    for r, c in BLACK:
        assert (cell_of(view, r, c) == 0).all(), (r, c)
    goals = {(g["row"], g["col"]) for g in doc_fit()["cells"]}
    plain = set(BRICKS) | set(BLACK) | goals | {AGENT}
    for r in range(R):
        for c in range(R):
            if (r, c) not in plain:
                assert (cell_of(view, r, c) == 255).all(), (r, c)
    L = oracle.lib()
    M = (C.c_double * 6)()
    L.orc_cv_get_rotation_matrix_2d(32.0, 32.0, 90 - (-np.pi / 2) * 180 / np.pi, 1.0, M)
    robot = np.ascontiguousarray(pal.icons64[path["agent/robot_1.jpg"]])
    out = np.zeros_like(robot)
    white = np.array([255, 255, 255], np.uint8)
    L.orc_cv_warp_affine_8uc3(robot.ctypes.data_as(oracle.u8p), 64, 64, out.ctypes.data_as(oracle.u8p), 64, 64, M,
                              white.ctypes.data_as(oracle.u8p))
    got = cell_of(view, *AGENT)                                           # 63 lines: the capture lost the last one
    assert got.shape[0] == 63 and np.array_equal(got, out[:63])
    # the closed form of that warp: a half turn about (32, 32) moves pixel (x, y) to (64 - x, 64 - y)
    assert np.array_equal(out[1:, 1:], robot[::-1, ::-1][:-1, :-1]) and (out[0] == 255).all() and (out[:, 0] == 255).all()


@pytest.mark.parametrize("dim,agent", [(8, (3, 6)), (5, (2, 4))])
def test_image_masking_matches_the_frame(oracle, dim, agent):
    """(b) XMap::image_masking on the map read off the image: the ROI origin and exactly the image's ten black cells"""
    w, _, _ = doc_world(oracle, dim, agent)
    x, y, sh = w.agent_masking()
    assert (x, y) == (agent[0] + R - R // 2, agent[1] + R - R // 2 - R // 2)
    if dim == 8:
        assert (x, y) == (6, 7)
    exp = np.zeros((R, R), np.uint8)
    for r, c in BLACK:
        exp[r, c] = 1
    assert np.array_equal(sh, exp)


def test_goal_warps_reproduce_the_frame(oracle):
    """(c) cv::warpAffine at non-trivial poses: with the fitted (yaw, scale, offset) the oracle's XItem::get_item_image
    reproduces each goal cell of the reference's frame; the residual the fit reached is recorded in the fixture (0)."""
    view = doc_view()
    w, ents, poses = doc_world(oracle, 5, (2, 4))
    fit = doc_fit()["cells"]
    assert len(fit) == 3
    for k, g in enumerate(fit):
        got = w.entity_image(k).astype(np.int32)
        tgt = cell_of(view, g["row"], g["col"]).astype(np.int32)
        d = np.abs(got - tgt)
        assert int((d.max(2) > 0).sum()) <= g["differing_pixels"] and int(d.max()) <= g["max_abs_diff"], (g, d.sum())
        assert g["differing_pixels"] == 0 and g["max_abs_diff"] == 0          # what the committed fit reached
        assert 0.5 <= g["scale"] <= 1 and 0 <= g["offset"] <= 1 - g["scale"]  # xworld_env.py:217-223
        # the pose matters: the unposed icon is far from the cell
        raw = w.pal.icons64[ents[k][3]].astype(np.int32)
        assert np.abs(raw - tgt).mean() > 5


@pytest.mark.parametrize("dim,agent", [(5, (2, 4)), (8, (3, 6))])
def test_whole_view_equals_the_frame(oracle, dim, agent):
    """XMap::to_image end to end (canvas, item images, padding, shadows, crop, view rotation by 90 + yaw = 0 degrees):
    the oracle's 320 x 320 view equals the reference's frame on every one of its 319 x 320 pixels."""
    view = doc_view()
    w, _, _ = doc_world(oracle, dim, agent)
    got = w.agent_view()
    assert got.shape == (320, 320, 3)
    assert np.array_equal(got[:319], view)


def frame_from_doc(oracle, world_px, color=True):
    """What XWorldSimulator::get_screen makes of the reference's own view pixels: get_screen_rgb's resize to the world's
    pixel size, down_sample_image's resize to 80 x 80 (xworld_simulator.cpp:287-307,508-545), through the oracle's
    cv::resize restatement.  Neither resize reads the view's missing last line (checked by filling it two ways)."""
    L = oracle.lib()
    view = doc_view()
    outs = []
    for fill in (0, 255):
        full = np.full((320, 320, 3), fill, np.uint8)
        full[:319] = view
        a = full
        if world_px != 320:
            b = np.zeros((world_px, world_px, 3), np.uint8)
            L.orc_cv_resize_linear_8u(a.ctypes.data_as(oracle.u8p), 320, 320, 3, b.ctypes.data_as(oracle.u8p), world_px, world_px)
            a = b
        o = np.zeros((80, 80, 3), np.uint8)
        L.orc_cv_resize_linear_8u(a.ctypes.data_as(oracle.u8p), world_px, world_px, 3, o.ctypes.data_as(oracle.u8p), 80, 80)
        outs.append(o)
    assert np.array_equal(outs[0], outs[1])
    o = outs[0]
    if not color:
        g = np.zeros((80, 80), np.uint8)
        L.orc_cv_bgr2gray_8u(o.ctypes.data_as(oracle.u8p), 80 * 80, g.ctypes.data_as(oracle.u8p))
        return g[None]
    return np.ascontiguousarray(o.transpose(2, 0, 1))                      # planar B,G,R


@pytest.mark.parametrize("dim,agent,color", [(5, (2, 4), 1), (8, (3, 6), 1), (5, (2, 4), 0)])
def test_screen_equals_downsampled_frame(oracle, dim, agent, color):
    """(d) the oracle's get_screen of that map == the reference's frame pushed through the two resizes"""
    w, _, _ = doc_world(oracle, dim, agent, color=color)
    assert w.dims == (80, 80, 3 if color else 1)
    assert np.array_equal(w.screen(), frame_from_doc(oracle, dim * 64, bool(color)))
