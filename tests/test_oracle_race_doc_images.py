"""SimpleRace against the two rendered frames the reference itself holds: doc/simple_race_1.png (circular track) and
doc/simple_race_2.png (straight track), shown by games/simple_race/README.md:2 and committed as
tests/golden/simple_race_doc_{circle,straight}.png.  They are RaceEngine::draw() canvases (simple_race_simulator.cpp:343-383,
480 x 720, scaled to 300 x 450 for the README): the track and the car drawn at their window coordinates, and -- as text --
the four numbers RaceEngine::get_screen (:412-430) returned for that state, printed with "%.2f": orientation in degrees
(+-acos(cos_theta) / PI * 180, PI = 3.1415926), cos_theta, sin_theta, "dist to middle of lane" (horizontal displacement),
"dist to finish line" (vertical displacement).  No reference test covers SimpleRace; these two frames are the only
reference-held known answers for it.  What they pin: the track geometry (StraightTrack / CircleTrack constructors, start /
end lines, WINDOW 480 x 720), get_tangent_vec, the cos / sin / sign convention of the orientation, both displacement
normalisations.  What they cannot pin: the dynamics and the rewards (one frame each).

The test: read the track geometry and the car off the image (blue disc -> window position to about half a pixel; the car's
heading is one of the twenty multiples of PI / 10 from PI / 2 the engine can reach), then require that exactly one heading
exists, and a position within 1.25 window pixels of the measured one, for which the oracle's get_screen prints EXACTLY the
five numbers of the image.  CPU only; tests/test_gpu_simple.py drives the same states through the HIP kernel."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PI = 3.1415926                                                    # simple_race_simulator.h:39
SCALE = 480 / 300                                                 # README image pixels -> window pixels

# what the images say (transcribed from the text RaceEngine::draw put on the canvas)
FRAMES = {
    "straight": dict(file="simple_race_doc_straight.png", printed=("-54.00", "0.59", "-0.81", "-0.44", "-0.70"),
                     opts=dict(track_type=0, track_length=400.0, track_width=80.0)),
    "circle": dict(file="simple_race_doc_circle.png", printed=("76.75", "0.23", "0.97", "0.14", "0.00"),
                   opts=dict(track_type=1, track_radius=80.0, track_width=40.0)),
}


def _img(name):
    from PIL import Image
    a = np.array(Image.open(os.path.join(GOLD, FRAMES[name]["file"])).convert("RGB")).astype(int)
    assert a.shape == (450, 300, 3)
    return a


def to_window(i_lo, i_hi):
    """centre, in window pixel coordinates, of the image pixels i_lo .. i_hi (inclusive): image pixel i covers window
    [1.6 i, 1.6 i + 1.6) and window pixel j is the point j + 0.5 of that axis"""
    return (i_lo + i_hi + 1) / 2 * SCALE - 0.5


def measured_car(name):
    a = _img(name)
    blue = (a[..., 2] > 150) & (a[..., 0] < 100) & (a[..., 1] < 100)       # CircleCar::draw: Scalar(255, 0, 0) in B,G,R
    ys, xs = np.nonzero(blue)
    assert 15 < len(xs) < 60
    return to_window(xs.min(), xs.max()), to_window(ys.min(), ys.max())


def printed_state(s):
    """RaceEngine::draw's sprintf of a get_screen result"""
    theta = np.arccos(np.float32(s[0])) if s[1] >= 0 else -np.arccos(np.float32(s[0]))
    return ("%.2f" % (float(theta) / PI * 180), "%.2f" % s[0], "%.2f" % s[1], "%.2f" % s[2], "%.2f" % s[3])


def solve(oracle, name, step=0.02, reach=1.25):
    """every (heading index k, x, y) on a grid around the measured car whose printed state equals the image's"""
    f = FRAMES[name]
    g = oracle.SimpleRace(**f["opts"])
    g.reset_game()
    cx, cy = measured_car(name)
    hits = {}
    for k in range(20):
        ang = float(np.float32((PI / 2 + k * PI / 10) % (2 * PI)))
        g.set_car(cx, cy, ang)
        s = g.screen()
        if abs(s[0] - float(f["printed"][1])) > 0.03 or abs(s[1] - float(f["printed"][2])) > 0.03:
            continue                                                 # not this heading, wherever within reach the car is
        for dx in np.arange(-reach, reach + 1e-9, step):
            for dy in np.arange(-reach, reach + 1e-9, step):
                g.set_car(cx + dx, cy + dy, ang)
                if printed_state(g.screen()) == f["printed"]:
                    hits.setdefault(k, []).append((cx + dx, cy + dy))
    return hits, (cx, cy)


def test_track_geometry_read_off_the_frames():
    """The flags the frames were made with, from the drawing itself (StraightTrack::draw / CircleTrack::draw, :65-70,120-170):
    straight: road base = start - (0.75 w, 10) .. end + (0.75 w, 30) with start = mid - 0.4 L, end = mid + 0.6 L, mid =
    (240, 360) -> L = 400, w = 80; circle: inner radius 80 about (240, 360), width 40."""
    a = _img("straight")
    brown = (a[..., 0] > 100) & (a[..., 2] < 60) & (a[..., 1] > 40) & (a[..., 1] < 110)
    ys, xs = np.nonzero(brown)
    x0, x1, y0, y1 = xs.min() * SCALE, (xs.max() + 1) * SCALE, ys.min() * SCALE, (ys.max() + 1) * SCALE
    assert abs(x0 - (240 - 0.75 * 80)) <= 1.6 and abs(x1 - (240 + 0.75 * 80)) <= 1.6
    assert abs(y0 - (360 - 0.4 * 400 - 10)) <= 1.6 and abs(y1 - (360 + 0.6 * 400 + 30)) <= 1.6
    a = _img("circle")
    brown = (a[..., 0] > 100) & (a[..., 2] < 60) & (a[..., 1] > 40) & (a[..., 1] < 110)
    ys, xs = np.nonzero(brown)
    assert abs(to_window(xs.min(), xs.max()) - 240) <= 1 and abs(to_window(ys.min(), ys.max()) - 360) <= 1
    assert abs((xs.max() + 1 - xs.min()) * SCALE / 2 - 80) <= 1.6
    gray = (abs(a[..., 0] - 105) <= 3) & (abs(a[..., 1] - 105) <= 3) & (abs(a[..., 2] - 105) <= 3)
    ys, xs = np.nonzero(gray & (np.arange(450)[:, None] > 100))       # (the text lines are above)
    assert abs((xs.max() + 1 - xs.min()) * SCALE / 2 - 120) <= 1.6


@pytest.mark.parametrize("name,k_expected", [("straight", 3), ("circle", 16)])
def test_get_screen_prints_what_the_reference_frame_shows(oracle, name, k_expected):
    hits, (cx, cy) = solve(oracle, name)
    assert list(hits) == [k_expected], (name, list(hits), (cx, cy))   # one heading: PI/2 + 3 PI/10 (straight), PI/2 - 4 PI/10 (circle)
    pts = np.array(hits[k_expected])
    assert len(pts) > 3                                               # a region, not a lucky grid point
    # the heading agrees with the arrow CircleCar::draw put on the car (white pixels next to the blue disc)
    a = _img(name)
    ang = (PI / 2 + k_expected * PI / 10) % (2 * PI)
    tip = np.array([cx + 7 * np.cos(ang), cy + 7 * np.sin(ang)])      # pos + 2 radius * (cos, sin): towards it
    ix, iy = int((tip[0] + 0.5) / SCALE), int((tip[1] + 0.5) / SCALE)
    near = a[iy - 2:iy + 3, ix - 2:ix + 3]
    assert (near.min(axis=2) > 150).any(), "no arrow where the heading points"


# Walks of the default two-action set (each action: turn by +-PI/10, then one unit forward; BaseCar::move, :227-235) from
# RaceEngine::reset_game's deterministic start that END in the frames' states.  Found by a search over the lattice the walks
# span (positions = start + sums of the twenty unit headings).  Circle: among ALL walks of <= 30 steps exactly one end
# point (heading PI/2 - 4 PI/10, 10 steps -- the fewest the distance allows -- reached by three orders of the same steps)
# prints the image's numbers; the region that prints them measures 0.2 x 0.02 px, the ~10^4 end points of such walks are
# spread over ~600 px^2, so this is the frame's own action history up to order, not a coincidence: it pins the start
# position and heading, the turn and forward step sizes and their order.  Straight: the region is larger (0.4 x 2 px); the
# 27-step walk below (the fewest steps the distance allows; one end point, 66 orders) is one of many that fit -- weak
# evidence, kept as a regression value.
WALKS = {"circle": [0, 1, 0, 1, 1, 0, 1, 1, 1, 1],
         "straight": [0, 0, 1, 0, 1, 0] + [0, 1] * 10 + [0]}


@pytest.mark.parametrize("name", ["circle", "straight"])
def test_a_walk_from_reset_ends_in_the_frame(oracle, name):
    f = FRAMES[name]
    g = oracle.SimpleRace(**f["opts"])
    g.reset_game()
    for a in WALKS[name]:
        g.take_actions(a)
        assert g.game_over() == 0
    assert printed_state(g.screen()) == f["printed"]
    cx, cy = measured_car(name)
    x, y, _ = g.car()
    assert np.hypot(x - cx, y - cy) < 1.5                             # where the image shows the car
    if name == "circle":                                             # every other order of a step changes the end point's print
        for i in range(len(WALKS[name])):
            alt = list(WALKS[name])
            alt[i] ^= 1
            g.reset_game()
            for a in alt:
                g.take_actions(a)
            assert printed_state(g.screen()) != f["printed"], i
