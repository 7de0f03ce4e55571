"""include/xwb_trig.h, the deterministic sin / cos of the HIP kernels, against the host's libm (the oracle's default) --
what replacing the reference's libm calls (simple_race_simulator.cpp:227-243,386-430; xitem.cpp:47-60 -> cv::getRotationMatrix2D)
changes on the arguments the path can produce.  CPU only."""
import ctypes as C
import math

import numpy as np


def _sincos(L, xs):
    s, c = C.c_double(), C.c_double()
    out = np.empty((len(xs), 2), np.float64)
    for i, x in enumerate(xs):
        L.orc_xwb_sincos(float(x), C.byref(s), C.byref(c))
        out[i] = (s.value, c.value)
    return out


def test_sincos_against_libm(oracle):
    """Sub-ulp agreement in double, exact agreement once narrowed to float (what SimpleRace does with every result)."""
    L = oracle.lib()
    rng = np.random.default_rng(0)
    # SimpleRace angles: float32(u * 2 * PI) for 24-bit u (set_angle(true)), plus k * float32(PI / 10) walks with wraps
    u = (rng.integers(0, 1 << 24, 60000).astype(np.float32) * np.float32(1 / 16777216.0))
    ang = (u * np.float32(2)).astype(np.float64) * 3.1415926
    xs = list(ang.astype(np.float32).astype(np.float64))
    a = np.float32(3.1415926 / 2)
    for k in range(20000):                                  # BaseCar::move: float += float, double compare, wrap
        a = np.float32(a + (np.float32(3.1415926 / 10) if (k * 7919) % 3 else -np.float32(3.1415926 / 10)))
        if float(a) > 2 * 3.1415926:
            a = np.float32(float(a) - 2 * 3.1415926)
        elif a < 0:
            a = np.float32(float(a) + 2 * 3.1415926)
        xs.append(float(a))
    # goal warps: angle = (90 - yaw * 180 / pi) * CV_PI / 180 with yaw in [0, 4 * 1.5707963)
    yaw = rng.random(40000) * 4 * 1.5707963
    xs += list((90 - yaw * 180 / math.pi) * 3.1415926535897932384626433832795 / 180)
    xs += [0.0, -0.0, 1e-300, math.pi / 4, -math.pi / 4, math.pi / 2, 1e5, -1e5, 123456.789]
    got = _sincos(L, xs)
    ref = np.array([(math.sin(x), math.cos(x)) for x in xs])
    ulp = np.spacing(np.abs(ref))
    assert (np.abs(got - ref) <= ulp).all()                                  # never more than one ulp apart
    differ = float((got != ref).mean())
    print("last-bit differences vs libm: %.3f %% of %d results" % (100 * differ, got.size))
    assert differ < 0.06
    assert np.array_equal(got.astype(np.float32), ref.astype(np.float32))    # narrowed to float: identical
    assert math.copysign(1, got[xs.index(-0.0)][0]) in (1.0, -1.0) and got[0 + xs.index(0.0)][1] == 1.0


def test_simple_race_same_bits_with_libm(oracle):
    """SimpleRace rollouts (straight / circle, random starts, full manoeuvre, hard): every reward bit, game-over code and
    observation checksum is the same whether the oracle's cos / sin come from xwb_trig.h or from libm."""
    L = oracle.lib()
    cfgs = [dict(), dict(random=1), dict(track_type=1), dict(track_type=1, random=1, race_full_manouver=1, difficulty_hard=1),
            dict(random=1, race_full_manouver=1)]
    try:
        for kw in cfgs:
            L.orc_set_trig_libm(0)
            a = oracle.race_rollout(512, oracle.race_cfg(**kw), seed=3, steps=400, policy_seed=5)
            L.orc_set_trig_libm(1)
            b = oracle.race_rollout(512, oracle.race_cfg(**kw), seed=3, steps=400, policy_seed=5)
            assert np.array_equal(a.rewards.view(np.uint32), b.rewards.view(np.uint32)), kw
            assert np.array_equal(a.codes, b.codes) and np.array_equal(a.obs_ck, b.obs_ck), kw
            assert a.stats.resets > 100
    finally:
        L.orc_set_trig_libm(1)
    assert L.orc_get_trig_libm() == 1


def test_goal_warps_same_pixels_with_libm(oracle):
    """XItem::get_item_image for random poses: the warped 64x64 icon is the same, pixel for pixel, with either cos / sin."""
    L = oracle.lib()
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    border = np.array([255, 255, 255], np.uint8)
    M = (C.c_double * 6)()

    def warp(yaw, scale, offset):
        L.orc_cv_get_rotation_matrix_2d(32.0, 32.0, 90 - yaw * 180 / math.pi, scale, M)
        M[2] += (offset + scale / 2 - 0.5) * 64
        M[5] += (offset + scale / 2 - 0.5) * 64
        dst = np.zeros_like(src)
        L.orc_cv_warp_affine_8uc3(src.ctypes.data_as(oracle.u8p), 64, 64, dst.ctypes.data_as(oracle.u8p), 64, 64, M,
                                  border.ctypes.data_as(oracle.u8p))
        return dst, list(M)
    try:
        mdiff = 0
        for _ in range(1500):
            yaw = rng.random() * 4 * 1.5707963
            scale = 0.5 + 0.5 * rng.random()
            offset = (1 - scale) * rng.random()
            L.orc_set_trig_libm(0)
            a, ma = warp(yaw, scale, offset)
            L.orc_set_trig_libm(1)
            b, mb = warp(yaw, scale, offset)
            assert np.array_equal(a, b), (yaw, scale, offset)
            mdiff += ma != mb
        print("rotation matrices that differ in some last bit: %d of 1500" % mdiff)
    finally:
        L.orc_set_trig_libm(1)
