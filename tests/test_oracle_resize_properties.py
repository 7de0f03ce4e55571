"""Fences around the two stages of the C4 render that NO reference-held byte confirms.

`XWorldSimulator::down_sample_image` (xworld_simulator.cpp:508-545) calls cv::resize(INTER_LINEAR) and
`get_screen_rgb` (xworld_simulator.cpp:287-307) calls cvtColor(BGR2GRAY).  OpenCV 3.2 is not in this image and the
reference holds no down-sampled frame, so `orc_cv_resize_linear_8u` / `orc_cv_bgr2gray_8u` (oracle/xworld2d.c:656-722)
are restated from the library's published fixed-point algorithm and stay *unpinned_by_reference*.  These tests do NOT
pin them -- nothing can, here.  They make a silent change impossible: what OpenCV documents (identity on equal sizes,
constants stay constants, coefficient sums), what SURVEY.md 8(a) computed for the three frame sizes BASELINE names
(the 12 in-cell taps and their weights), and a frozen checksum of the tile table both sides of every C4 comparison
are built from."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tile_table_checksum.json")

# in-cell source offsets of the 12 output pixels of a 64 px cell and the cycling weights
S_K = [2, 7, 12, 18, 23, 28, 34, 39, 44, 50, 55, 60]
W_K = [(1707, 341), (1024, 1024), (341, 1707)]


def _resize(oracle, src, dh, dw):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), dtype=np.uint8)
    oracle.lib().orc_cv_resize_linear_8u(oracle.ptr(src, oracle.u8p), sh, sw, cn, oracle.ptr(dst, oracle.u8p), dh, dw)
    return dst


def _gray(oracle, bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    out = np.empty(bgr.shape[:-1], dtype=np.uint8)
    oracle.lib().orc_cv_bgr2gray_8u(oracle.ptr(bgr, oracle.u8p), int(out.size), oracle.ptr(out, oracle.u8p))
    return out


def _coeffs(src, dst):
    """cv::resize's own coefficient computation in float32 (imgproc/src/imgwarp.cpp, INTER_LINEAR, 8-bit)."""
    scale = 1.0 / (float(dst) / float(src))
    out = []
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        w1 = int(np.rint(np.float32(f * np.float32(2048))))
        w0 = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
        out.append((s, w0, w1))
    return out


def test_resize_identity_on_equal_size_unpinned_by_reference(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(_resize(oracle, img, 37, 53), img)


def test_resize_constant_in_constant_out_unpinned_by_reference(oracle):
    for v in (0, 1, 7, 127, 128, 200, 254, 255):
        for (sh, sw, dh, dw) in ((448, 448, 84, 84), (512, 512, 96, 96), (704, 704, 132, 132), (320, 320, 512, 512), (512, 512, 80, 80)):
            out = _resize(oracle, np.full((sh, sw, 3), v, dtype=np.uint8), dh, dw)
            assert (out == v).all(), (v, sh, dh)


def test_resize_taps_of_the_three_frame_sizes_unpinned_by_reference(oracle):
    """448 -> 84, 512 -> 96, 704 -> 132 (7x7, 8x8, 11x11 cells of 64 px): every cell uses the same 12 taps, inside the cell."""
    for cells in (7, 8, 11):
        src, dst = 64 * cells, 12 * cells
        co = _coeffs(src, dst)
        for d, (s, w0, w1) in enumerate(co):
            k = d % 12
            assert s == 64 * (d // 12) + S_K[k], (cells, d, s)
            assert (w0, w1) == W_K[k % 3], (cells, d, w0, w1)
            assert w0 + w1 == 2048
        # and the oracle's function applies exactly SURVEY 8(a)'s formula with these tables:
        # out[y][x] = V(H(row s_y)[x], H(row s_y + 1)[x]), H(row)[x] = src[row][s_x] * w0 + src[row][s_x + 1] * w1 (int32),
        # V(S0, S1) = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
        rng = np.random.default_rng(cells)
        img = rng.integers(0, 256, (src, src), dtype=np.uint8)
        s = np.array([64 * (d // 12) + S_K[d % 12] for d in range(dst)])
        w0 = np.array([W_K[(d % 12) % 3][0] for d in range(dst)], dtype=np.int64)
        w1 = np.array([W_K[(d % 12) % 3][1] for d in range(dst)], dtype=np.int64)
        im = img.astype(np.int64)
        H = im[:, s] * w0[None, :] + im[:, s + 1] * w1[None, :]
        S0, S1 = H[s, :], H[s + 1, :]
        exp = (((w0[:, None] * (S0 >> 4)) >> 16) + ((w1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2
        assert np.array_equal(_resize(oracle, img, dst, dst).astype(np.int64), exp), cells


def test_resize_never_mixes_neighbouring_cells_unpinned_by_reference(oracle):
    """What makes the tile table exact: a 64 px cell's 12 output pixels depend on that cell only."""
    rng = np.random.default_rng(2)
    for cells in (7, 8, 11):
        a = rng.integers(0, 256, (64 * cells, 64 * cells, 3), dtype=np.uint8)
        b = a.copy()
        cy, cx = cells // 2, cells - 1
        b[64 * cy:64 * cy + 64, 64 * cx:64 * cx + 64] = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
        oa, ob = _resize(oracle, a, 12 * cells, 12 * cells), _resize(oracle, b, 12 * cells, 12 * cells)
        diff = (oa != ob).any(axis=2)
        outside = diff.copy()
        outside[12 * cy:12 * cy + 12, 12 * cx:12 * cx + 12] = False
        assert not outside.any() and diff.any()


def test_bgr2gray_weights_unpinned_by_reference(oracle):
    assert 1868 + 9617 + 4899 == 1 << 14
    rng = np.random.default_rng(3)
    px = rng.integers(0, 256, (4096, 3), dtype=np.uint8)
    exp = ((px[:, 0].astype(np.int64) * 1868 + px[:, 1].astype(np.int64) * 9617 + px[:, 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(_gray(oracle, px), exp)
    for v in range(256):                                     # gray in -> the same gray out
        assert int(_gray(oracle, np.full((1, 3), v, dtype=np.uint8))[0]) == v
    assert int(_gray(oracle, np.array([[255, 0, 0]], dtype=np.uint8))[0]) == (255 * 1868 + 8192) >> 14
    assert int(_gray(oracle, np.array([[0, 255, 0]], dtype=np.uint8))[0]) == (255 * 9617 + 8192) >> 14
    assert int(_gray(oracle, np.array([[0, 0, 255]], dtype=np.uint8))[0]) == (255 * 4899 + 8192) >> 14


def oracle_tile_table(oracle):
    """The 363 x 3 x 12 x 12 colour tile table and its gray twin, from the shipped atlas through the oracle's resize."""
    icons = np.load(os.path.join(os.path.dirname(GOLDEN), "..", "..", "xworld_amd", "assets", "icons64.npz"))["icons"]
    col = np.empty((icons.shape[0], 3, 12, 12), dtype=np.uint8)
    gray = np.empty((icons.shape[0], 1, 12, 12), dtype=np.uint8)
    for i in range(icons.shape[0]):
        t = _resize(oracle, icons[i], 12, 12)
        col[i] = t.transpose(2, 0, 1)
        gray[i, 0] = _gray(oracle, t)
    return col, gray


def test_tile_table_checksum_frozen_unpinned_by_reference(oracle):
    """The frozen checksum: a change of the atlas, of the resize restatement or of the gray conversion shows up HERE, by
    name, rather than as 'both sides moved together'.  (tests/golden/tile_table_checksum.json is this function's own
    output at round 5 -- a fence, not a pin: the reference holds no such table.)"""
    col, gray = oracle_tile_table(oracle)
    got = {"shape_color": list(col.shape), "sha256_color": hashlib.sha256(col.tobytes()).hexdigest(),
           "shape_gray": list(gray.shape), "sha256_gray": hashlib.sha256(gray.tobytes()).hexdigest()}
    if os.environ.get("XWB_WRITE_GOLDEN"):
        with open(GOLDEN, "w") as f:
            json.dump({"unpinned_by_reference": True, "made_by": "tests/test_oracle_resize_properties.py (XWB_WRITE_GOLDEN=1)", **got}, f, indent=1)
    with open(GOLDEN) as f:
        exp = json.load(f)
    for k, v in got.items():
        assert exp[k] == v, k
