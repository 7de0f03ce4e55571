// xworld_amd/csrc/kernels_xworld_ego.hip -- the egocentric observation of XWorld2D (FLAGS_visible_radius = r > 0).
//
// Reference pipeline, per env and per call (all on 8-bit BGR images):
//   XMap::to_image            xmap.cpp:125-206    world canvas from 64x64 item images, r cells of black padding, crop of
//                                                 the r x r cells in front of the agent, wall shadows (image_masking,
//                                                 :273-362) painted black, rotation by 90 + yaw degrees (cv::warpAffine)
//   XItem::get_item_image     xitem.cpp:33-63     every item image is warped by its own (yaw, scale, offset)
//   get_screen_rgb            xworld_simulator.cpp:287-307   cv::resize of the (64 r)^2 view to the (64 max_dim)^2 canvas size
//   down_sample_image         :508-545            cv::resize to (r * (84 / r))^2, optional BGR2GRAY, planar output
// Nothing here is materialised except the final frame: every output pixel is the fixed-point bilinear blend
// (cv::resize: 11-bit coefficients, the intermediate image rounded to 8 bits exactly as OpenCV does) of 2 x 2 pixels of
// the intermediate image, each of which blends 2 x 2 view pixels; a view pixel is found by undoing the quarter-turn view
// rotation (exact integer map, one border row / column), the cell lookup, and for goals the inverse affine warp with
// cv::remap's 5-bit sub-pixel bilinear weights.  One workgroup renders one env; the frame is assembled in LDS and
// leaves as aligned 16-byte stores.  Work per env: (84^2) x 16 view-pixel evaluations -- compute-bound, not HBM-bound
// (21 KB written per env); icons (64x64x3, 4.2 MB for the XWorldNav palette) are read through L2.
//
// OpenCV 3.2 arithmetic restated (third party, cmake/opencv.cmake:5-6; DESIGN.md lists the pieces): the tests compare
// this kernel bit for bit with a CPU restatement of the same pipeline; pixel parity with the real library is unpinned.
#include "xwb_common.h"
#include "xw_device.h"

#include <cmath>
#include <vector>

namespace xwb {

struct EgoTap { int16_t s0, s1, w0, w1; };        // cv::resize: source indices and 11-bit weights of one output index

namespace {

struct EgoCell { uint8_t kind; uint8_t aux; uint16_t icon; };       // kind: 0 black, 1 white, 2 block icon, 3 agent (aux = heading), 4 goal (aux = slot)

struct EgoCtx {
    const uint8_t *atlas64;
    const EgoCell *cells;        // LDS, r * r
    const double *warp;          // LDS, [slot][6]
    int r, S, dir;
};

__device__ __forceinline__ void icon_px(const uint8_t *atlas64, int icon, int x, int y, int &b, int &g, int &rr) {
    const uint8_t *q = atlas64 + (((size_t)icon * 64 + y) * 64 + x) * 3;
    b = q[0]; g = q[1]; rr = q[2];
}

// one pixel of the rotated view: BGR
__device__ __forceinline__ void view_px(const EgoCtx &c, int vy, int vx, int &b, int &g, int &rr) {
    // undo cv::warpAffine(view, rot(centre S/2, 90 + yaw deg)): quarter turns are exact integer maps; the source index S
    // falls outside, which leaves one black row / column (borderValue 0)
    const int S = c.S;
    int sx, sy;
    switch (c.dir) {
        case 3: sx = vx; sy = vy; break;                   // heading up: 0 deg
        case 0: sx = S - vy; sy = vx; break;               // right: 90 deg
        case 1: sx = S - vx; sy = S - vy; break;           // down: 180 deg
        default: sx = vy; sy = S - vx; break;              // left: 270 deg
    }
    b = g = rr = 0;
    if ((unsigned)sx >= (unsigned)S || (unsigned)sy >= (unsigned)S) return;
    const EgoCell cell = c.cells[(sy >> 6) * c.r + (sx >> 6)];
    const int px = sx & 63, py = sy & 63;
    if (cell.kind == 0) return;
    if (cell.kind == 1) { b = g = rr = 255; return; }
    if (cell.kind == 2) { icon_px(c.atlas64, cell.icon, px, py, b, g, rr); return; }
    if (cell.kind == 3) {
        // XItem::get_item_image for the agent: rotation by 90 - yaw deg about (32, 32), border white
        int ix, iy;
        switch (cell.aux) {
            case 1: ix = px; iy = py; break;               // down: 0 deg
            case 0: ix = 64 - py; iy = px; break;          // right: 90 deg
            case 3: ix = 64 - px; iy = 64 - py; break;     // up: 180 deg
            default: ix = py; iy = 64 - px; break;         // left: -90 deg
        }
        if ((unsigned)ix >= 64u || (unsigned)iy >= 64u) { b = g = rr = 255; return; }
        icon_px(c.atlas64, cell.icon, ix, iy, b, g, rr);
        return;
    }
    // goal: cv::warpAffine with the stored inverse matrix, INTER_LINEAR, BORDER_CONSTANT white
    const double *M = c.warp + cell.aux * 6;
    const int X0 = __double2int_rn((M[1] * py + M[2]) * 1024) + 16, Y0 = __double2int_rn((M[4] * py + M[5]) * 1024) + 16;
    const int X = (X0 + __double2int_rn(M[0] * px * 1024)) >> 5, Y = (Y0 + __double2int_rn(M[3] * px * 1024)) >> 5;
    const int ix = X >> 5, iy = Y >> 5, fx = X & 31, fy = Y & 31;
    if (ix >= 64 || ix + 1 < 0 || iy >= 64 || iy + 1 < 0) { b = g = rr = 255; return; }
    int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
    if (w0 == 32768) { w0 = 32767; w3 = 1; }               // BilinearTab_i: saturated entry and its compensation
    int pb[4], pg[4], pr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int qx = ix + (k & 1), qy = iy + (k >> 1);
        if ((unsigned)qx < 64u && (unsigned)qy < 64u) icon_px(c.atlas64, cell.icon, qx, qy, pb[k], pg[k], pr[k]);
        else pb[k] = pg[k] = pr[k] = 255;
    }
    b = (pb[0] * w0 + pb[1] * w1 + pb[2] * w2 + pb[3] * w3 + (1 << 14)) >> 15;
    g = (pg[0] * w0 + pg[1] * w1 + pg[2] * w2 + pg[3] * w3 + (1 << 14)) >> 15;
    rr = (pr[0] * w0 + pr[1] * w1 + pr[2] * w2 + pr[3] * w3 + (1 << 14)) >> 15;
}

// cv::resize INTER_LINEAR on 8-bit data, one output pixel: HResizeLinear (11-bit) then VResizeLinear<uchar>
__device__ __forceinline__ int vresize(int b0, int h0, int b1, int h1) {
    return ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
}

// one pixel of the intermediate (canvas-sized) image
__device__ __forceinline__ void mid_px(const EgoCtx &c, const EgoTap &ty, const EgoTap &tx, int &b, int &g, int &rr) {
    int hb[2], hg[2], hr[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = k ? ty.s1 : ty.s0;
        int b0, g0, r0, b1, g1, r1;
        view_px(c, row, tx.s0, b0, g0, r0);
        if (tx.s1 != tx.s0) view_px(c, row, tx.s1, b1, g1, r1); else { b1 = b0; g1 = g0; r1 = r0; }
        hb[k] = b0 * tx.w0 + b1 * tx.w1; hg[k] = g0 * tx.w0 + g1 * tx.w1; hr[k] = r0 * tx.w0 + r1 * tx.w1;
    }
    b = vresize(ty.w0, hb[0], ty.w1, hb[1]);
    g = vresize(ty.w0, hg[0], ty.w1, hg[1]);
    rr = vresize(ty.w0, hr[0], ty.w1, hr[1]);
}

}  // namespace

// MODE 0: every env; 1: the compacted done list; 2: every env whose done code is 0 (step_autoreset)
template <int CH, int MODE>
__global__ __launch_bounds__(256) void xw_render_ego_kernel(XwParams p, const uint8_t *atlas64, const EgoTap *tap_h1,
                                                            const EgoTap *tap_v1, const EgoTap *tap_h2, const EgoTap *tap_v2,
                                                            const int32_t *count_now) {
    extern __shared__ uint4 smem4[];
    const int r = p.visible_radius, S = 64 * r, D = p.max_dim, O = p.out_dim;
    uint8_t *s_frame = reinterpret_cast<uint8_t *>(smem4);                       // CH * O * O, planar
    double *s_warp = reinterpret_cast<double *>(s_frame + ((CH * O * O + 15) & ~15));
    EgoCell *s_cells = reinterpret_cast<EgoCell *>(s_warp + XW_MAX_GOALS * 6);
    uint8_t *s_shadow = reinterpret_cast<uint8_t *>(s_cells + r * r);
    uint8_t *s_ray = s_shadow + r * r;
    uint8_t *s_gc = s_ray + ((r + 3) & ~3);
    __shared__ int s_geo[4];
    const int tid = threadIdx.x;
    const int cells = D * D;
    const int cpf = CH * O * O / 16;
    const int n_items = MODE == 1 ? *count_now : p.n;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int e = MODE == 1 ? p.done_list[item] : item;
        if (MODE == 2 && p.done[e] != 0) continue;
        const uint16_t *grid = p.grid + (size_t)e * cells;
        const int axy = p.agent_xy[e], ax = axy & 0xffff, ay = axy >> 16;
        const int dir = p.agent_dir[e];
        __syncthreads();
        if (tid < XW_MAX_GOALS) s_gc[tid] = p.goal_cells[(size_t)e * XW_MAX_GOALS + tid];
        if (tid < XW_MAX_GOALS * 6) s_warp[tid] = p.goal_warp[(size_t)e * XW_MAX_GOALS * 6 + tid];
        auto is_block = [&](int x, int y) {
            if ((unsigned)x >= (unsigned)D || (unsigned)y >= (unsigned)D) return false;
            const int code = grid[y * D + x] & CELL_ICON_MASK;
            return code != 0 && p.icon_type[code - 1] == 1;
        };
        // XMap::image_masking (xmap.cpp:273-362)
        int major_x = 0, major_y = 0, minor_x = 0, minor_y = 0, scan_x0 = 0, scan_y0 = 0, xa = ax + r, ya = ay + r;
        if (dir == 0) { xa += r / 2; major_y = 1; minor_x = 1; }
        else if (dir == 3) { ya -= r / 2; major_x = 1; minor_y = -1; scan_y0 = r - 1; }
        else if (dir == 2) { xa -= r / 2; major_y = 1; minor_x = -1; scan_x0 = r - 1; }
        else { ya += r / 2; major_x = 1; minor_y = 1; }
        const int x_st = xa - r / 2, y_st = ya - r / 2;
        if (tid < r) s_ray[tid] = 1;
        __syncthreads();
        if (tid < 2) {                                          // rays to either side of the agent
            const int o = tid ? 1 : -1;
            bool block = false;
            int rx = ax, ry = ay;
            for (int k = 1; k <= r / 2; ++k) {
                rx += o * major_x; ry += o * major_y;
                if (block) s_ray[r / 2 + o * k] = 0;
                if (is_block(rx, ry)) block = true;
            }
        }
        __syncthreads();
        if (tid < r) {                                          // one scan line per lane
            bool block = !s_ray[tid];
            int cx = scan_x0 + tid * major_x, cy = scan_y0 + tid * major_y;
            for (int j = 0; j < r; ++j) {
                s_shadow[cy * r + cx] = block ? 1 : 0;
                if (is_block(x_st - r + cx, y_st - r + cy)) block = true;
                cx = (cx + minor_x + r) % r;
                cy = (cy + minor_y + r) % r;
            }
        }
        __syncthreads();
        for (int k = tid; k < r * r; k += 256) {                // what each view cell shows
            const int gx = x_st - r + k % r, gy = y_st - r + k / r;
            EgoCell c{0, 0, 0};
            if ((unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && !s_shadow[k]) {
                const int code = grid[gy * D + gx] & CELL_ICON_MASK;
                if (code == 0) c.kind = 1;
                else {
                    c.icon = (uint16_t)(code - 1);
                    const int t = p.icon_type[code - 1];
                    if (t == 1) c.kind = 2;
                    else if (t == 2) { c.kind = 3; c.aux = (uint8_t)dir; }
                    else {
                        c.kind = 4;
                        int slot = 0;
                        for (int i = 0; i < XW_MAX_GOALS; ++i) if (s_gc[i] == gy * D + gx) slot = i;
                        c.aux = (uint8_t)slot;
                    }
                }
            }
            s_cells[k] = c;
        }
        __syncthreads();
        EgoCtx ctx{atlas64, s_cells, s_warp, r, S, dir};
        for (int o = tid; o < O * O; o += 256) {
            const int oy = o / O, ox = o - oy * O;
            const EgoTap ty = tap_v2[oy], tx = tap_h2[ox];
            int hb[2], hg[2], hr[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const EgoTap my = tap_v1[k ? ty.s1 : ty.s0];
                int b0, g0, r0, b1, g1, r1;
                mid_px(ctx, my, tap_h1[tx.s0], b0, g0, r0);
                if (tx.s1 != tx.s0) mid_px(ctx, my, tap_h1[tx.s1], b1, g1, r1); else { b1 = b0; g1 = g0; r1 = r0; }
                hb[k] = b0 * tx.w0 + b1 * tx.w1; hg[k] = g0 * tx.w0 + g1 * tx.w1; hr[k] = r0 * tx.w0 + r1 * tx.w1;
            }
            const int b = vresize(ty.w0, hb[0], ty.w1, hb[1]);
            const int g = vresize(ty.w0, hg[0], ty.w1, hg[1]);
            const int rr = vresize(ty.w0, hr[0], ty.w1, hr[1]);
            if (CH == 3) {
                s_frame[o] = (uint8_t)b; s_frame[O * O + o] = (uint8_t)g; s_frame[2 * O * O + o] = (uint8_t)rr;
            } else {
                s_frame[o] = (uint8_t)((b * 1868 + g * 9617 + rr * 4899 + (1 << 13)) >> 14);   // cvtColor BGR2GRAY
            }
        }
        __syncthreads();
        const int flag = MODE == 1 ? 2 : p.fresh[e];
        uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * p.context * cpf;
        for (int cc = tid; cc < cpf; cc += 256) xw_store_chunk(frame0, cc, cpf, p.context, p.context > 1 ? flag : 1, smem4[cc]);
        if (MODE == 1 && tid == 0) p.fresh[e] = 0;
    }
}

// ------------------------------------------------------------------------------------------------- host side ----
static void resize_taps(int src, int dst, std::vector<EgoTap> &h, std::vector<EgoTap> &v) {
    // cv::resize INTER_LINEAR (imgwarp.cpp): fx = (dx + 0.5) * scale - 0.5 in float; left edge: sx < 0 -> (0, fx = 0);
    // right edge: columns from the first one with sx + 1 >= src on take the single tap S[min(sx, src - 1)] * 2048;
    // rows are clipped instead; coefficients = cvRound(c * 2048) as short
    const double scale = (double)src / dst;
    h.resize(dst); v.resize(dst);
    int xmax = dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        {   // vertical
            const float fy = f;
            const int r0 = s < 0 ? 0 : (s >= src ? src - 1 : s), r1 = s + 1 < 0 ? 0 : (s + 1 >= src ? src - 1 : s + 1);
            v[d] = EgoTap{(int16_t)r0, (int16_t)r1, (int16_t)lrintf((1.f - fy) * 2048), (int16_t)lrintf(fy * 2048)};
        }
        float fx = f;
        int sx = s;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= src) {
            if (d < xmax) xmax = d;
            if (sx >= src - 1) { fx = 0; sx = src - 1; }
        }
        if (d >= xmax) h[d] = EgoTap{(int16_t)sx, (int16_t)sx, 2048, 0};
        else h[d] = EgoTap{(int16_t)sx, (int16_t)(sx + 1), (int16_t)lrintf((1.f - fx) * 2048), (int16_t)lrintf(fx * 2048)};
    }
}

hipError_t xw_ego_tables(int r, int max_dim, int out_dim, EgoTap **dev_out /* h1, v1, h2, v2 contiguous */) {
    std::vector<EgoTap> h1, v1, h2, v2;
    resize_taps(64 * r, 64 * max_dim, h1, v1);
    resize_taps(64 * max_dim, out_dim, h2, v2);
    std::vector<EgoTap> all;
    all.insert(all.end(), h1.begin(), h1.end()); all.insert(all.end(), v1.begin(), v1.end());
    all.insert(all.end(), h2.begin(), h2.end()); all.insert(all.end(), v2.begin(), v2.end());
    EgoTap *d = nullptr;
    hipError_t err = hipMalloc(&d, all.size() * sizeof(EgoTap));
    if (err != hipSuccess) return err;
    err = hipMemcpy(d, all.data(), all.size() * sizeof(EgoTap), hipMemcpyHostToDevice);
    *dev_out = d;
    return err;
}

hipError_t launch_xw_render_ego(const XwParams &p, int indexed, hipStream_t s) {
    const int r = p.visible_radius, O = p.out_dim, P = 64 * p.max_dim;
    const EgoTap *h1 = reinterpret_cast<const EgoTap *>(p.ego_taps), *v1 = h1 + P, *h2 = v1 + P, *v2 = h2 + O;
    const int CH = p.channels;
    const size_t lds = (size_t)((CH * O * O + 15) & ~15) + XW_MAX_GOALS * 6 * sizeof(double) + (size_t)r * r * (sizeof(EgoCell) + 1) +
                       (size_t)((r + 3) & ~3) + XW_MAX_GOALS + 16;
    const unsigned blocks = indexed == 1 ? 2048u : (unsigned)(p.n < 16384 ? p.n : 16384);
    const int32_t *cnt = (const int32_t *)p.done_count;
#define EGO_LAUNCH(CHV, MODEV) hipLaunchKernelGGL((xw_render_ego_kernel<CHV, MODEV>), dim3(blocks), dim3(256), lds, s, p, p.atlas64, h1, v1, h2, v2, cnt)
    if (CH == 3) { if (indexed == 1) EGO_LAUNCH(3, 1); else if (indexed == 2) EGO_LAUNCH(3, 2); else EGO_LAUNCH(3, 0); }
    else { if (indexed == 1) EGO_LAUNCH(1, 1); else if (indexed == 2) EGO_LAUNCH(1, 2); else EGO_LAUNCH(1, 0); }
#undef EGO_LAUNCH
    return hipGetLastError();
}

}  // namespace xwb
