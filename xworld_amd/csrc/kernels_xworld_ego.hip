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

// What one cell of the view shows: a 64 x 64 image (block icon, this env's warped goal image, the agent icon turned for
// its heading -- the three turned copies of every agent icon are appended to the atlas at create time) or one constant
// pixel (mask = 0).  The table makes the per-pixel lookup branch-free: one 16-byte LDS read, an AND and an add.
struct EgoCell {
    const uint32_t *img;
    int mask;                    // -1: index the image; 0: a constant pixel
    int pad;
};

struct EgoCtx {
    const EgoCell *cells;        // LDS, r * r
    const uint32_t *white, *black;
    int r, S;
};

// cv::resize INTER_LINEAR on 8-bit data, one output value: HResizeLinear (11-bit) then VResizeLinear<uchar>
__device__ __forceinline__ int vresize(int b0, int h0, int b1, int h1) {
    // operands < 2^24 and products < 2^31: v_mul_u32_u24 is exact and full rate
    return (int)((((__umul24((unsigned)b0, (unsigned)(h0 >> 4))) >> 16) + ((__umul24((unsigned)b1, (unsigned)(h1 >> 4))) >> 16) + 2u) >> 2);
}

// All pixels of one frame.  DIR = the agent's heading: cv::warpAffine(view, rot(centre S/2, 90 + yaw deg)) is undone
// per tap row / column -- quarter turns are exact integer maps, separable in x and y; the source index S falls outside
// and leaves one black row / column (borderValue 0).
template <int CH, int DIR, int BS>
__device__ __forceinline__ void ego_pixels(const EgoCtx &c, const EgoTap (*s_row)[3], const EgoTap (*s_col)[3],
                                           uint8_t *s_frame, int O, int tid) {
    const int S = c.S;
    for (int o = tid; o < O * O; o += BS) {
        const int oy = o / O, ox = o - oy * O;
        // the 2 x 2 intermediate pixels this output pixel blends, and the 4 x 4 view pixels behind them
        const EgoTap ty = s_row[oy][2], tx = s_col[ox][2];
        const EgoTap my[2] = {s_row[oy][0], s_row[oy][1]}, mx[2] = {s_col[ox][0], s_col[ox][1]};
        const int R[4] = {my[0].s0, my[0].s1, my[1].s0, my[1].s1}, C[4] = {mx[0].s0, mx[0].s1, mx[1].s0, mx[1].s1};
        // source coordinate contributed by a view row (vr) and by a view column (vc):
        //   up (3): sx = vc, sy = vr;  right (0): sx = S - vr, sy = vc;  down (1): sx = S - vc, sy = S - vr;  left (2): sx = vr, sy = S - vc
        int fr[4], fc[4];                                   // coordinate from the row index, from the column index
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fr[i] = (DIR == 3 || DIR == 2) ? R[i] : S - R[i];
            fc[i] = (DIR == 3 || DIR == 0) ? C[i] : S - C[i];
        }
        // fr is sy for headings up / down and sx for right / left (and fc the other one)
        constexpr bool ROW_IS_Y = DIR == 3 || DIR == 1;
        int cr[4], cc[4], pr[4], pc[4];
        bool okr[4], okc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            okr[i] = (unsigned)fr[i] < (unsigned)S; okc[i] = (unsigned)fc[i] < (unsigned)S;
            cr[i] = ROW_IS_Y ? __mul24(fr[i] >> 6, c.r) : (fr[i] >> 6);
            cc[i] = ROW_IS_Y ? (fc[i] >> 6) : __mul24(fc[i] >> 6, c.r);
            pr[i] = fr[i] & 63; pc[i] = fc[i] & 63;
        }
        const uint32_t *src[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = k >> 2, j = k & 3;
            const bool inview = okr[i] && okc[j];
            const EgoCell cell = c.cells[inview ? cr[i] + cc[j] : 0];
            const int px = ROW_IS_Y ? pc[j] : pr[i], py = ROW_IS_Y ? pr[i] : pc[j];
            const uint32_t *q = cell.img + ((py * 64 + px) & cell.mask);
            src[k] = inview ? q : c.black;
        }
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *src[k];
        int out[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            int hB[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {                  // intermediate row a
                int A[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {              // intermediate column b
                    const int h0 = __mul24((int)((v[(2 * a) * 4 + 2 * b] >> (8 * ch)) & 255u), mx[b].w0) +
                                   __mul24((int)((v[(2 * a) * 4 + 2 * b + 1] >> (8 * ch)) & 255u), mx[b].w1);
                    const int h1 = __mul24((int)((v[(2 * a + 1) * 4 + 2 * b] >> (8 * ch)) & 255u), mx[b].w0) +
                                   __mul24((int)((v[(2 * a + 1) * 4 + 2 * b + 1] >> (8 * ch)) & 255u), mx[b].w1);
                    A[b] = vresize(my[a].w0, h0, my[a].w1, h1);
                }
                hB[a] = __mul24(A[0], tx.w0) + __mul24(A[1], tx.w1);
            }
            out[ch] = vresize(ty.w0, hB[0], ty.w1, hB[1]);
        }
        if (CH == 3) {
            s_frame[o] = (uint8_t)out[0]; s_frame[O * O + o] = (uint8_t)out[1]; s_frame[2 * O * O + o] = (uint8_t)out[2];
        } else {
            s_frame[o] = (uint8_t)((out[0] * 1868 + out[1] * 9617 + out[2] * 4899 + (1 << 13)) >> 14);   // cvtColor BGR2GRAY
        }
    }
}

}  // namespace

// MODE 0: every env; 1: the compacted done list; 2: every env whose done code is 0 (step_autoreset)
// BS threads per workgroup: 256 for the whole batch; 1024 for the short done list, where the latency of one env counts
template <int CH, int MODE, int BS>
__global__ __launch_bounds__(BS) void xw_render_ego_kernel(XwParams p, const uint32_t *atlas4, const EgoTap *tap_h1,
                                                            const EgoTap *tap_v1, const EgoTap *tap_h2, const EgoTap *tap_v2,
                                                            const int32_t *count_now) {
    extern __shared__ uint4 smem4[];
    const int r = p.visible_radius, S = 64 * r, D = p.max_dim, O = p.out_dim;
    uint8_t *s_frame = reinterpret_cast<uint8_t *>(smem4);                       // CH * O * O, planar
    EgoCell *s_cells = reinterpret_cast<EgoCell *>(s_frame + ((CH * O * O + 15) & ~15));
    uint8_t *s_shadow = reinterpret_cast<uint8_t *>(s_cells + r * r);
    uint8_t *s_ray = s_shadow + r * r;
    uint8_t *s_gc = s_ray + ((r + 3) & ~3);
    // composed taps of one output row / column: the two intermediate indices' taps and the output tap (static: O <= 84)
    __shared__ EgoTap s_row[84][3], s_col[84][3];
    for (int i = threadIdx.x; i < O; i += BS) {
        const EgoTap ty = tap_v2[i], tx = tap_h2[i];
        s_row[i][0] = tap_v1[ty.s0]; s_row[i][1] = tap_v1[ty.s1]; s_row[i][2] = ty;
        s_col[i][0] = tap_h1[tx.s0]; s_col[i][1] = tap_h1[tx.s1]; s_col[i][2] = tx;
    }
    const int tid = threadIdx.x;
    const int cells = D * D;
    const int cpf = CH * O * O / (p.obs_f32 ? 4 : 16);    // 16-byte chunks per frame: 16 uint8 pixels, or 4 float32 ones
    const int n_items = MODE == 1 ? *count_now : p.n;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int e = MODE == 1 ? p.done_list[item] : item;
        if (MODE == 2 && p.done[e] != 0) continue;
        const uint16_t *grid = p.grid + (size_t)e * cells;
        const int axy = p.agent_xy[e], ax = axy & 0xffff, ay = axy >> 16;
        const int dir = p.agent_dir[e];
        __syncthreads();
        if (tid < XW_MAX_GOALS) s_gc[tid] = p.goal_cells[(size_t)e * XW_MAX_GOALS + tid];
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
        const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
        const uint32_t *gimg = p.goal_img + (size_t)e * p.num_goals * 4096;
        for (int k = tid; k < r * r; k += BS) {                 // what each view cell shows
            const int gx = x_st - r + k % r, gy = y_st - r + k / r;
            EgoCell c{black, 0, 0};                             // outside the map, or in a wall's shadow
            if ((unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && !s_shadow[k]) {
                const int code = grid[gy * D + gx] & CELL_ICON_MASK;
                if (code == 0) c.img = white;
                else {
                    const int t = p.icon_type[code - 1];
                    c.img = atlas4 + (size_t)(code - 1) * 4096;
                    c.mask = -1;
                    if (t == 2) {                               // the agent: XItem::get_item_image turns its icon by 90 - yaw deg
                        if (dir != 1) c.img = atlas4 + p.ego_agent_rot[code - 1] + (size_t)(dir == 0 ? 0 : (dir == 2 ? 1 : 2)) * 4096;
                    } else if (t == 0) {
                        int slot = 0;
                        for (int i = 0; i < XW_MAX_GOALS; ++i) if (s_gc[i] == gy * D + gx) slot = i;
                        c.img = gimg + slot * 4096;
                    }
                }
            }
            s_cells[k] = c;
        }
        __syncthreads();
        EgoCtx ctx{s_cells, white, black, r, S};
        switch (dir) {
            case 0: ego_pixels<CH, 0, BS>(ctx, s_row, s_col, s_frame, O, tid); break;
            case 1: ego_pixels<CH, 1, BS>(ctx, s_row, s_col, s_frame, O, tid); break;
            case 2: ego_pixels<CH, 2, BS>(ctx, s_row, s_col, s_frame, O, tid); break;
            default: ego_pixels<CH, 3, BS>(ctx, s_row, s_col, s_frame, O, tid); break;
        }
        __syncthreads();
        const int flag = MODE == 1 ? p.list_flag : p.fresh[e];
        uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * p.context * cpf;
        if (p.obs_f32) {
            // float32 frames: pixel * (1 / 255.0f), the product py_simulator.cpp:262-272 computes in get_state()
            const float scale = (float)(1 / 255.0);
            for (int cc = tid; cc < cpf; cc += BS) {
                const uchar4 b = reinterpret_cast<const uchar4 *>(s_frame)[cc];
                const float f0 = (float)b.x * scale, f1 = (float)b.y * scale, f2 = (float)b.z * scale, f3 = (float)b.w * scale;
                xw_store_chunk(frame0, cc, cpf, p.context, p.context > 1 ? flag : 1,
                               make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)));
            }
        } else {
            for (int cc = tid; cc < cpf; cc += BS) xw_store_chunk(frame0, cc, cpf, p.context, p.context > 1 ? flag : 1, smem4[cc]);
        }
        if (MODE == 1 && tid == 0 && p.list_flag == 2) { p.fresh[e] = 0; if (p.auto_reset == 2) p.done[e] = 0; }
    }
}

// The warped 64x64 image of every goal of the listed envs (XItem::get_item_image, xitem.cpp:46-60): cv::warpAffine with
// the goal's inverse matrix, INTER_LINEAR, BORDER_CONSTANT white.  A goal keeps its pose for the whole episode, so this
// runs once per reset (~0.4 % of the envs per step) and the render reads goal pixels like any other icon.
template <bool LIST>
__global__ __launch_bounds__(256) void xw_warp_goals_kernel(XwParams p, const uint32_t *atlas4, const int32_t *count_now) {
    const int G = p.num_goals, D = p.max_dim;
    const int n_items = (LIST ? *count_now : p.n) * G;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ei = item / G, slot = item - ei * G;
        const int e = LIST ? p.done_list[ei] : ei;
        const int cell = p.goal_cells[(size_t)e * XW_MAX_GOALS + slot];
        uint32_t *out = p.goal_img + ((size_t)e * G + slot) * 4096;
        if (cell == 0xff) continue;
        const int icon = (int)(p.grid[(size_t)e * D * D + cell] & CELL_ICON_MASK) - 1;
        if (icon < 0) continue;
        const double *M = p.goal_warp + ((size_t)e * XW_MAX_GOALS + slot) * 6;
        const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5];
        for (int q = threadIdx.x; q < 4096; q += 256) {
            const int px = q & 63, py = q >> 6;
            const int X0 = __double2int_rn((m1 * py + m2) * 1024) + 16, Y0 = __double2int_rn((m4 * py + m5) * 1024) + 16;
            const int X = (X0 + __double2int_rn(m0 * px * 1024)) >> 5, Y = (Y0 + __double2int_rn(m3 * px * 1024)) >> 5;
            const int ix = X >> 5, iy = Y >> 5, fx = X & 31, fy = Y & 31;
            uint32_t res = 0xffffffu;
            if (!(ix >= 64 || ix + 1 < 0 || iy >= 64 || iy + 1 < 0)) {
                int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
                if (w[0] == 32768) { w[0] = 32767; w[3] = 1; }     // BilinearTab_i: saturated entry and its compensation
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int qx = ix + (k & 1), qy = iy + (k >> 1);
                    t[k] = ((unsigned)qx < 64u && (unsigned)qy < 64u) ? atlas4[(uint32_t)icon * 4096u + (uint32_t)(qy * 64 + qx)] : 0xffffffu;
                }
                res = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    int acc = 1 << 14;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc += (int)((t[k] >> (8 * ch)) & 255u) * w[k];
                    res |= (uint32_t)(acc >> 15) << (8 * ch);
                }
            }
            out[q] = res;
        }
    }
}

hipError_t launch_xw_warp_goals(const XwParams &p, bool list, hipStream_t s) {
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    if (list) hipLaunchKernelGGL((xw_warp_goals_kernel<true>), dim3(1024), dim3(256), 0, s, p, a4, (const int32_t *)p.done_count);
    else hipLaunchKernelGGL((xw_warp_goals_kernel<false>), dim3(8192), dim3(256), 0, s, p, a4, (const int32_t *)p.done_count);
    return hipGetLastError();
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
    const size_t lds = (size_t)((CH * O * O + 15) & ~15) + (size_t)r * r * (sizeof(EgoCell) + 1) +
                       (size_t)((r + 3) & ~3) + XW_MAX_GOALS + 16;
    const unsigned blocks = indexed == 1 ? 2048u : (unsigned)(p.n < 16384 ? p.n : 16384);
    const int32_t *cnt = (const int32_t *)p.done_count;
#define EGO_LAUNCH(CHV, MODEV) do { if (MODEV == 1) hipLaunchKernelGGL((xw_render_ego_kernel<CHV, MODEV, 1024>), dim3(blocks), dim3(1024), lds, s, p, reinterpret_cast<const uint32_t *>(p.atlas64), h1, v1, h2, v2, cnt); \
    else hipLaunchKernelGGL((xw_render_ego_kernel<CHV, MODEV, 256>), dim3(blocks), dim3(256), lds, s, p, reinterpret_cast<const uint32_t *>(p.atlas64), h1, v1, h2, v2, cnt); } while (0)
    if (CH == 3) { if (indexed == 1) EGO_LAUNCH(3, 1); else if (indexed == 2) EGO_LAUNCH(3, 2); else EGO_LAUNCH(3, 0); }
    else { if (indexed == 1) EGO_LAUNCH(1, 1); else if (indexed == 2) EGO_LAUNCH(1, 2); else EGO_LAUNCH(1, 0); }
#undef EGO_LAUNCH
    return hipGetLastError();
}

}  // namespace xwb
