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
// leaves as aligned 16-byte stores.  Evaluating all 84^2 pixels that way is instruction-bound (16 view pixels and ~400
// VALU operations each), so only the pixels that need it are: an output pixel whose 4 x 4 view pixels all lie inside
// ONE view cell depends on nothing but that cell's image, its position in the frame and the heading, and for blocks,
// the agent, empty cells and black cells that image is one of a few constants.  xw_ego_build_tab_kernel renders, once
// per batch, the frame "every cell shows icon i" for each icon and heading with the very same pixel code; the render
// copies interior pixels from those frames (4 pixels per load, through L2) and evaluates only the pixels on a cell
// border and the pixels of goal cells (whose images are per env: pose-warped).
//
// OpenCV 3.2 arithmetic restated (third party, cmake/opencv.cmake:5-6; DESIGN.md lists the pieces): the tests compare
// this kernel bit for bit with a CPU restatement of the same pipeline; pixel parity with the real library is unpinned.
#include "xwb_common.h"
#include "xw_device.h"

#include <cmath>
#include <vector>

namespace xwb {

#ifdef XWB_EGO_PROF
__device__ unsigned long long g_ego_prof[12];
#define EGO_T0() unsigned long long t_last = wall_clock64()
#define EGO_T(i) do { if (tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&g_ego_prof[i], now - t_last); t_last = now; } } while (0)
#else
#define EGO_T0()
#define EGO_T(i)
#endif

struct EgoTap { int16_t s0, s1, w0, w1; };        // cv::resize: source indices and 11-bit weights of one output index

namespace {

// What one cell of the view shows: a 64 x 64 image (block icon, this env's warped goal image, the agent icon turned for
// its heading -- the three turned copies of every agent icon are appended to the atlas at create time) or one constant
// pixel (mask = 0).  The table makes the per-pixel lookup branch-free: one 16-byte LDS read, an AND and an add.
struct EgoCell {
    const uint32_t *img;
    int mask;                    // -1: index the image; 0: a constant pixel
    int tab;                     // frame of the interior-pixel table that shows this cell's image, -1: none (a goal)
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

// One output pixel.  DIR = the agent's heading: cv::warpAffine(view, rot(centre S/2, 90 + yaw deg)) is undone per tap
// row / column -- quarter turns are exact integer maps, separable in x and y; the source index S falls outside and
// leaves one black row / column (borderValue 0).
// ONE: all sixteen view pixels lie in the view cell `one` (an interior pixel of that cell, whose image is indexed: a goal)
template <int CH, int DIR, bool ONE>
__device__ __forceinline__ void ego_pixel(const EgoCtx &c, const EgoTap (*s_row)[3], const EgoTap (*s_col)[3],
                                          uint8_t *s_frame, int O, int ox, int oy, int one) {
    const int S = c.S, o = oy * O + ox;
    {
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
        const uint32_t *src[16];
        if (ONE) {
            const uint32_t *img = c.cells[one].img;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = k >> 2, j = k & 3;
                const int px = (ROW_IS_Y ? fc[j] : fr[i]) & 63, py = (ROW_IS_Y ? fr[i] : fc[j]) & 63;
                src[k] = img + (py * 64 + px);
            }
        } else {
            int cr[4], cc[4], pr[4], pc[4];
            bool okr[4], okc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                okr[i] = (unsigned)fr[i] < (unsigned)S; okc[i] = (unsigned)fc[i] < (unsigned)S;
                cr[i] = ROW_IS_Y ? __mul24(fr[i] >> 6, c.r) : (fr[i] >> 6);
                cc[i] = ROW_IS_Y ? (fc[i] >> 6) : __mul24(fc[i] >> 6, c.r);
                pr[i] = fr[i] & 63; pc[i] = fc[i] & 63;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = k >> 2, j = k & 3;
                const bool inview = okr[i] && okc[j];
                const EgoCell cell = c.cells[inview ? cr[i] + cc[j] : 0];
                const int px = ROW_IS_Y ? pc[j] : pr[i], py = ROW_IS_Y ? pr[i] : pc[j];
                const uint32_t *q = cell.img + ((py * 64 + px) & cell.mask);
                src[k] = inview ? q : c.black;
            }
        }
        // the descriptors come from LDS, so the compiler cannot tell these pointers are global: say so (global_load instead
        // of flat_load, which would also wait on the LDS counter)
        typedef const uint32_t __attribute__((address_space(1))) *global_u32;
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *(global_u32)src[k];
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

// The per-heading layout tables (xw_ego_tables builds them; uint16 words):
//   [0, O4)            row term: the view-cell index part every interior pixel of this output row adds (cell row * r, or
//                      the cell column for the sideways headings); bit 15: the row touches a cell border or the black
//                      border the quarter turn leaves -- all of its pixels are evaluated one by one
//   [O4, 2 O4)         column term, same
//   [2 O4, 2 O4 + Q)   column term per group of four columns (cell boundaries fall on multiples of four here, else the
//                      table is not used at all), Q = O4 / 4 rounded up to a multiple of 4
//   then 4 words       number of border rows, of border columns, largest edge of a cell's pixel rectangle, 0
//   then O4, O4        the border rows, the border columns
//   then r * r * 4     per view cell: x0, y0, width, height of its interior pixels in the frame
//   then 3 * (O4 / 4)  column segments (x4 start, dwords, column term): maximal runs of dwords of a frame row that show the
//                      same view-cell column -- the unit of the interior copy; their number is the header's 4th word
// Term flags: 0x8000 = border (the taps straddle two cells: every pixel evaluated), 0x4000 = edge (some taps fall outside
// the view -- the black line the quarter turn leaves -- but the rest lie in ONE cell: still a function of that cell's image
// alone, so the table frame of that image holds the pixel; only goal cells, whose images are per env, evaluate it).
struct EgoLayout {
    const uint16_t *rt, *ct, *ct4, *br, *bc, *rect, *seg;
    int nbr, nbc, cw, nseg;
};
constexpr uint32_t EGO_BORDER = 0x8000u, EGO_EDGE = 0x4000u, EGO_TERM = 0x3fffu;
__host__ __device__ inline int ego_layout_words(int O4, int r) {
    return 2 * O4 + ((O4 / 4 + 3) & ~3) + 4 + 2 * O4 + 4 * r * r + 3 * (O4 / 4);
}
__device__ __forceinline__ EgoLayout ego_layout(const uint16_t *base, int O4, int r) {
    EgoLayout l;
    l.rt = base; l.ct = base + O4; l.ct4 = base + 2 * O4;
    const uint16_t *h = l.ct4 + ((O4 / 4 + 3) & ~3);
    l.nbr = h[0]; l.nbc = h[1]; l.cw = h[2]; l.nseg = h[3];
    l.br = h + 4; l.bc = l.br + O4; l.rect = l.bc + O4; l.seg = l.rect + 4 * r * r;
    return l;
}

__device__ __forceinline__ int ego_div(int i, float inv_n) { return (int)(((float)i + 0.5f) * inv_n); }   // i / n, exact: i < 2^16, n <= 84 * 84

// The pixels that are evaluated one by one: FAST: every pixel of the border rows, of the border columns, and of the
// cells that show a goal (goal_k: their view-cell ids); otherwise every pixel of the frame.
template <int CH, int DIR, int BS, bool FAST>
__device__ __forceinline__ void ego_pixels(const EgoCtx &c, const EgoTap (*s_row)[3], const EgoTap (*s_col)[3],
                                           uint8_t *s_frame, int O, int tid, const EgoLayout &l, const uint8_t *goal_k, int n_goal) {
    const float inv_O = __builtin_amdgcn_rcpf((float)O);         // 1 ulp: far inside ego_div's margin
    if (!FAST) {
        for (int i = tid; i < O * O; i += BS) {
            const int oy = ego_div(i, inv_O);
            ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O, i - oy * O, oy, 0);
        }
        return;
    }
    const int cw2 = l.cw * l.cw, n_row_px = l.nbr * O, n_border_px = n_row_px + l.nbc * O;
    const float inv_cw = __builtin_amdgcn_rcpf((float)l.cw), inv_cw2 = __builtin_amdgcn_rcpf((float)cw2);
    // border rows and columns: any of the sixteen view pixels may belong to another cell, or to none
    for (int i = tid; i < n_border_px; i += BS) {
        int ox, oy;
        bool ok = true;
        if (i < n_row_px) {
            const int q = ego_div(i, inv_O);
            oy = l.br[q]; ox = i - q * O;
        } else {
            const int j = i - n_row_px, q = ego_div(j, inv_O);
            ox = l.bc[q]; oy = j - q * O;
            ok = !(l.rt[oy] & EGO_BORDER);                      // already done with its row
        }
        if (ok) ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O, ox, oy, 0);
    }
    // goal cells: interior pixels only
    for (int j = tid; j < n_goal * cw2; j += BS) {
        const int g = ego_div(j, inv_cw2), jj = j - g * cw2, k = goal_k[g];
        const uint16_t *rc = l.rect + 4 * k;
        const int py = ego_div(jj, inv_cw), px = jj - py * l.cw;
        bool ok = px < (int)rc[2] && py < (int)rc[3];
        const int ox = ok ? (int)rc[0] + px : 0, oy = ok ? (int)rc[1] + py : 0;
        const uint32_t fl = (uint32_t)l.rt[oy] | (uint32_t)l.ct[ox];
        ok = ok && !(fl & EGO_BORDER);
        if (ok) {
            if (fl & EGO_EDGE) ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O, ox, oy, 0);   // some taps are outside the view
            else ego_pixel<CH, DIR, true>(c, s_row, s_col, s_frame, O, ox, oy, k);
        }
    }
}

template <int CH, int BS, bool FAST>
__device__ __forceinline__ void ego_pixels_dir(int dir, const EgoCtx &ctx, const EgoTap (*s_row)[3], const EgoTap (*s_col)[3],
                                               uint8_t *s_frame, int O, int tid, const EgoLayout &l, const uint8_t *goal_k, int n_goal) {
    switch (dir) {
        case 0: ego_pixels<CH, 0, BS, FAST>(ctx, s_row, s_col, s_frame, O, tid, l, goal_k, n_goal); break;
        case 1: ego_pixels<CH, 1, BS, FAST>(ctx, s_row, s_col, s_frame, O, tid, l, goal_k, n_goal); break;
        case 2: ego_pixels<CH, 2, BS, FAST>(ctx, s_row, s_col, s_frame, O, tid, l, goal_k, n_goal); break;
        default: ego_pixels<CH, 3, BS, FAST>(ctx, s_row, s_col, s_frame, O, tid, l, goal_k, n_goal); break;
    }
}

// Interior pixels are copied from the table frame of the view cell they fall into.  The unit is a column segment: the
// dwords of one frame row (and plane) that show the same cell column -- 28 bytes at r = 3 -- fetched with dwordx4 / x3 /
// x2 loads (global loads only need dword alignment) instead of one gather per dword: 3.5 x fewer load instructions, which
// is what this phase is bound by (it was 44 % of the kernel).  All loads of an item are issued before its LDS writes;
// pixels of border rows / columns and of goal cells get whatever the table holds there and are overwritten by ego_pixels.
template <int CH, int BS>
__device__ __forceinline__ void ego_copy_interior(const EgoCell *s_cells, const EgoLayout &l, const uint8_t *tab, uint32_t frame_bytes,
                                                  uint8_t *s_frame, int O, int tid) {
    const int nseg = l.nseg, per_plane = O * nseg, items = CH * per_plane, rowd = O >> 2;
    const float inv_pp = __builtin_amdgcn_rcpf((float)per_plane), inv_ns = __builtin_amdgcn_rcpf((float)nseg);
    uint32_t *f32 = reinterpret_cast<uint32_t *>(s_frame);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const unsigned int __attribute__((address_space(1))) *g_u32;
    typedef const u32x2 __attribute__((address_space(1))) *g_u32x2;
    typedef const u32x4 __attribute__((address_space(1))) *g_u32x4;
    for (int it = tid; it < items; it += BS) {
        const int ch = ego_div(it, inv_pp), rem = it - ch * per_plane;
        const int oy = ego_div(rem, inv_ns), sg = rem - oy * nseg;
        const int x4 = l.seg[3 * sg], ndw = l.seg[3 * sg + 1], cterm = l.seg[3 * sg + 2];
        const int cell = (int)(l.rt[oy] & EGO_TERM) + cterm;
        const int t = s_cells[cell].tab;
        const int d0 = ch * (O * rowd) + oy * rowd + x4;                      // dword index in the planar frame
        const uint8_t *src = tab + (uint32_t)(t < 0 ? 0 : t) * frame_bytes + 4u * (uint32_t)d0;
        u32x4 q[6];
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int left = ndw - 4 * pc;
            q[pc] = (u32x4)(0u);
            if (left >= 4) q[pc] = *(g_u32x4)(src + 16 * pc);
            else if (left == 3) { const u32x2 a = *(g_u32x2)(src + 16 * pc); q[pc].x = a.x; q[pc].y = a.y; q[pc].z = *(g_u32)(src + 16 * pc + 8); }
            else if (left == 2) { const u32x2 a = *(g_u32x2)(src + 16 * pc); q[pc].x = a.x; q[pc].y = a.y; }
            else if (left == 1) q[pc].x = *(g_u32)(src + 16 * pc);
        }
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int left = ndw - 4 * pc;
            if (left >= 1) f32[d0 + 4 * pc] = q[pc].x;
            if (left >= 2) f32[d0 + 4 * pc + 1] = q[pc].y;
            if (left >= 3) f32[d0 + 4 * pc + 2] = q[pc].z;
            if (left >= 4) f32[d0 + 4 * pc + 3] = q[pc].w;
        }
    }
}

// What one view cell shows.  dir: heading; tab: -1 for goals (their images are per env)
__device__ __forceinline__ EgoCell ego_icon_cell(const uint8_t *icon_type, const uint32_t *agent_rot, const uint32_t *atlas4,
                                                 int icon, int dir) {
    EgoCell c{atlas4 + (size_t)icon * 4096, -1, icon * 4 + dir};
    // the agent: XItem::get_item_image turns its icon by 90 - yaw deg
    if (icon_type[icon] == 2 && dir != 1) c.img = atlas4 + agent_rot[icon] + (size_t)(dir == 0 ? 0 : (dir == 2 ? 1 : 2)) * 4096;
    return c;
}

__device__ __forceinline__ void ego_compose_taps(EgoTap (*s_row)[3], EgoTap (*s_col)[3], const EgoTap *tap_h1, const EgoTap *tap_v1,
                                                 const EgoTap *tap_h2, const EgoTap *tap_v2, int O, int tid, int bs) {
    for (int i = tid; i < O; i += bs) {
        const EgoTap ty = tap_v2[i], tx = tap_h2[i];
        s_row[i][0] = tap_v1[ty.s0]; s_row[i][1] = tap_v1[ty.s1]; s_row[i][2] = ty;
        s_col[i][0] = tap_h1[tx.s0]; s_col[i][1] = tap_h1[tx.s1]; s_col[i][2] = tx;
    }
}

}  // namespace

// The interior-pixel table: frame (slot, heading) = the frame of a view whose every cell shows slot's image; slots
// 0 .. n_icons - 1 = the icons, n_icons = an empty (white) cell, n_icons + 1 = a black cell.  One workgroup per frame.
template <int CH>
__global__ __launch_bounds__(256) void xw_ego_build_tab_kernel(XwParams p, const uint32_t *atlas4, const EgoTap *tap_h1,
                                                               const EgoTap *tap_v1, const EgoTap *tap_h2, const EgoTap *tap_v2,
                                                               uint8_t *tab, size_t frame_bytes) {
    extern __shared__ uint4 smem4[];
    const int r = p.visible_radius, O = p.out_dim;
    uint8_t *s_frame = reinterpret_cast<uint8_t *>(smem4);
    EgoCell *s_cells = reinterpret_cast<EgoCell *>(s_frame + ((CH * O * O + 15) & ~15));
    __shared__ EgoTap s_row[84][3], s_col[84][3];
    const int tid = threadIdx.x, slot = blockIdx.x >> 2, dir = blockIdx.x & 3;
    ego_compose_taps(s_row, s_col, tap_h1, tap_v1, tap_h2, tap_v2, O, tid, 256);
    const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
    EgoCell c{slot == p.n_icons ? white : black, 0, -1};
    if (slot < p.n_icons) c = ego_icon_cell(p.icon_type, p.ego_agent_rot, atlas4, slot, dir);
    for (int k = tid; k < r * r; k += 256) s_cells[k] = c;
    __syncthreads();
    EgoCtx ctx{s_cells, white, black, r, 64 * r};
    ego_pixels_dir<CH, 256, false>(dir, ctx, s_row, s_col, s_frame, O, tid, EgoLayout{}, nullptr, 0);
    __syncthreads();
    uint8_t *out = tab + (size_t)blockIdx.x * frame_bytes;
    for (int i = tid; i < CH * O * O; i += 256) out[i] = s_frame[i];
}

// MODE 0: every env; 1: the compacted done list; 2: every env the last step did not finish (the rest is drawn from the list)
// BS threads per workgroup: 256 for the whole batch; 1024 for the short done list, where the latency of one env counts
// FAST: frame rows are whole dwords and cell boundaries fall on dwords (r <= 7): interior pixels are copied from the
// table and frames leave as 16-byte chunks.  Otherwise (r >= 9: 81, 77, 78, 75 pixel edges) every pixel is evaluated
// and frames leave element by element -- their byte size is not a multiple of 16.
template <int CH, int MODE, int BS, bool FAST>
__global__ __launch_bounds__(BS, 4) void xw_render_ego_kernel(XwParams p, const uint32_t *atlas4, const EgoTap *tap_h1,
                                                            const EgoTap *tap_v1, const EgoTap *tap_h2, const EgoTap *tap_v2,
                                                            const uint16_t *layout, const uint8_t *tab,
                                                            const int32_t *count_now) {
    extern __shared__ uint4 smem4[];
    const int r = p.visible_radius, S = 64 * r, D = p.max_dim, O = p.out_dim, O4 = (O + 3) & ~3;
    const uint32_t frame_bytes = (uint32_t)((CH * O * O + 15) & ~15);
    const int lw = ego_layout_words(O4, r);
    uint8_t *s_frame = reinterpret_cast<uint8_t *>(smem4);                       // CH * O * O, planar
    EgoCell *s_cells = reinterpret_cast<EgoCell *>(s_frame + frame_bytes);
    uint16_t *s_layout = reinterpret_cast<uint16_t *>(s_cells + r * r);          // FAST: the four headings' layout tables
    uint32_t *s_rot = reinterpret_cast<uint32_t *>(s_layout + (FAST ? 4 * lw : 0));   // [n_icons] ego_agent_rot
    uint8_t *s_itype = reinterpret_cast<uint8_t *>(s_rot + p.n_icons);           // [n_icons] icon_type
    uint8_t *s_type = s_itype + ((p.n_icons + 3) & ~3);                          // [D * D] type of the entity in a cell, 3 = none
    uint8_t *s_shadow = s_type + ((D * D + 3) & ~3);
    uint8_t *s_ray = s_shadow + ((r * r + 3) & ~3);
    uint8_t *s_gc = s_ray + ((r + 3) & ~3);
    uint8_t *s_goal_k = s_gc + XW_MAX_GOALS;                                     // [XW_MAX_GOALS] view cells that show a goal
    uint8_t *s_goal_slot = s_goal_k + XW_MAX_GOALS;                              // [XW_MAX_GOALS] their goal slots
    uint8_t *s_miss_k = s_goal_slot + XW_MAX_GOALS;                              // [XW_MAX_GOALS] those not in the cache yet
    uint8_t *s_miss_slot = s_miss_k + XW_MAX_GOALS;
    // composed taps of one output row / column: the two intermediate indices' taps and the output tap (static: O <= 84)
    __shared__ EgoTap s_row[84][3], s_col[84][3];
    __shared__ uint16_t s_code[XW_MAX_DIM * XW_MAX_DIM];                         // the env's grid, target bit stripped
    __shared__ int s_ngoal, s_nmiss;
    __shared__ uint32_t s_valid[64];                                             // the env's cache bits (ego_cache_words <= 64)
    const int tid = threadIdx.x;
    const int n_items = MODE == 1 ? *count_now : p.n;
    if ((int)blockIdx.x >= n_items) return;                    // the done list is short: most of its workgroups leave here
    ego_compose_taps(s_row, s_col, tap_h1, tap_v1, tap_h2, tap_v2, O, tid, BS);
    for (int i = tid; i < p.n_icons; i += BS) { s_itype[i] = p.icon_type[i]; s_rot[i] = p.ego_agent_rot[i]; }
    if (FAST) for (int i = tid; i < 4 * lw; i += BS) s_layout[i] = layout[i];
    const int cells = D * D;
    // Everything the env's setup reads from global memory is fetched one env ahead and staged in LDS, so the serial part
    // -- shadow rays, scan lines, cell table -- never waits for HBM / L2.  The setup is the first wavefront's job alone
    // (its lanes hold the grid: cells <= 256 = 4 per lane): it is scalar-heavy code that every wavefront would otherwise
    // repeat, and inside one wavefront its phases need no workgroup barrier (LDS operations of a wave complete in order).
    constexpr int CPL = XW_MAX_DIM * XW_MAX_DIM / 64;           // grid cells per lane of the first wavefront
    const bool wave0 = tid < 64;
    struct Fetch { int e, axy, dir, skip; uint32_t code[CPL], gc, valid; };
    auto fetch = [&](int item) {
        Fetch f;
        f.e = MODE == 1 ? p.done_list[item] : item;
        f.skip = MODE == 2 ? (int)p.term_flag[f.e] : 0;      // finished by this step: drawn from the list instead
        f.axy = p.agent_xy[f.e]; f.dir = p.agent_dir[f.e];
#pragma unroll
        for (int k = 0; k < CPL; ++k) f.code[k] = wave0 && tid + 64 * k < cells ? (uint32_t)p.grid[(size_t)f.e * cells + tid + 64 * k] : 0u;
        f.gc = tid < XW_MAX_GOALS ? (uint32_t)p.goal_cells[(size_t)f.e * XW_MAX_GOALS + tid] : 0xffu;
        // the env's goal-cell cache bits, lane i = word i (fetched with the rest, one env ahead: the look-up never waits)
        f.valid = (FAST && p.ego_cache_valid && tid < (int)p.ego_cache_words && tid < 64) ? p.ego_cache_valid[(size_t)f.e * p.ego_cache_words + tid] : 0u;
        return f;
    };
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    Fetch nxt{};
    if ((int)blockIdx.x < n_items) nxt = fetch(blockIdx.x);
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const Fetch f = nxt;
        const int e = f.e, ax = f.axy & 0xffff, ay = f.axy >> 16, dir = f.dir;
        __syncthreads();                                        // the previous env's frame has left LDS
        EGO_T0();
        if (wave0) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = tid + 64 * k;
                if (c < cells) {
                    const int code = (int)(f.code[k] & CELL_ICON_MASK);
                    s_code[c] = (uint16_t)code;
                    s_type[c] = code ? s_itype[code - 1] : (uint8_t)3;
                }
            }
            if (tid < XW_MAX_GOALS) s_gc[tid] = (uint8_t)f.gc;
            if (FAST && tid < (int)p.ego_cache_words) s_valid[tid] = f.valid;
            if (tid < r) s_ray[tid] = 1;
            if (tid == 0) s_ngoal = 0;
        }
        if (item + (int)gridDim.x < n_items) nxt = fetch(item + gridDim.x);
        if (f.skip) continue;
        EgoLayout lay{};
        if (FAST) lay = ego_layout(s_layout + dir * lw, O4, r);
        const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
        if (wave0) {
            auto is_block = [&](int x, int y) {
                return (unsigned)x < (unsigned)D && (unsigned)y < (unsigned)D && s_type[y * D + x] == 1;
            };
            // XMap::image_masking (xmap.cpp:273-362)
            int major_x = 0, major_y = 0, minor_x = 0, minor_y = 0, scan_x0 = 0, scan_y0 = 0, xa = ax + r, ya = ay + r;
            if (dir == 0) { xa += r / 2; major_y = 1; minor_x = 1; }
            else if (dir == 3) { ya -= r / 2; major_x = 1; minor_y = -1; scan_y0 = r - 1; }
            else if (dir == 2) { xa -= r / 2; major_y = 1; minor_x = -1; scan_x0 = r - 1; }
            else { ya += r / 2; major_x = 1; minor_y = 1; }
            const int x_st = xa - r / 2, y_st = ya - r / 2;
            wave_sync();
            if (tid < 2) {                                      // rays to either side of the agent
                const int o = tid ? 1 : -1;
                bool block = false;
                int rx = ax, ry = ay;
                for (int k = 1; k <= r / 2; ++k) {
                    rx += o * major_x; ry += o * major_y;
                    if (block) s_ray[r / 2 + o * k] = 0;
                    if (is_block(rx, ry)) block = true;
                }
            }
            wave_sync();
            if (tid < r) {                                      // one scan line per lane
                bool block = !s_ray[tid];
                int cx = scan_x0 + tid * major_x, cy = scan_y0 + tid * major_y;
                for (int j = 0; j < r; ++j) {
                    s_shadow[cy * r + cx] = block ? 1 : 0;
                    if (is_block(x_st - r + cx, y_st - r + cy)) block = true;
                    cx = (cx + minor_x + r) % r;
                    cy = (cy + minor_y + r) % r;
                }
            }
            wave_sync();
            const uint32_t *gimg = p.goal_img + (size_t)e * p.num_goals * 4096;
            for (int k = tid; k < r * r; k += 64) {             // what each view cell shows
                const int gx = x_st - r + k % r, gy = y_st - r + k / r;
                EgoCell c{black, 0, (p.n_icons + 1) * 4 + dir}; // outside the map, or in a wall's shadow
                if ((unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && !(s_shadow[k] && !p.no_wall_shadow)) {
                    const int code = s_code[gy * D + gx];
                    if (code == 0) { c.img = white; c.tab = p.n_icons * 4 + dir; }
                    else {
                        c = ego_icon_cell(s_itype, s_rot, atlas4, code - 1, dir);
                        if (s_type[gy * D + gx] == 0) {         // a goal: this env's warped copy
                            int slot = 0;
                            for (int i = 0; i < XW_MAX_GOALS; ++i) if (s_gc[i] == gy * D + gx) slot = i;
                            c.img = gimg + slot * 4096;
                            c.tab = -1;
                            if (FAST) { const int j = atomicAdd(&s_ngoal, 1); s_goal_k[j] = (uint8_t)k; s_goal_slot[j] = (uint8_t)slot; }
                        }
                    }
                }
                s_cells[k] = c;
            }
        }
        __syncthreads();
        EGO_T(3);
        EgoCtx ctx{s_cells, white, black, r, S};
        if (FAST) {
            ego_copy_interior<CH, BS>(s_cells, lay, tab, frame_bytes, s_frame, O, tid);
            __syncthreads();
            EGO_T(4);
        }
        const uint8_t *eval_k = s_goal_k;
        int n_eval = s_ngoal;
        // (four frames in five show no goal at all: nothing to look up, nothing to evaluate, no barrier)
        const bool cached = FAST && p.ego_cache != nullptr && s_ngoal > 0;
        uint8_t *cache_env = nullptr;
        uint32_t *valid_env = nullptr;
        if (cached) {
            // goal cells: copy the ones this env has already rendered in this place and heading, evaluate the rest (and keep them)
            cache_env = p.ego_cache + (size_t)e * p.num_goals * (r * r * 4) * p.ego_cache_entry;
            valid_env = p.ego_cache_valid + (size_t)e * p.ego_cache_words;
            if (tid == 0) {
                int nm = 0;
                for (int j = 0; j < s_ngoal; ++j) {
                    const int bit = (s_goal_slot[j] * r * r + s_goal_k[j]) * 4 + dir;
                    if (!((s_valid[bit >> 5] >> (bit & 31)) & 1u)) { s_miss_k[nm] = s_goal_k[j]; s_miss_slot[nm] = s_goal_slot[j]; nm++; s_goal_k[j] = 0xff; }
                }
                s_nmiss = nm;
            }
            __syncthreads();
            for (int j = 0; j < s_ngoal; ++j) {
                const int k = s_goal_k[j];
                if (k == 0xff) continue;                                 // a miss
                const uint16_t *rc = lay.rect + 4 * k;
                const int x0 = rc[0], y0 = rc[1], w = rc[2], h = rc[3], wh = w * h;
                const uint8_t *src = cache_env + (size_t)((s_goal_slot[j] * r * r + k) * 4 + dir) * p.ego_cache_entry;
                for (int i = tid; i < wh * CH; i += BS) {
                    const int ch = i / wh, rem = i - ch * wh, py = rem / w, px = rem - py * w;
                    s_frame[ch * O * O + (y0 + py) * O + x0 + px] = src[i];
                }
            }
            eval_k = s_miss_k;
            n_eval = s_nmiss;
#ifdef XWB_EGO_PROF
            if (tid == 0) { atomicAdd(&g_ego_prof[8], (unsigned long long)s_ngoal); atomicAdd(&g_ego_prof[9], (unsigned long long)s_nmiss); atomicAdd(&g_ego_prof[10], 1ull); }
#endif
        }
        ego_pixels_dir<CH, BS, FAST>(dir, ctx, s_row, s_col, s_frame, O, tid, lay, eval_k, n_eval);
        __syncthreads();
        if (cached && n_eval > 0) {
            for (int j = 0; j < n_eval; ++j) {
                const int k = s_miss_k[j];
                const uint16_t *rc = lay.rect + 4 * k;
                const int x0 = rc[0], y0 = rc[1], w = rc[2], h = rc[3], wh = w * h;
                const int entry = (s_miss_slot[j] * r * r + k) * 4 + dir;
                uint8_t *dst = cache_env + (size_t)entry * p.ego_cache_entry;
                for (int i = tid; i < wh * CH; i += BS) {
                    const int ch = i / wh, rem = i - ch * wh, py = rem / w, px = rem - py * w;
                    dst[i] = s_frame[ch * O * O + (y0 + py) * O + x0 + px];
                }
                if (tid == 0) atomicOr(valid_env + (entry >> 5), 1u << (entry & 31));
            }
        }
        EGO_T(5);
        const int flag = p.context > 1 ? (MODE == 1 ? p.list_flag : (int)p.fresh[e]) : 1;
        const float scale = (float)(1 / 255.0);   // float32 frames: pixel * (1 / 255.0f), the product py_simulator.cpp:262-272 computes
        if (FAST) {
            const int cpf = CH * O * O / (p.obs_f32 ? 4 : 16);  // 16-byte chunks per frame: 16 uint8 pixels, or 4 float32 ones
            uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * p.context * cpf;
            if (p.obs_f32) {
                for (int cc = tid; cc < cpf; cc += BS) {
                    const uchar4 b = reinterpret_cast<const uchar4 *>(s_frame)[cc];
                    const float f0 = (float)b.x * scale, f1 = (float)b.y * scale, f2 = (float)b.z * scale, f3 = (float)b.w * scale;
                    xw_store_chunk(frame0, cc, cpf, p.context, flag,
                                   make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3)));
                }
            } else {
                for (int cc = tid; cc < cpf; cc += BS) xw_store_chunk(frame0, cc, cpf, p.context, flag, smem4[cc]);
            }
        } else if (flag != 0) {
            // shift_context / init_screen element by element (simulator.cpp:36-85)
            const int F = CH * O * O, ctxn = p.context;
            if (p.obs_f32) {
                float *q = reinterpret_cast<float *>(p.obs) + (size_t)e * ctxn * F;
                for (int i = tid; i < F; i += BS) {
                    for (int f = 0; f + 1 < ctxn; ++f) q[(size_t)f * F + i] = flag == 2 ? 0.f : q[(size_t)(f + 1) * F + i];
                    q[(size_t)(ctxn - 1) * F + i] = (float)s_frame[i] * scale;
                }
            } else {
                uint8_t *q = p.obs + (size_t)e * ctxn * F;
                for (int i = tid; i < F; i += BS) {
                    for (int f = 0; f + 1 < ctxn; ++f) q[(size_t)f * F + i] = flag == 2 ? (uint8_t)0 : q[(size_t)(f + 1) * F + i];
                    q[(size_t)(ctxn - 1) * F + i] = s_frame[i];
                }
            }
        }
        EGO_T(6);
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
        // new poses: whatever the render cached of this env's goal cells is stale
        if (slot == 0 && p.ego_cache_valid)
            for (int q = threadIdx.x; q < (int)p.ego_cache_words; q += 256) p.ego_cache_valid[(size_t)e * p.ego_cache_words + q] = 0;
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

// The taps of both resizes (h1, v1: view -> canvas size; h2, v2: canvas size -> frame), then the four headings' layout
// tables (EgoLayout).  An output row is interior when the four view rows behind it exist and lie in one cell row (or
// column, for the sideways headings).  *fast_out: frame rows are whole dwords and no dword holds interior pixels of two
// cells -- the condition for copying interior pixels from the table.
hipError_t xw_ego_tables(int r, int max_dim, int out_dim, EgoTap **dev_out, int *fast_out, int *cell_edge_out) {
    int cell_edge = 1;
    std::vector<EgoTap> h1, v1, h2, v2;
    resize_taps(64 * r, 64 * max_dim, h1, v1);
    resize_taps(64 * max_dim, out_dim, h2, v2);
    std::vector<EgoTap> all;
    all.insert(all.end(), h1.begin(), h1.end()); all.insert(all.end(), v1.begin(), v1.end());
    all.insert(all.end(), h2.begin(), h2.end()); all.insert(all.end(), v2.begin(), v2.end());
    const int O = out_dim, O4 = (O + 3) & ~3, S = 64 * r, lw = ego_layout_words(O4, r), q4 = (O4 / 4 + 3) & ~3;
    std::vector<uint16_t> lay((size_t)4 * lw, 0);
    bool fast = (O & 3) == 0 && r * r <= 64;
    for (int dir = 0; dir < 4; ++dir) {
        uint16_t *L = lay.data() + (size_t)dir * lw;
        uint16_t *rt = L, *ct = L + O4, *ct4 = L + 2 * O4, *hd = ct4 + q4, *br = hd + 4, *bc = br + O4, *rect = bc + O4;
        uint16_t *seg = rect + 4 * r * r;
        const bool row_is_y = dir == 3 || dir == 1;
        std::vector<int> cell_of[2];                           // per axis: the cell coordinate of an interior row / column, -1 border
        for (int axis = 0; axis < 2; ++axis) {                 // 0: output rows, 1: output columns
            const std::vector<EgoTap> &t1 = axis ? h1 : v1, &t2 = axis ? h2 : v2;
            const bool flip = axis ? !(dir == 3 || dir == 0) : !(dir == 3 || dir == 2);
            const bool times_r = axis ? !row_is_y : row_is_y;
            cell_of[axis].assign(O, -1);
            uint16_t *term = axis ? ct : rt, *border = axis ? bc : br;
            int nb = 0;
            for (int o = 0; o < O; ++o) {
                const int idx[4] = {t1[t2[o].s0].s0, t1[t2[o].s0].s1, t1[t2[o].s1].s0, t1[t2[o].s1].s1};
                int cell = -1;
                bool ok = true, edge = false;
                for (int i = 0; i < 4; ++i) {
                    const int f = flip ? S - idx[i] : idx[i];
                    if (f < 0 || f >= S) { edge = true; continue; }       // outside the view: black whatever the cells show
                    if (cell < 0) cell = f >> 6;
                    else if (cell != (f >> 6)) ok = false;
                }
                if (cell < 0) cell = 0;                                   // (all four outside: cannot happen, taps are adjacent pairs)
                if (ok) { term[o] = (uint16_t)((times_r ? cell * r : cell) | (edge ? EGO_EDGE : 0u)); cell_of[axis][o] = cell; }
                else { term[o] = (uint16_t)EGO_BORDER; border[nb++] = (uint16_t)o; }
            }
            hd[axis] = (uint16_t)nb;
        }
        for (int x4 = 0; x4 < O4 / 4; ++x4) {                  // the column term of a dword
            int term = -1;
            for (int j = 0; j < 4 && 4 * x4 + j < O; ++j) {
                if (ct[4 * x4 + j] & EGO_BORDER) continue;
                const int tj = ct[4 * x4 + j] & EGO_TERM;
                if (term < 0) term = tj;
                else if (term != tj) fast = false;
            }
            ct4[x4] = (uint16_t)(term < 0 ? 0 : term);
        }
        {   // column segments: maximal runs of dwords with the same column term (at most 24 dwords: six x4 loads per item)
            int ns = 0;
            for (int x4 = 0; x4 < O4 / 4; ++x4) {
                if (ns > 0 && seg[3 * (ns - 1) + 2] == ct4[x4] && seg[3 * (ns - 1) + 1] < 24) seg[3 * (ns - 1) + 1]++;
                else { seg[3 * ns] = (uint16_t)x4; seg[3 * ns + 1] = 1; seg[3 * ns + 2] = ct4[x4]; ns++; }
            }
            hd[3] = (uint16_t)ns;
        }
        int cw = 1;
        for (int k = 0; k < r * r; ++k) {                       // view cell k = vy * r + vx: where its interior pixels are
            const int vx = k % r, vy = k / r;
            const int row_cell = row_is_y ? vy : vx, col_cell = row_is_y ? vx : vy;
            int y0 = O, y1 = -1, x0 = O, x1 = -1;
            for (int o = 0; o < O; ++o) {
                if (cell_of[0][o] == row_cell) { if (o < y0) y0 = o; if (o > y1) y1 = o; }
                if (cell_of[1][o] == col_cell) { if (o < x0) x0 = o; if (o > x1) x1 = o; }
            }
            const int w = x1 >= x0 ? x1 - x0 + 1 : 0, h = y1 >= y0 ? y1 - y0 + 1 : 0;
            rect[4 * k] = (uint16_t)(w ? x0 : 0); rect[4 * k + 1] = (uint16_t)(h ? y0 : 0);
            rect[4 * k + 2] = (uint16_t)w; rect[4 * k + 3] = (uint16_t)h;
            if (w > cw) cw = w;
            if (h > cw) cw = h;
        }
        hd[2] = (uint16_t)cw;
        if (cw > cell_edge) cell_edge = cw;
    }
    const size_t tap_bytes = all.size() * sizeof(EgoTap), lay_bytes = lay.size() * sizeof(uint16_t);
    uint8_t *d = nullptr;
    hipError_t err = hipMalloc(&d, tap_bytes + lay_bytes);
    if (err != hipSuccess) return err;
    err = hipMemcpy(d, all.data(), tap_bytes, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + tap_bytes, lay.data(), lay_bytes, hipMemcpyHostToDevice);
    *dev_out = reinterpret_cast<EgoTap *>(d);
    *fast_out = fast ? 1 : 0;
    if (cell_edge_out) *cell_edge_out = cell_edge;
    return err;
}

namespace {
struct EgoTables { const EgoTap *h1, *v1, *h2, *v2; const uint16_t *lut; };
EgoTables ego_tables_of(const XwParams &p) {
    const int P = 64 * p.max_dim, O = p.out_dim;
    EgoTables t;
    t.h1 = reinterpret_cast<const EgoTap *>(p.ego_taps); t.v1 = t.h1 + P; t.h2 = t.v1 + P; t.v2 = t.h2 + O;
    t.lut = reinterpret_cast<const uint16_t *>(t.v2 + O);
    return t;
}
size_t ego_frame_bytes(const XwParams &p) { return (size_t)((p.channels * p.out_dim * p.out_dim + 15) & ~15); }
}  // namespace

// bytes of one cache entry: the largest cell rectangle of any heading, all channels
size_t xw_ego_cache_entry_bytes(const XwParams &p, int cell_edge) { return (size_t)((cell_edge * cell_edge * p.channels + 15) & ~15); }

size_t xw_ego_tab_bytes(const XwParams &p) { return (size_t)(p.n_icons + 2) * 4 * ego_frame_bytes(p); }

// fills p.ego_tab (xw_ego_tab_bytes) -- once per batch, after the atlas and the taps are on the device
hipError_t launch_xw_ego_build_tab(const XwParams &p, hipStream_t s) {
    const EgoTables t = ego_tables_of(p);
    const int r = p.visible_radius;
    const size_t fb = ego_frame_bytes(p), lds = fb + (size_t)r * r * sizeof(EgoCell);
    const unsigned blocks = (unsigned)(p.n_icons + 2) * 4;
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    uint8_t *tab = const_cast<uint8_t *>(p.ego_tab);
    if (p.channels == 3) hipLaunchKernelGGL((xw_ego_build_tab_kernel<3>), dim3(blocks), dim3(256), lds, s, p, a4, t.h1, t.v1, t.h2, t.v2, tab, fb);
    else hipLaunchKernelGGL((xw_ego_build_tab_kernel<1>), dim3(blocks), dim3(256), lds, s, p, a4, t.h1, t.v1, t.h2, t.v2, tab, fb);
    return hipGetLastError();
}

hipError_t launch_xw_render_ego(const XwParams &p, int indexed, hipStream_t s) {
    const EgoTables t = ego_tables_of(p);
    const int r = p.visible_radius, O = p.out_dim, O4 = (O + 3) & ~3, D = p.max_dim;
    const int CH = p.channels;
    const bool fast = p.ego_fast != 0;
    const size_t lds = ego_frame_bytes(p) + (size_t)r * r * sizeof(EgoCell) + (fast ? (size_t)ego_layout_words(O4, r) * 8 : 0) +
                       (size_t)p.n_icons * 4 + (size_t)((p.n_icons + 3) & ~3) + (size_t)((D * D + 3) & ~3) +
                       (size_t)((r * r + 3) & ~3) + (size_t)((r + 3) & ~3) + 5 * XW_MAX_GOALS + 16;
    // whole batch: looping workgroups, each with its next env's state in flight, so the per-workgroup prologue (taps and
    // layout tables -> LDS) is amortised; 8192 of them rather than the 1024 that are resident at once: a shorter tail, and
    // a reset_done running on the side stream finds free slots (MI355X, C4 batch: 0.518 ms per step with 1024, 0.494 with 8192)
    const unsigned blocks = indexed == 1 ? 2048u : (unsigned)(p.n < 8192 ? p.n : 8192);
    const int32_t *cnt = (const int32_t *)p.done_count;
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
#define EGO_LAUNCH2(CHV, MODEV, BSV, FASTV) hipLaunchKernelGGL((xw_render_ego_kernel<CHV, MODEV, BSV, FASTV>), dim3(blocks), dim3(BSV), lds, s, p, a4, t.h1, t.v1, t.h2, t.v2, t.lut, p.ego_tab, cnt)
#define EGO_LAUNCH1(CHV, MODEV, BSV) do { if (fast) EGO_LAUNCH2(CHV, MODEV, BSV, true); else EGO_LAUNCH2(CHV, MODEV, BSV, false); } while (0)
#define EGO_LAUNCH(CHV) do { if (indexed == 1) { if (p.ego_list_beside) EGO_LAUNCH1(CHV, 1, 256); else EGO_LAUNCH1(CHV, 1, 1024); } else if (indexed == 2) EGO_LAUNCH1(CHV, 2, 256); else EGO_LAUNCH1(CHV, 0, 256); } while (0)
    if (CH == 3) EGO_LAUNCH(3); else EGO_LAUNCH(1);
#undef EGO_LAUNCH
#undef EGO_LAUNCH1
#undef EGO_LAUNCH2
    return hipGetLastError();
}

}  // namespace xwb

#ifdef XWB_EGO_PROF
extern "C" int xwb_debug_ego_prof(unsigned long long *out) {
    unsigned long long z[12] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(xwb::g_ego_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(xwb::g_ego_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
