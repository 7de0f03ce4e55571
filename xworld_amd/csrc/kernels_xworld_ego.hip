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
// cv::remap's 5-bit sub-pixel bilinear weights.  Evaluating all 84^2 pixels that way is instruction-bound (16 view pixels
// and ~400 VALU operations each), so only the pixels that need it are: an output pixel whose 4 x 4 view pixels all lie
// inside ONE view cell depends on nothing but that cell's image, its position in the frame and the heading, and for
// blocks, the agent, empty cells and black cells that image is one of a few constants.
//
// Two renders share that pixel code (both bit-exact against the oracle and against each other):
//   - the SPAN PATH (second half of this file; r = 3, 5, 7): cell table -> evaluated pixels -> a gather of 16-byte pieces
//     from tables of whole squares; what draws the whole batch and the done list whenever the geometry allows;
//   - ONE WORKGROUP PER ENV (xw_render_ego_kernel, first half; round 1's kernel and the fallback): the frame is assembled in
//     LDS from table frames "every cell shows icon i" (xw_ego_build_tab_kernel) plus evaluated border pixels and goal cells.
//
// OpenCV 3.2 arithmetic restated (third party, cmake/opencv.cmake:5-6; DESIGN.md lists the pieces): the tests compare
// this kernel bit for bit with a CPU restatement of the same pipeline; pixel parity with the real library is unpinned.
#include "xwb_common.h"
#include "xw_device.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace xwb {

#ifdef XWB_EGO_PROF
__device__ unsigned long long g_ego_prof[12];
__device__ unsigned long long g_ego_prof2[12];     // stage stamps of the whole-batch cells kernel
#define EGO_C(i) do { if (!LIST && tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&g_ego_prof2[i], now - t_c); t_c = now; } } while (0)
#define EGO_C0() unsigned long long t_c = wall_clock64()
#define EGO_T0() unsigned long long t_last = wall_clock64()
#define EGO_T(i) do { if (tid == 0) { const unsigned long long now = wall_clock64(); atomicAdd(&g_ego_prof[i], now - t_last); t_last = now; } } while (0)
#else
#define EGO_T0()
#define EGO_T(i)
#define EGO_C0()
#ifdef EGO_CELLS_STOP                                    // lab: the whole-batch cells kernel ends at stage i (wrong frames; for a kernel trace:
#define EGO_C(i) do { if (!LIST && (i) == EGO_CELLS_STOP) return; } while (0)      // tools/lab/ego_cells_ablate.sh)
#else
#define EGO_C(i)
#endif
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
    int dir;                     // the heading, where it is not a template argument (ego_pixel<.., -1, ..>)
};

// cv::resize INTER_LINEAR on 8-bit data, one output value: HResizeLinear (11-bit) then VResizeLinear<uchar>
__device__ __forceinline__ int vresize(int b0, int h0, int b1, int h1) {
    // operands < 2^24 and products < 2^31: v_mul_u32_u24 is exact and full rate
    return (int)((((__umul24((unsigned)b0, (unsigned)(h0 >> 4))) >> 16) + ((__umul24((unsigned)b1, (unsigned)(h1 >> 4))) >> 16) + 2u) >> 2);
}

// One output pixel.  DIR = the agent's heading: cv::warpAffine(view, rot(centre S/2, 90 + yaw deg)) is undone per tap
// row / column -- quarter turns are exact integer maps, separable in x and y; the source index S falls outside and
// leaves one black row / column (borderValue 0).
// (DIR = -1: the heading is c.dir, a run-time value -- the same arithmetic with selects, for lanes of mixed headings)
// ONE: all sixteen view pixels lie in the view cell `one` (an interior pixel of that cell, whose image is indexed: a goal)
template <int CH, int DIR, bool ONE>
__device__ __forceinline__ void ego_pixel(const EgoCtx &c, const EgoTap (*s_row)[3], const EgoTap (*s_col)[3],
                                          uint8_t *s_frame, int plane, int o, int ox, int oy, int one) {
    const int S = c.S;
    {
        // the 2 x 2 intermediate pixels this output pixel blends, and the 4 x 4 view pixels behind them
        const EgoTap ty = s_row[oy][2], tx = s_col[ox][2];
        const EgoTap my[2] = {s_row[oy][0], s_row[oy][1]}, mx[2] = {s_col[ox][0], s_col[ox][1]};
        const int R[4] = {my[0].s0, my[0].s1, my[1].s0, my[1].s1}, C[4] = {mx[0].s0, mx[0].s1, mx[1].s0, mx[1].s1};
        // source coordinate contributed by a view row (vr) and by a view column (vc):
        //   up (3): sx = vc, sy = vr;  right (0): sx = S - vr, sy = vc;  down (1): sx = S - vc, sy = S - vr;  left (2): sx = vr, sy = S - vc
        int fr[4], fc[4];                                   // coordinate from the row index, from the column index
        const int dir = DIR >= 0 ? DIR : c.dir;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fr[i] = (dir == 3 || dir == 2) ? R[i] : S - R[i];
            fc[i] = (dir == 3 || dir == 0) ? C[i] : S - C[i];
        }
        // fr is sy for headings up / down and sx for right / left (and fc the other one)
        const bool ROW_IS_Y = dir == 3 || dir == 1;
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
            s_frame[o] = (uint8_t)out[0]; s_frame[plane + o] = (uint8_t)out[1]; s_frame[2 * plane + o] = (uint8_t)out[2];
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
            ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O * O, i, i - oy * O, oy, 0);
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
        if (ok) ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O * O, oy * O + ox, ox, oy, 0);
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
            if (fl & EGO_EDGE) ego_pixel<CH, DIR, false>(c, s_row, s_col, s_frame, O * O, oy * O + ox, ox, oy, 0);   // some taps are outside the view
            else ego_pixel<CH, DIR, true>(c, s_row, s_col, s_frame, O * O, oy * O + ox, ox, oy, k);
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
        const bool cached = FAST && p.ego_cache != nullptr && p.ego_cellinfo == nullptr && s_ngoal > 0;   // (the span path keeps another entry layout)
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


// ------------------------------------------------------------------------------------------------ span path ----
// The whole-batch render when the frame is a grid of r x r equal squares, one per view cell (U = O / r pixels), and the only
// rows / columns whose taps straddle two cells are first rows / columns of a square (xw_ego_tables checks; true of r = 3,
// 5, 7 on every map size tried): then a frame is U-byte runs, each copied from the table frame of what its view cell
// shows or from the env's rendered goal cell; the first row / column of a square that blends a goal's image comes from that
// goal's cache entry, the pixel where a border row crosses a border column from a table of four classes (round 5; rounds 2-4
// evaluated those lines for every env on every step).  The one-workgroup-per-env kernel above spends its time waiting (three
// barriers and a serial set-up per frame, four workgroups per CU: 14 us per frame and workgroup, 0.20 of the HBM roofline);
// split by what is parallel in:
//   xw_ego_cells_kernel   lane per env: shadow rays and scan lines on bit masks -> cellinfo[env][view cell], and the list
//                         of goal cells the cache does not hold yet
//   xw_ego_eval_kernel    the pixels that have to be evaluated: four workgroups per listed goal cell -- the U x U pixels of its
//                         square and the border lines next to it that blend the goal's image -> cache entry (EgoEntry), valid
//                         bit.  (Rounds 2-4 evaluated those lines and the crossing pixels of EVERY env on every step into a
//                         per-env buffer: a second kind of workgroup whose chain of dependent reads made this kernel 42 us.)
//   xw_ego_gather_kernel  one-shot workgroups over 16-byte chunk spans of the batch's frame bytes, cut by the global chunk
//                         index exactly like the full-observation render (kernels_xworld.hip): U-byte runs gathered through
//                         L2, assembled in LDS in output order, border-column bytes patched in, one non-temporal 16-byte
//                         store per lane
// The frames of the done list's envs (new episodes) take the same three stages over the list, on the reset's queue.

// threads per workgroup of the gather kernels (A/B hook: -DEGO_BS=...).  Round 3: 256 threads x 4 chunks = 16 KB spans, four waves
// per barrier: r = 3 colour 0.236 -> 0.228 ms per step, +2 .. 6 % on every geometry tried; 128 (round 2) and 512 lose
#ifndef EGO_BS
#define EGO_BS 256
#endif
// Round 4, r >= 5: a wavefront computes exactly the units its own lanes' pieces read and hands them over lane to lane
// (ds_bpermute) -- no LDS arrays, no barrier between the two phases, the four wavefronts of a workgroup run independently up to
// the one barrier in front of the stores, 3-4 KB less LDS per workgroup (r = 7: 7 -> 8 waves per SIMD).  Measured on one box
// (same run, both builds; the whole-batch render's four launches): r = 5 231.3 -> 221.6 us, r = 7 247.0 -> 245.4 us, but r = 3
// 202.8 -> 208.4 us (fifteen ds_bpermute per lane against ten LDS reads, and only 40 of a wavefront's 64 lanes hold a unit):
// r = 3 keeps round 3's hand-over through LDS.
template <int R> struct EgoUnitShfl { static constexpr bool value = R >= 5; };

// a square's pixels in the span path's sources (ego_tab3, the goal-cell cache): [channel][U rows][UP bytes], rows padded to whole
// 16-byte pieces
template <int R>
struct EgoSq {
    static constexpr int U = 84 / R, UD = U / 4, UDP = (UD + 3) & ~3, UP = 4 * UDP, CBP = U * UP, RR = R * R, PBP = RR * CBP;
};

// lane j of the wavefront appends (a, b) when flag: one atomic per wavefront
__device__ __forceinline__ void ego_wave_append(bool flag, uint32_t a, uint32_t b, uint2 *list, int32_t *count) {
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (flag) list[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(a, b);
}

// 64 envs per workgroup: all four wavefronts stage their grids (and the entity types) in LDS, the first one then walks them
// LIST: the envs of the done list (the frames of new episodes, drawn on the reset's queue) instead of the whole batch, 16 per
// workgroup: beside the whole-batch gather, whose 13 workgroups per CU leave about one of their own LDS allocations free, a
// workgroup that asks for more (64 envs: 13 / 17 / 24 KB at r = 3 / 5 / 7) is not placed until the gather drains
template <bool LIST> struct EgoCellsGeom { static constexpr int EPW = LIST ? 16 : 64; };
// ALL_MISS (list of freshly reset envs whose goal images are being redrawn beside this: xw_ego_list_front_kernel): every goal
// cell in view goes on the miss list, the cache bits are not looked at
// NW: wavefronts per workgroup (they share the walk over the view cells of the same envs)
template <int R, bool LIST, bool ALL_MISS, int NW = 4>
__device__ __forceinline__ void ego_cells_body(const XwParams &p, const uint8_t *map, int skip_term, const int32_t *count_now, int bid, uint4 *smem4) {
    constexpr int EPW = EgoCellsGeom<LIST>::EPW;
    const int D = p.max_dim, cells = D * D, tid = threadIdx.x, lane = tid & 63;
    uint16_t *s_code = reinterpret_cast<uint16_t *>(smem4);                // [EPW][cells]
    uint8_t *s_type = reinterpret_cast<uint8_t *>(s_code + EPW * cells);   // [EPW][cells] type of the entity in a cell, 3 = none
    __shared__ uint4 s_gc[EPW];                                             // the envs' goal slot -> cell tables
    __shared__ uint32_t s_sq[EPW][R * R];                                   // the cell words, frame order
    const int e_base = bid * EPW, total = LIST ? *count_now : p.n;
    if (e_base >= total) return;
    EGO_C0();
    const int n_here = total - e_base < EPW ? total - e_base : EPW;
    uint8_t *s_itype = s_type + EPW * cells;                                // [n_icons]
    uint8_t *s_cls = s_itype + ((p.n_icons + 15) & ~15);                   // [n_icons + 2]
    __shared__ uint8_t s_map[8 * R * R + 8 * R];
    __shared__ unsigned long long s_shadow[64];                            // per env: the shadow mask, a quarter from each wavefront
    if (tid < 64) s_shadow[tid] = 0;
    const bool valid = lane < EPW && e_base + lane < total;
    const int li = valid ? e_base + lane : total - 1;
    const int e = LIST ? p.done_list[li] : li, ec = e;
    int axy = 0, dir = 0, term = 0;
    int fresh = 0;
    { axy = p.agent_xy[ec]; dir = p.agent_dir[ec] & 3; term = p.term_flag[ec]; fresh = p.fresh[ec]; }
    for (int i = tid; i < p.n_icons; i += 64 * NW) s_itype[i] = p.icon_type[i];
    for (int i = tid; i < p.n_icons + 2; i += 64 * NW) s_cls[i] = p.ego_cls[i];
    for (int i = tid; i < 8 * R * R + 8 * R; i += 64 * NW) s_map[i] = map[i];
    if (LIST) {
#pragma unroll 4
        for (int i = tid; i < n_here * cells; i += 64 * NW) {
            const int le = i / cells;
            s_code[i] = (uint16_t)(p.grid[(size_t)p.done_list[e_base + le] * cells + (i - le * cells)] & CELL_ICON_MASK);
        }
    } else if (n_here == 64) {                              // (whole batch: EPW = 64)
        // 64 consecutive grids = 128 * cells contiguous bytes, a multiple of 16: a few 16-byte loads per lane, all in flight
        // (the element-wise loop below is a chain of a dozen dependent round trips)
        const uint4 *g4 = reinterpret_cast<const uint4 *>(p.grid + (size_t)e_base * cells);
        uint4 *s4 = reinterpret_cast<uint4 *>(s_code);
        const uint32_t m2 = CELL_ICON_MASK | CELL_ICON_MASK << 16;
#pragma unroll 4
        for (int i = tid; i < 8 * cells; i += 64 * NW) { uint4 v = g4[i]; v.x &= m2; v.y &= m2; v.z &= m2; v.w &= m2; s4[i] = v; }
    } else {
        for (int i = tid; i < n_here * cells; i += 64 * NW) s_code[i] = (uint16_t)(p.grid[(size_t)e_base * cells + i] & CELL_ICON_MASK);
    }
    static_assert(XW_MAX_GOALS == 16, "one uint4 per env");
    if (tid >= 64 && tid < 64 + n_here) s_gc[tid - 64] = reinterpret_cast<const uint4 *>(p.goal_cells)[LIST ? p.done_list[e_base + tid - 64] : e_base + tid - 64];
    __syncthreads();
    EGO_C(0);
    for (int i = tid; i < n_here * cells; i += 64 * NW) { const int code = s_code[i]; s_type[i] = code ? s_itype[code - 1] : (uint8_t)3; }
    __syncthreads();
    // the goal slot of a cell rides in its type byte (bits 2-5): one LDS read in the walk below instead of a search through the
    // env's sixteen slots per visible goal (that search was a third of the kernel's instructions)
    for (int i = tid; i < n_here * XW_MAX_GOALS; i += 64 * NW) {
        const int le = i / XW_MAX_GOALS, slot = i - le * XW_MAX_GOALS;
        const int cell = reinterpret_cast<const uint8_t *>(&s_gc[le])[slot];
        if (cell < cells) s_type[le * cells + cell] |= (uint8_t)(slot << 2);
    }
    __syncthreads();
    EGO_C(1);
    // The walk: lane = env, and every wavefront of the workgroup takes a share of the r * r view cells of the same envs (one
    // wavefront walking them all was 3.4 / 6.8 / 12.6 thousand instructions at r = 3 / 5 / 7 -- issue-bound with the other
    // three gone, and at r = 7 more code than the instruction cache holds)
    constexpr int Q = (R * R + NW - 1) / NW;
    const int kb = (tid >> 6) * Q;
    const bool active = valid && !(skip_term && term);
    const int ax = axy & 0xffff, ay = axy >> 16;
    const uint16_t *code_e = s_code + (valid ? lane : 0) * cells;
    const uint8_t *type_e = s_type + (valid ? lane : 0) * cells;
    auto is_block = [&](int x, int y) { return (unsigned)x < (unsigned)D && (unsigned)y < (unsigned)D && (type_e[y * D + x] & 3) == 1; };
    // XMap::image_masking (xmap.cpp:273-362), as in the kernel above
    constexpr int r = R;
    int major_x = 0, major_y = 0, minor_x = 0, minor_y = 0, scan_x0 = 0, scan_y0 = 0, xa = ax + r, ya = ay + r;
    if (dir == 0) { xa += r / 2; major_y = 1; minor_x = 1; }
    else if (dir == 3) { ya -= r / 2; major_x = 1; minor_y = -1; scan_y0 = r - 1; }
    else if (dir == 2) { xa -= r / 2; major_y = 1; minor_x = -1; scan_x0 = r - 1; }
    else { ya += r / 2; major_x = 1; minor_y = 1; }
    const int x_st = xa - r / 2, y_st = ya - r / 2;
    uint32_t ray = (1u << r) - 1u;                              // bit t: scan line t starts in the light
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int o = side ? 1 : -1;
        bool block = false;
        int rx = ax, ry = ay;
#pragma unroll
        for (int k = 1; k <= r / 2; ++k) {
            rx += o * major_x; ry += o * major_y;
            if (block) ray &= ~(1u << (r / 2 + o * k));
            if (is_block(rx, ry)) block = true;
        }
    }
    // bit k: view cell k lies behind a wall.  The r scan lines are independent: each of the workgroup's four wavefronts (they walk
    // different cells of the SAME envs) takes every fourth line and the masks meet in LDS -- repeated by all four, the scan was a
    // third of a wavefront's instructions at r = 7 (round 5: cells kernel 42 -> 39 us there, 156 -> 87 VGPRs)
    unsigned long long shadow = 0;
    {
        unsigned long long part = 0;
        const int wv = tid >> 6;
#pragma unroll
        for (int t = 0; t < r; ++t) {
            if (t % NW != wv) continue;                         // (uniform per wavefront)
            bool block = !((ray >> t) & 1u);
            int cx = scan_x0 + t * major_x, cy = scan_y0 + t * major_y;
#pragma unroll
            for (int j = 0; j < r; ++j) {
                if (block) part |= 1ull << (cy * r + cx);
                if (is_block(x_st - r + cx, y_st - r + cy)) block = true;
                cx += minor_x; cx = cx < 0 ? cx + r : (cx >= r ? cx - r : cx);
                cy += minor_y; cy = cy < 0 ? cy + r : (cy >= r ? cy - r : cy);
            }
        }
        EGO_C(7);
        if (part != 0) atomicOr(&s_shadow[lane], part);
        __syncthreads();
        shadow = s_shadow[lane];
        EGO_C(8);
    }
    if (p.no_wall_shadow) shadow = 0;
    uint32_t *info_e = p.ego_cellinfo + (size_t)ec * (r * r);
    const uint32_t *valid_e = p.ego_cache_valid + (size_t)ec * p.ego_cache_words;
    const uint32_t cls_white = s_cls[p.n_icons], cls_black = s_cls[p.n_icons + 1];
    // The words are stored in FRAME order (square fy * r + fx) and each carries all the gather needs beside the image:
    // bits 24-25 the heading, 26 "finished by this step", 27-28 fresh[], 29 / 30 the square's first row / column is a border
    // line of this heading -- the gather reads nothing else of the env, which a reset on the other queue may be rewriting.
    // A goal: bit 15, bits 0-3 its slot, bits 4-9 the view cell (the cache is indexed by it).
    constexpr int RL = 4 * r * r, CL = RL + 4 * r, INV = CL + 4 * r;
    const uint32_t hd = (uint32_t)dir << 24 | (term ? 1u << 26 : 0u) | ((uint32_t)fresh & 3u) << 27;
    unsigned long long goal_mask = 0;                           // this wavefront's view cells that show a goal (r * r <= 49)
    uint8_t gslot[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int k = kb + j;
        if (k >= r * r) break;
        const int gx = x_st - r + k % r, gy = y_st - r + k / r;
        uint32_t info = (uint32_t)((p.n_icons + 1) * 4 + dir) | cls_black << 16;   // outside the map, or in a wall's shadow: black
        int slot = 0;
        if (active && (unsigned)gx < (unsigned)D && (unsigned)gy < (unsigned)D && !((shadow >> k) & 1ull)) {
            const int code = code_e[gy * D + gx], ty = type_e[gy * D + gx];
            if (code == 0) info = (uint32_t)(p.n_icons * 4 + dir) | cls_white << 16;
            else if ((ty & 3) != 0) info = (uint32_t)((code - 1) * 4 + dir) | (uint32_t)s_cls[code - 1] << 16;
            else {                                              // a goal: this env's warped copy, through the cache
                slot = ty >> 2;
                info = 0x8000u | (uint32_t)slot | (uint32_t)k << 4 | 0xffu << 16;
                goal_mask |= 1ull << k;
            }
        }
        gslot[j] = (uint8_t)slot;
        const int f = s_map[INV + dir * (r * r) + k];
        const uint32_t lines = (s_map[RL + dir * r + f / r] != 0xff ? 1u << 29 : 0u) | (s_map[CL + dir * r + f % r] != 0xff ? 1u << 30 : 0u);
        if (valid) s_sq[lane][f] = info | hd | lines;          // (LIST: lanes past EPW have no row)
        if (active) info_e[f] = info | hd | lines;
    }
    EGO_C(2);
    __syncthreads();
    EGO_C(3);                                            // (a square's word needs its neighbours', other wavefronts' work)
    // What the gather reads, per square -- two words.
    // .x: where its pixels come from (bits 0-22, 16-byte units: into ego_tab3, keyed by the classes of the cell, the one above
    // and the one to the left -- the cell's own where the neighbour does not show in this square or is a goal -- or, bit 23,
    // into this env's part of the goal-cell cache), bit 24 / 25 its border row / column blends a goal's image (the cell above /
    // to the left shows a goal): the gather places that line itself, from the goal's cache entry; 26 a border row crosses a
    // border column here; 27 finished by this step; 28-29 fresh[]; 30-31 flat.
    // .y: what the gather places itself.  Bits 24 / 25 of .x: the cache entry (slot * r * r + view cell) * 4 + heading of the
    // goal above (bits 0-11: its BELOW line) and of the goal to the left (bits 12-23: its RIGHT line); both lines were
    // evaluated on the real view and hold the crossing pixel.  Bit 26 alone: the crossing pixel itself, B | G << 8 | R << 16
    // from ego_xtab (four classes: the same in every env), or, bit 31, the entry of the goal above left (its DIAG pixel).
    // A square that shows a goal carries no flags: its cache entry holds its border row, column and crossing as well.
    // (Why a goal's lines can be cached: the entry is keyed by (goal slot, view cell, heading), which fixes the agent's cell --
    // and with it the whole view, the map being constant over an episode but for the agent; whatever changes a map or a
    // pose redraws the goal images, which clears the env's valid bits: warp_goals_body.)
    if (!active && valid) {
        for (int f = kb; f < kb + Q && f < r * r; ++f) p.ego_cellsrc[(size_t)e * (r * r) + f] = make_uint2(1u << 27, 0u);      // (skipped: finished by this step)
    }
    if (active) {
        typedef EgoSq<r> Sq;
        const uint32_t nc = (uint32_t)p.ego_ncls, ch_n = (uint32_t)p.channels, entry16 = p.ego_cache_entry / 16;
        uint2 *src_e = p.ego_cellsrc + (size_t)e * (r * r);
        auto eidx = [&](uint32_t wg) { return ((wg & 0xfu) * (r * r) + ((wg >> 4) & 0x3fu)) * 4u + (uint32_t)dir; };
        uint32_t sx[Q], sy[Q];
        int fi[Q];                                              // index into ego_flat of a square's table entry, -1: a goal's square
        int xi[Q];                                              // index into ego_xtab of a square's crossing pixel, -1: none
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int f = kb + j;
            sx[j] = 0; sy[j] = 0; xi[j] = -1; fi[j] = -1;
            if (f >= r * r) break;
            const uint32_t w = s_sq[lane][f], wa = f >= r ? s_sq[lane][f - r] : w, wl = f % r ? s_sq[lane][f - 1] : w;
            const uint32_t wd = (f >= r && f % r) ? s_sq[lane][f - r - 1] : w;
            const bool rowb = (w >> 29 & 1u) != 0, colb = (w >> 30 & 1u) != 0, goal = (w & 0x8000u) != 0;
            const bool ga = !goal && rowb && (wa & 0x8000u), gl = !goal && colb && (wl & 0x8000u), cross = !goal && rowb && colb;
            const uint32_t c = (w >> 16) & 0xffu, ca = rowb && !ga ? (wa >> 16) & 0xffu : c, cl = colb && !gl ? (wl >> 16) & 0xffu : c;
            const uint32_t key = (((uint32_t)dir * nc + c) * nc + ca) * nc + cl;
            const uint32_t off = goal ? eidx(w) * entry16 : key * ch_n * (Sq::PBP / 16) + f * (Sq::CBP / 16);
            // bits 30-31 (added below): the table entry is one flat colour (1: 255, 2: 0) -- the gather reads the shared constant line instead
            if (!goal) fi[j] = (int)(key * (r * r) + f);
            sx[j] = off | (goal ? 1u << 23 : 0u) | (ga ? 1u << 24 : 0u) | (gl ? 1u << 25 : 0u) | (cross ? 1u << 26 : 0u) |
                    (term ? 1u << 27 : 0u) | ((uint32_t)fresh & 3u) << 28;
            if (ga) sy[j] |= eidx(wa);
            if (gl) sy[j] |= eidx(wl) << 12;
            if (cross && !ga && !gl) {
                if (wd & 0x8000u) sy[j] = 1u << 31 | eidx(wd);
                else xi[j] = (int)((((key * nc) + ((wd >> 16) & 0xffu)) * (r * r)) + f);
            }
        }
        EGO_C(9);
        // (every table read of this wavefront's squares in flight together, no branch around any: read one by one inside the loop
        // above -- a branch and a wait per square -- the flat bytes were up to Q dependent round trips, 8 us of this kernel at r = 7)
        uint32_t xv[Q];
        uint8_t fv[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) { xv[j] = p.ego_xtab[xi[j] >= 0 ? xi[j] : 0]; fv[j] = p.ego_flat[fi[j] >= 0 ? fi[j] : 0]; }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int f = kb + j;
            if (f >= r * r) break;
            src_e[f] = make_uint2(sx[j] | (fi[j] >= 0 ? (uint32_t)fv[j] << 30 : 0u), xi[j] >= 0 ? xv[j] : sy[j]);
        }
    }
    EGO_C(4);
    // the cache bits of this wavefront's goal cells, fetched together, then one list append for its whole lot (one atomic per view
    // cell was up to r * r dependent round trips)
    unsigned long long miss = 0;                                // bit k: view cell k shows a goal whose square is not cached
    if (ALL_MISS) {
        miss = goal_mask;
    } else {
        uint32_t vw[Q];                                         // (short-lived: every read in flight, then folded into the mask)
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int k = kb + j;
            const bool goal = (goal_mask >> k) & 1ull;
            const int bit = (gslot[j] * r * r + k) * 4 + dir;
            vw[j] = *(goal ? valid_e + (bit >> 5) : p.ego_cache_valid);       // (no branch around the read; the lanes without a goal share one line)
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const int k = kb + j;
            const bool goal = (goal_mask >> k) & 1ull;
            const int bit = (gslot[j] * r * r + k) * 4 + dir;
            if (goal && !((vw[j] >> (bit & 31)) & 1u)) miss |= 1ull << k;
        }
    }
    EGO_C(5);
    int total_miss = 0;
#pragma unroll
    for (int j = 0; j < Q; ++j) total_miss += __popcll(__ballot((miss >> (kb + j)) & 1ull));
    if (total_miss == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(p.ego_miss_count, total_miss);
    base = __shfl(base, 0);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int k = kb + j;
        const bool m = (miss >> k) & 1ull;
        const unsigned long long mk = __ballot(m);
        if (m) p.ego_miss[base + __popcll(mk & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)e, (uint32_t)(k | gslot[j] << 8 | dir << 16));
        base += __popcll(mk);
    }
    EGO_C(6);
}

// (A/B hook -DEGO_CELLS_NW=8 | 16: the whole batch's workgroups with more wavefronts per 64 envs -- measured, no gain: profiles/r5)
#ifndef EGO_CELLS_NW
#define EGO_CELLS_NW 4
#endif
template <bool LIST> struct EgoCellsWaves { static constexpr int NW = LIST ? 4 : EGO_CELLS_NW; };
template <int R, bool LIST>
__global__ __launch_bounds__(64 * EgoCellsWaves<LIST>::NW) void xw_ego_cells_kernel(XwParams p, const uint8_t *map, int skip_term, const int32_t *count_now, int publish_step) {
    extern __shared__ uint4 smem4[];
    // (xwb_step_autoreset: this kernel running = the step kernel before it is complete; the reset's queue waits for that)
    if (publish_step && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 1, p.sig_epoch);
    ego_cells_body<R, LIST, false, EgoCellsWaves<LIST>::NW>(p, map, skip_term, count_now, (int)blockIdx.x, smem4);
}

// A cache entry [env][goal slot][view cell][heading] on the span path (EgoEntry): everything of the frame that blends this goal's
// image while it shows in that view cell --
//   SQ     the square of the frame the cell occupies, in EgoSq's layout ([channel][U rows][UP bytes]), its border row, border
//          column and crossing pixel included;
//   BELOW  [channel][U] the first row of the square below, where that is a border row (it blends the goal with the cell below);
//   RIGHT  [channel][U] the first column of the square to the right, where that is a border column;
//   DIAG   [channel] the first pixel of the square below right, where a border row crosses a border column (four cells).
// The lines are evaluated on the env's real view (ego_cellinfo, which the cells kernel queued before this one wrote): the entry's
// key fixes the agent's cell and heading, so for the rest of the episode the same key means the same view (xw_ego_cells_kernel).
template <int CH, int R>
struct EgoEntry {
    typedef EgoSq<R> Q;
    static constexpr int BELOW = CH * Q::CBP, RIGHT = BELOW + CH * Q::U, DIAG = RIGHT + CH * Q::U, BYTES = (DIAG + 4 + 15) & ~15;
};

// The goal cells the cache does not hold yet (ego_miss): PARTS workgroups per listed cell; a workgroup takes a contiguous share of
// the entry's pixels -- the square's U * U, then the 2 U + 1 of the BELOW / RIGHT / DIAG lines -- 256 at a time.
template <int CH, int R>
__device__ __forceinline__ void ego_miss_body(const XwParams &p, const uint32_t *atlas4, const uint16_t *layout, const uint8_t *map,
                                              int block, int nblocks, EgoTap (*s_row)[3], EgoTap (*s_col)[3]) {
    constexpr int U = 84 / R, O = R * U, O4 = O, RR = R * R;
    // Workgroups per goal cell: as few as give every lane one pixel -- r = 7: the entry's 169 pixels are ONE workgroup's single pass,
    // r = 5: 289 pixels in two workgroups, r = 3: 841 pixels in four.  (Round 5, same box, render's four launches: four workgroups
    // per cell at every radius 208.4 / 228.8 us at r = 5 / 7, this 201.5 / 215.8; r = 3 with 4 / 2 / 1 workgroups of 1 / 2 / 4 passes:
    // 184.8 / 185.5 / 188.1 us -- profiles/r5/ego_concurrent_eval_experiments.txt.)
#ifndef EGO_EVAL_PARTS3
#define EGO_EVAL_PARTS3 4
#endif
    constexpr int PARTS = R >= 7 ? 1 : (R >= 5 ? 2 : EGO_EVAL_PARTS3);
    constexpr int NX = 2 * U + 1, NP = U * U + NX, PPT = (NP + PARTS - 1) / PARTS;      // the square's pixels, then its lines
    typedef EgoEntry<CH, R> E;
    // (s_row / s_col: the kernel's)
    // The count and (speculatively) the first item come in one round trip, the flag rows of all four headings, the view-cell ->
    // square maps and the class images in the next: per goal cell the chain is item -> its env's cell words -> pixel reads -> stores
    __shared__ EgoCell s_cells[RR];
    __shared__ uint16_t s_flags[4][2][84];                 // [heading][row terms | column terms]
    __shared__ uint8_t s_inv[4 * RR], s_fwd[4 * RR];
    __shared__ uint2 s_clsimg[4 * 16];
    const int tid = threadIdx.x, part = block % PARTS, first = block / PARTS;
    const int cap = p.n * (p.num_goals < RR ? p.num_goals : RR);
    uint2 item = p.ego_miss[first < cap ? first : cap - 1];
    const int cnt = *p.ego_miss_count;
    // (the tables are requested before the count is looked at: waiting for it first put one more round trip in front of them;
    // the workgroups that then leave have asked for a few hundred bytes for nothing)
    const int lw = ego_layout_words(O4, R);
    constexpr int NF = (4 * 2 * O + 255) / 256;
    uint16_t fl[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const int i = tid + q * 256, d = i / (2 * O), rem = i - d * 2 * O;
        fl[q] = i < 4 * 2 * O ? layout[d * lw + (rem / O) * O4 + rem % O] : (uint16_t)0;
    }
    const uint8_t inv = tid < 4 * RR ? map[8 * R + 4 * RR + tid] : (uint8_t)0, fwd = tid < 4 * RR ? map[tid] : (uint8_t)0;
    const uint2 ci = p.ego_clsimg[tid < 4 * p.ego_ncls ? tid : 0];
    // (speculatively, with the item: the cell words of its env)
    uint32_t info = tid < RR ? p.ego_cellinfo[(size_t)(item.x < (uint32_t)p.n ? item.x : 0u) * RR + tid] : 0u;      // (a slot past the count holds anything)
    if (first >= cnt) return;                              // (most workgroups: the list is short)
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const int i = tid + q * 256, d = i / (2 * O), rem = i - d * 2 * O;
        if (i < 4 * 2 * O) s_flags[d][rem / O][rem % O] = fl[q];
    }
    if (tid < 4 * RR) { s_inv[tid] = inv; s_fwd[tid] = fwd; }
    if (tid < 4 * 16) s_clsimg[tid] = ci;
    const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
    for (int it = first; it < cnt; it += nblocks / PARTS) {
        if (it != first) { item = p.ego_miss[it]; info = tid < RR ? p.ego_cellinfo[(size_t)item.x * RR + tid] : 0u; }
        const int e = (int)item.x, k = item.y & 0xff, slot = (item.y >> 8) & 0xff, dir = (item.y >> 16) & 3;
        __syncthreads();
        // the env's view: lane = square of the frame, stored under the view cell it shows
        if (tid < RR) {
            EgoCell c;
            if (info & 0x8000u) c = EgoCell{p.goal_img + ((size_t)e * p.num_goals + (info & 0xfu)) * 4096, -1, -1};
            else { const uint2 q = s_clsimg[dir * p.ego_ncls + (int)((info >> 16) & 0xffu)]; c = EgoCell{atlas4 + q.x, (int)q.y, -1}; }
            s_cells[s_fwd[dir * RR + tid]] = c;
        }
        __syncthreads();
        const int f = s_inv[dir * RR + k];                                   // the square view cell k occupies
        const int fx = f % R, fy = f / R, x0 = fx * U, y0 = fy * U;
        const uint16_t *rt = s_flags[dir][0], *ct = s_flags[dir][1];
        EgoCtx ctx{s_cells, white, black, R, 64 * R, dir};
        const int entry = (slot * RR + k) * 4 + dir;
        uint8_t *dst = p.ego_cache + ((size_t)e * p.num_goals * (RR * 4) + entry) * p.ego_cache_entry;
        // The entry's pixels in an order that keeps the two code paths of ego_pixel (~400 VALU instructions each) in different
        // wavefronts: first the pixels whose sixteen taps all lie in the goal's image (rows [r0, r1) x columns [c0, c1) of the square),
        // then the rest -- the square's border row / column and, at the frame's edges, its edge rows / columns, then the three lines.
        // In the natural order every wavefront holds a border-column pixel (one every U lanes) and runs BOTH paths.  The rows / columns
        // that take the general path are a prefix and a suffix of the square; if they ever were not, everything takes the general
        // path, which is right for every pixel.
        // (r = 3 only -- same box, render's four launches: 180.0 against 183.1 us there, 198.9 / 214.9 against 197.7 / 212.7 at r = 5 / 7)
        constexpr bool ORDERED = R == 3;
        int r0 = 0, r1 = 0, c0 = 0, c1 = 0;
        if (ORDERED) {
            const int ln = tid & 63;
            const unsigned long long all = (1ull << U) - 1ull;
            const unsigned long long rm = __ballot(ln < U && (rt[y0 + (ln < U ? ln : 0)] & (EGO_BORDER | EGO_EDGE))) & all;
            const unsigned long long cm = __ballot(ln < U && (ct[x0 + (ln < U ? ln : 0)] & (EGO_BORDER | EGO_EDGE))) & all;
            auto range = [&](unsigned long long m, int &lo, int &hi) {
                lo = m == all ? U : __ffsll((long long)(~m & all)) - 1;                            // leading rows of the general path
                int t = 0;
                while (t < U - lo && ((m >> (U - 1 - t)) & 1ull)) ++t;                              // trailing ones (uniform: a scalar loop)
                hi = U - t;
                const unsigned long long want = ((1ull << lo) - 1ull) | (all & ~((1ull << hi) - 1ull));
                if (m != want) { lo = 0; hi = 0; }                                                  // not a prefix and a suffix: no fast pixels
            };
            range(rm, r0, r1); range(cm, c0, c1);
            if (r1 <= r0 || c1 <= c0) { r0 = r1 = 0; c0 = c1 = 0; }
        }
        const int h1 = r1 - r0, w1 = c1 - c0, n_one = h1 * w1, n_a = (U - h1) * U, gw = U - w1, n_b = h1 * gw;
        for (int qq = tid; qq < PPT; qq += 256) {
            const int q = part * PPT + qq;
            if (q >= NP) break;
            int px = 0, py = 0, x = -1;
            bool one = false;
            if (!ORDERED) {
                if (q < U * U) { py = q / U; px = q - py * U; one = !(((uint32_t)rt[y0 + py] | (uint32_t)ct[x0 + px]) & (EGO_BORDER | EGO_EDGE)); }
                else x = q - U * U;
            }
            else if (q < n_one) { const int i = q / w1; py = r0 + i; px = c0 + (q - i * w1); one = true; }
            else if (q < n_one + n_a) { const int g = q - n_one, a = g / U; px = g - a * U; py = a < r0 ? a : r1 + (a - r0); }
            else if (q < n_one + n_a + n_b) { const int g = q - n_one - n_a, i = g / gw, b = g - i * gw; py = r0 + i; px = b < c0 ? b : c1 + (b - c0); }
            else x = q - n_one - n_a - n_b;
            // (a pixel of the border row / column blends the neighbours; an edge pixel has taps outside the view; the rest lie in cell k.
            // ONE call of the general path for the square's pixels and the three lines: a wavefront that holds several kinds runs it once)
            uint8_t *d = dst;
            int plane = EgoSq<R>::CBP, o = py * EgoSq<R>::UP + px, ox = x0 + px, oy = y0 + py;
            bool ok = true;
            if (x >= 0) {
                const bool below = x < U, right = !below && x < 2 * U;
                const int t = below ? x : x - U;
                ox = below ? x0 + t : x0 + U; oy = below ? y0 + U : (right ? y0 + t : y0 + U);
                ok = below ? (fy + 1 < R && (rt[oy] & EGO_BORDER)) : (right ? (fx + 1 < R && (ct[ox] & EGO_BORDER))
                           : (fx + 1 < R && fy + 1 < R && (rt[oy] & EGO_BORDER) && (ct[ox] & EGO_BORDER)));
                d = dst + (below ? E::BELOW : (right ? E::RIGHT : E::DIAG)); plane = below || right ? U : 1; o = below || right ? t : 0;
            }
            if (one) ego_pixel<CH, -1, true>(ctx, s_row, s_col, dst, EgoSq<R>::CBP, o, ox, oy, k);
            else if (ok) ego_pixel<CH, -1, false>(ctx, s_row, s_col, d, plane, o, ox, oy, 0);
        }
        // (the bit is read by kernels launched after this one: all parts are complete by then)
        if (tid == 0 && part == 0) atomicOr(p.ego_cache_valid + (size_t)e * p.ego_cache_words + (entry >> 5), 1u << (entry & 31));
    }
}

// The goal cells the cache lacks, four workgroups each.
template <int CH, int R>
__global__ __launch_bounds__(256) void xw_ego_eval_kernel(XwParams p, const uint32_t *atlas4, const uint16_t *layout, const uint8_t *map,
                                                          int publish, const EgoTap *comp) {
    // (this kernel running = the cells kernel queued before it is complete: xw_device.h, epochs instead of event packets)
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 5, p.sig_epoch);
    __shared__ EgoTap s_row[84][3], s_col[84][3];         // composed taps
    {   // (the host composed them: xw_ego_tables -- requested here, in front of everything else the body waits for)
        constexpr int O = R * (84 / R);
        for (int i = threadIdx.x; i < 3 * O; i += 256) { (&s_row[0][0])[i] = comp[i]; (&s_col[0][0])[i] = comp[3 * O + i]; }
    }
    ego_miss_body<CH, R>(p, atlas4, layout, map, (int)blockIdx.x, (int)gridDim.x, s_row, s_col);
}

// ego_cell_of_info: what a cell word of xw_ego_cells_kernel's ego_cellinfo shows (the table kernels below)
__device__ __forceinline__ EgoCell ego_cell_of_info(const XwParams &p, const uint32_t *atlas4, uint32_t info, int e, int dir) {
    const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
    EgoCell c{black, 0, -1};
    const int t = (int)((info & 0x7fffu) >> 2);
    if (info & 0x8000u) c = EgoCell{p.goal_img + ((size_t)e * p.num_goals + (info & 0xfu)) * 4096, -1, -1};
    else if (t == p.n_icons) c.img = white;
    else if (t < p.n_icons) c = ego_icon_cell(p.icon_type, p.ego_agent_rot, atlas4, t, dir);
    return c;
}

// ego_clsimg [heading][class]: the image a class shows under a heading, as (pixel offset in the atlas, index mask)
__global__ __launch_bounds__(64) void xw_ego_build_clsimg_kernel(XwParams p, const uint32_t *atlas4, uint2 *out) {
    const int nc = p.ego_ncls, tid = threadIdx.x;
    if (tid >= 4 * nc) return;
    const EgoCell c = ego_cell_of_info(p, atlas4, (uint32_t)p.ego_cls_icon[tid % nc] << 2, 0, tid / nc);
    out[tid] = make_uint2((uint32_t)(c.img - atlas4), (uint32_t)c.mask);
}

// ego_xtab [heading][c][a][l][d][square]: the pixel where the border row and the border column of a square cross -- it blends the
// square's own cell (class c), the cell above (a), the one to the left (l) and the one above left (d), which no table of
// squares keyed by three classes can hold.  One workgroup per (heading, c, a, l, d); a lane per crossing would need a cell table
// of its own, so the crossings take turns (once per batch).
template <int CH, int R>
__global__ __launch_bounds__(64) void xw_ego_build_xtab_kernel(XwParams p, const uint32_t *atlas4, const EgoTap *comp, const uint16_t *layout,
                                                               const uint8_t *map, uint32_t *xtab) {
    constexpr int U = 84 / R, O = R * U, RR = R * R;
    __shared__ EgoTap s_row[84][3], s_col[84][3];
    __shared__ EgoCell s_cells[RR];
    __shared__ uint8_t s_px[4];
    const int tid = threadIdx.x, nc = p.ego_ncls;
    int id = blockIdx.x;
    const int d = id % nc; id /= nc;
    const int l = id % nc; id /= nc;
    const int a = id % nc; id /= nc;
    const int c = id % nc, dir = id / nc;
    for (int i = tid; i < 3 * O; i += 64) { (&s_row[0][0])[i] = comp[i]; (&s_col[0][0])[i] = comp[3 * O + i]; }
    const uint16_t *L = layout + (size_t)dir * ego_layout_words(O, R), *rt = L, *ct = L + O;
    const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
    for (int sq = 0; sq < RR; ++sq) {
        const int fy = sq / R, fx = sq % R;
        uint32_t v = 0;
        if (fy > 0 && fx > 0 && (rt[fy * U] & EGO_BORDER) && (ct[fx * U] & EGO_BORDER)) {       // (uniform)
            __syncthreads();
            if (tid < RR) {
                const int gy = tid / R, gx = tid % R;
                const int cls = (gy == fy - 1 && gx == fx) ? a : ((gy == fy && gx == fx - 1) ? l : ((gy == fy - 1 && gx == fx - 1) ? d : c));
                s_cells[map[dir * RR + tid]] = ego_cell_of_info(p, atlas4, (uint32_t)p.ego_cls_icon[cls] << 2, 0, dir);
            }
            __syncthreads();
            EgoCtx ctx{s_cells, white, black, R, 64 * R, dir};
            if (tid == 0) ego_pixel<CH, -1, false>(ctx, s_row, s_col, s_px, 1, 0, fx * U, fy * U, 0);
            __syncthreads();
            v = CH == 3 ? (uint32_t)s_px[0] | (uint32_t)s_px[1] << 8 | (uint32_t)s_px[2] << 16 : (uint32_t)s_px[0];
        }
        if (tid == 0) xtab[(size_t)blockIdx.x * RR + sq] = v;
    }
}

// ego_tab3: the squares of every constant-image neighbourhood.  Entry (heading, c, a, l, channel, square) = the pixels of that
// square of the frame when its cell shows class c's image, the cell above class a's and the cell to the left class l's: the
// square's first row / column, where that is a border line, blends two cells (the pixel where both cross blends four and is
// not in the table).  One workgroup per (heading, c, a, l, square).
template <int CH, int R>
__global__ __launch_bounds__(256) void xw_ego_build_squares_kernel(XwParams p, const uint32_t *atlas4, const EgoTap *tap_h1, const EgoTap *tap_v1,
                                                                 const EgoTap *tap_h2, const EgoTap *tap_v2, const uint8_t *map, uint8_t *tab3) {
    typedef EgoSq<R> Q;
    constexpr int U = Q::U, O = R * U, RR = R * R;
    __shared__ EgoTap s_row[84][3], s_col[84][3];
    __shared__ EgoCell s_cells[RR];
    const int tid = threadIdx.x, nc = p.ego_ncls;
    int id = blockIdx.x;
    const int sq = id % RR; id /= RR;
    const int l = id % nc; id /= nc;
    const int a = id % nc; id /= nc;
    const int c = id % nc, dir = id / nc;
    const int fy = sq / R, fx = sq % R;
    ego_compose_taps(s_row, s_col, tap_h1, tap_v1, tap_h2, tap_v2, O, tid, 256);
    if (tid < RR) {
        const int gy = tid / R, gx = tid % R;
        const int cls = (gy == fy - 1 && gx == fx) ? a : ((gy == fy && gx == fx - 1) ? l : c);
        s_cells[map[dir * RR + tid]] = ego_cell_of_info(p, atlas4, (uint32_t)p.ego_cls_icon[cls] << 2, 0, dir);
    }
    __syncthreads();
    const uint32_t *white = atlas4 + (size_t)p.n_icons * 4096, *black = white + 1;
    EgoCtx ctx{s_cells, white, black, R, 64 * R, dir};
    uint8_t *dst = tab3 + (((((size_t)dir * nc + c) * nc + a) * nc + l) * CH) * Q::PBP + (size_t)sq * Q::CBP;
    for (int j = tid; j < U * U; j += 256) {
        const int py = j / U, px = j - py * U;
        ego_pixel<CH, -1, false>(ctx, s_row, s_col, dst, Q::PBP, py * Q::UP + px, fx * U + px, fy * U + py, 0);
    }
}

template <int CH, int R, int ES, int PER_>
struct EgoSpanGeom {
    static constexpr int BS = EGO_BS, PER = PER_, SPAN = BS * PER;
    static constexpr int U = 84 / R, O = R * U;
    static constexpr unsigned FB = CH * O * O;
    static constexpr int BPC = 16 / ES;                                     // frame bytes behind one 16-byte chunk
    static constexpr int cpf = (int)FB / BPC;                               // chunks per frame
    static constexpr int SPE = (cpf + SPAN - 1) / SPAN;                     // list render: spans per env
    // (what ego_gather_span keeps in LDS, to within a few bytes)
    static constexpr int GB = 4 * O, SB = SPAN * BPC, NU = ((SB + GB - 1) / GB + 1) * R;
    static constexpr int LDS = GB + SB + GB + 4 * (SB / (int)FB + 2) + 12 * NU + 32;
};

// Chunks [cr, cr + nc) of env e0's frame and on into the next envs' (nc <= SPAN).
// What was measured on the way here (MI355X, 32 768 envs, 84 x 84 x 3; HBM time of the stores alone: 90 us; an empty kernel
// of this grid: 54 us), each a different wall at the same ~195 us:
//  - one lane per U-byte run from frame-planar tables: 417 VALU instructions per wave (a wave64 VALU instruction takes four
//    cycles: 229 us) and a separate pass for the border-column bytes;
//  - one lane per four runs: 112 VALU, but the texture addresser busy 75 % of the time -- a load costs about one cycle per
//    cache line its lanes touch, and 28-byte runs at 84-byte strides touch 29 lines per instruction;
//  - 16-byte pieces of square-contiguous sources, with staged cell words, border rows from (above, cell) line tables and
//    border-column bytes from (left, cell) ones: five dependent phases per workgroup, 4.7 us at 16 workgroups per CU;
//  - the same with the look-ups folded into one pass: 359 VALU per wave again (five 64-bit table addresses per unit).
// Hence this shape: the sources hold whole squares with their border row and column already in them (ego_tab3 is keyed by
// the classes of the cell, the one above and the one to the left), rows padded to whole 16-byte pieces; a UNIT is four
// consecutive frame rows of one square column (U is a multiple of four: one square, one plane, one env), 4 UP contiguous
// source bytes.  One lane per unit reads the square's two words and posts one address; one lane per piece loads 16 bytes and
// drops its dwords into output order in LDS; one barrier; 16-byte non-temporal stores.  Only where a goal is next to the
// cell (the line that blends its image lies in the goal's cache entry: EgoEntry) or where a border row crosses a border column
// (four cells: the pixel rides in the square's second word) does the unit's lane place a row or first dwords itself -- the
// pieces leave those dwords alone.
// flag_all: the context flag of every env touched (list render), -1: the cell words say.
struct EgoGatherLds { uint4 *out4; uint32_t *env; const uint8_t **usrc; int *uo; };
#define EGO_GATHER_LDS(G, R_, name) \
    __shared__ uint4 name##_out4[(G::GB + G::SB + G::GB) / 16 + 17]; \
    __shared__ uint32_t name##_env[G::SB / (int)G::FB + 2]; \
    __shared__ const uint8_t *name##_usrc[EgoUnitShfl<R_>::value ? 1 : G::NU]; \
    __shared__ int name##_uo[EgoUnitShfl<R_>::value ? 1 : G::NU]; \
    const EgoGatherLds name{name##_out4, name##_env, name##_usrc, name##_uo}
template <int CH, int R, bool CTX1, int ES, int PER>
__device__ __forceinline__ void ego_gather_span(const XwParams &p, const EgoGatherLds &lds, unsigned e0, unsigned cr, int nc, int skip_term, int flag_all) {
    typedef EgoSq<R> Q;
    constexpr int BS = EGO_BS, SPAN = BS * PER;
    constexpr int U = Q::U, UD = Q::UD, O = R * U, RR = R * R;
    constexpr unsigned PB = O * O, FB = CH * PB;                            // bytes per plane, per frame
    constexpr int BPC = 16 / ES;                                            // frame bytes behind one 16-byte chunk
    constexpr int SB = SPAN * BPC;                                          // ... behind one span
    constexpr unsigned GB = 4 * O, GPP = O / 4, GPF = CH * GPP;             // bytes per row group; groups per plane, per frame
    constexpr int NG = (SB + GB - 1) / GB + 1;                              // row groups a span can touch
    constexpr int NU = NG * R, PPU = Q::UDP, PPR = Q::UDP / 4;              // units (square column major), pieces per unit, per row
    constexpr int ITP = (NU * PPU + BS - 1) / BS;
    // SHFL: unit slots of a wavefront -- iteration `it` of the pieces loop reads the UPI units it * (BS / PPU) + wave * UPI +
    // [0, UPI); slot k = it * UPI + j is computed by lane k % 64 (its k / 64-th unit)
    constexpr bool SHFL = EgoUnitShfl<R>::value;
    constexpr int UPI = 64 / PPU, UPW = ITP * UPI, ITU = SHFL ? (UPW + 63) / 64 : (NU + BS - 1) / BS;
    static_assert(64 % PPU == 0 && BS % 64 == 0, "whole units per wavefront");
    constexpr int NE = SB / (int)FB + 2;                                    // envs a span can touch
    constexpr int cpf = (int)FB / BPC;
    typedef EgoEntry<CH, R> EN;
    static_assert(GB % 16 == 0 && U % 4 == 0, "aligned pieces");
    static_assert(4 * Q::UP <= 128, "a unit fits the constant line");
    static_assert(NE == EgoSpanGeom<CH, R, ES, PER>::SB / (int)EgoSpanGeom<CH, R, ES, PER>::FB + 2 && NU == EgoSpanGeom<CH, R, ES, PER>::NU, "EGO_GATHER_LDS sizes");
    uint4 *const s_out4 = lds.out4;                                         // [(GB + SB + GB) / 16 + 17] (+ the dump of dwords nobody wants, see below: 64 + 3 dwords)
    uint32_t *const s_env = lds.env;                                        // [NE] a cell word of each env: its flags
    const uint8_t **const s_usrc = lds.usrc;                                // [SHFL ? 1 : NU]
    int *const s_uo = lds.uo;                                               // [SHFL ? 1 : NU] the unit's first dword in s_out | flags << 24, -1: none
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_out4);
    const int tid = threadIdx.x;
    const unsigned br = cr * BPC, be = br + (unsigned)nc * BPC;             // bytes, from the start of env e0's frame
    const int ne = (int)((be - 1) / FB) + 1;
    typedef const unsigned int __attribute__((address_space(1))) *g_u32;
    typedef const unsigned char __attribute__((address_space(1))) *g_u8;
    typedef const u32x4 __attribute__((address_space(1))) *g_u32x4;
    const unsigned g0 = br / GB, g1 = (be + GB - 1) / GB;                   // row groups, counted from env e0's first
    if (tid >= BS - ne) s_env[BS - 1 - tid] = p.ego_cellsrc[((size_t)e0 + (BS - 1 - tid)) * RR].x;
    const size_t env_cache = (size_t)p.num_goals * (RR * 4) * p.ego_cache_entry;
    int uo[ITU];                                                            // the unit's first dword in s_out | flags << 24, -1: none
    const uint8_t *usrc_r[ITU];
    uint32_t pcb[ITU], prow[ITU][UD], pcol[ITU][4];                         // the patch data of this lane's units (see below)
#pragma unroll
    for (int iu = 0; iu < ITU; ++iu) {
        pcb[iu] = 0;
#pragma unroll
        for (int d = 0; d < UD; ++d) prow[iu][d] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) pcol[iu][j] = 0;
    }
#pragma unroll
    for (int iu = 0; iu < ITU; ++iu) {
        const int k = iu * 64 + (tid & 63), itk = k / UPI;
        const int ut = !SHFL ? iu * BS + tid : (k < UPW ? itk * (BS / PPU) + (tid >> 6) * UPI + (k - itk * UPI) : NU);
        const unsigned fx = (unsigned)ut / NG, gq = g0 + ((unsigned)ut - fx * NG);
        uo[iu] = -1;
        usrc_r[iu] = p.ego_tab3;
        if (ut < NU && gq < g1) {
            const unsigned le = gq / GPF, gi = gq - le * GPF, ch = gi / GPP, oy0 = 4u * (gi - ch * GPP), fy = oy0 / (unsigned)U, py0 = oy0 - fy * U;
            const uint2 ww = p.ego_cellsrc[((size_t)e0 + le) * RR + fy * R + fx];
            uint32_t w = ww.x;
            const uint32_t w2 = ww.y;
            if (skip_term && (w >> 27 & 1u)) w = 0;
            const bool cached = (w >> 23 & 1u) != 0;
            const uint32_t flat = w >> 30;
            const uint8_t *ecache = p.ego_cache + ((size_t)e0 + le) * env_cache;
            const uint8_t *base = cached ? ecache : p.ego_tab3;
            const uint8_t *from = base + (size_t)(w & 0x7fffffu) * 16 + ch * (cached ? (unsigned)Q::CBP : (unsigned)Q::PBP) + py0 * Q::UP;
            // a flat square: every unit of it is the same 4 * UP bytes -- one line shared by the whole batch (L1-resident)
            const uint8_t *usrc = flat ? p.ego_constline + (flat - 1u) * 128u : from;
            usrc_r[iu] = usrc;
            // what this lane places itself: bit 0 the first row (it blends the goal above), bit 1 the first dword of every row
            // (a border column that blends the goal to the left), bit 2 the first dword of the first row (the crossing)
            const bool f_row = (w >> 24 & 1u) && py0 == 0, f_col = (w >> 25 & 1u) != 0, f_x = (w >> 26 & 1u) && py0 == 0 && !f_col;
            uo[iu] = ((int)(GB + gq * GB - br) / 4 + (int)(fx * UD)) | (f_row ? 1 << 24 : 0) | (f_col ? 2 << 24 : 0) | (f_x ? 4 << 24 : 0);
            // What this lane will place itself (rare: a goal next to the cell; one byte per crossing) is fetched NOW, with the
            // cell words just read: the round trip runs under the barrier and the pieces' own loads instead of after them
            // (round 4: it was a dependent round trip at the end of nearly every workgroup, ~0.4 of its ~6 us).  Round 5: the
            // lines come from the goals' cache entries (EgoEntry) and the plain crossing pixel rides in the second cell word.
            if (f_row || f_col || f_x) {
                const uint8_t *src = usrc;
                if (f_col) pcb[iu] = *(g_u32)(ecache + (size_t)((w2 >> 12) & 0xfffu) * p.ego_cache_entry + EN::RIGHT + ch * U + py0);
                else if (!f_row) pcb[iu] = (w2 >> 31) ? (uint32_t)*(g_u8)(ecache + (size_t)(w2 & 0xfffu) * p.ego_cache_entry + EN::DIAG + ch) : (w2 >> (8 * ch)) & 0xffu;
                if (f_row) {
                    const uint8_t *row = ecache + (size_t)(w2 & 0xfffu) * p.ego_cache_entry + EN::BELOW + ch * U;
#pragma unroll
                    for (int d = 0; d < UD; ++d) prow[iu][d] = *(g_u32)(row + 4 * d);
                }
                if (f_col || f_x) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) pcol[iu][j] = *(g_u32)(src + j * Q::UP);
                }
            }
        }
        if (!SHFL && ut < NU) { s_uo[ut] = uo[iu]; s_usrc[ut] = usrc_r[iu]; }
    }
    if (!SHFL) __syncthreads();
    // ---- one lane per 16-byte piece
    {
        u32x4 q[ITP];
        int po[ITP];
        // (PPU divides BS: the place of a lane's piece inside its unit is the same in every iteration)
        static_assert(BS % PPU == 0, "a lane's piece index inside its unit does not change from iteration to iteration");
        const int pi0 = tid % PPU;
#pragma unroll
        for (int it = 0; it < ITP; ++it) {
            const int P = it * BS + tid, u = P / PPU;
            const uint8_t *from;
            if (SHFL) {
                // the unit of this piece sits in slot it * UPI + lane / PPU of this wavefront: lane (slot % 64)'s (slot / 64)-th
                const int reg = SHFL ? (it * UPI) / 64 : 0;    // (the loop is unrolled: a constant)
                const int from_lane = (it * UPI) % 64 + (tid & 63) / PPU;
                po[it] = __shfl(uo[reg], from_lane);
                const unsigned long long a = (unsigned long long)usrc_r[reg];
                const unsigned lo = (unsigned)__shfl((int)(unsigned)a, from_lane), hi = (unsigned)__shfl((int)(unsigned)(a >> 32), from_lane);
                from = (const uint8_t *)((unsigned long long)hi << 32 | lo) + 16 * pi0;
            } else {
                po[it] = s_uo[P < NU * PPU ? u : 0];
                if (P >= NU * PPU) po[it] = -1;
                from = s_usrc[P < NU * PPU ? u : 0] + 16 * pi0;
            }
            // (no branch around the load: all of a lane's pieces are in flight together; an idle lane reads the table's start)
            q[it] = *(g_u32x4)(po[it] >= 0 ? from : p.ego_tab3);
        }
        // Placing the dwords, r <= 5, is branch-free (round 4): a dword that is not this piece's to write -- an idle lane, the
        // padding of a row's last piece, a dword the unit lane places itself -- goes to a per-lane dump slot behind the span
        // instead of around a divergent branch (the loop was a dozen exec-mask regions per piece: 376 scalar instructions per
        // wavefront against 445 vector ones at r = 3; now 241 / 418).  Which dwords a lane may lose depends on its piece's
        // place in the unit, which is the same in every iteration.  Kernel trace, same box, both builds: gather r = 3
        // 133.2 -> 127.8 us, r = 5 render 232.8 -> 215.9 us; r = 7 154.4 -> 168.8 us -- its rows are ONE 12-byte piece, the
        // branchy form stores them with fewer, wider LDS writes -- so r = 7 keeps the branches.
        constexpr int LASTD = UD - 4 * (PPR - 1);                                      // dwords of a row's last piece
        constexpr bool BRANCH_FREE = R <= 5;
        const int j0 = pi0 / PPR, h0 = pi0 - j0 * PPR;
        if (BRANCH_FREE) {
            // (one dump slot per lane: sixty-four lanes storing to ONE address serialise)
            const int DUMP = (int)(GB + SB + GB) / 4 + (tid & 63);
            const int lane_off = j0 * (O / 4) + 4 * h0;
            const int kill_all = j0 == 0 ? 1 : 0;                                      // fl & 1: the unit lane places the whole first row
            const int kill_0 = h0 == 0 ? (2 | (j0 == 0 ? 4 : 0)) : 0;                  // fl & 2 / 4: ... the first dword of every / of the first row
            const bool pad2 = h0 == PPR - 1 && LASTD <= 2, pad3 = h0 == PPR - 1 && LASTD <= 3;
#pragma unroll
            for (int it = 0; it < ITP; ++it) {
                const int fl = po[it] >> 24;                                           // (-1 for an idle lane: every test below kills)
                const bool dead = po[it] < 0 || (fl & kill_all);
                const int base = dead ? DUMP : (po[it] & 0xffffff) + lane_off;
                s_out[(fl & kill_0) ? DUMP : base] = q[it].x;
                s_out[base + 1] = q[it].y;
                if (!(PPR == 1 && LASTD <= 2)) s_out[pad2 ? DUMP + 2 : base + 2] = q[it].z;      // (a row that is one piece: known at compile time)
                if (!(PPR == 1 && LASTD <= 3)) s_out[pad3 ? DUMP + 3 : base + 3] = q[it].w;
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITP; ++it) {
                if (po[it] < 0) continue;
                const uint32_t w[4] = {q[it].x, q[it].y, q[it].z, q[it].w};
                const int o = po[it] & 0xffffff, fl = po[it] >> 24;
                uint32_t *dst = s_out + o + j0 * (O / 4) + 4 * h0;
                if (fl == 0) {
                    dst[0] = w[0];
                    if (LASTD > 1 || h0 < PPR - 1) dst[1] = w[1];
                    if (LASTD > 2 || h0 < PPR - 1) dst[2] = w[2];
                    if (LASTD > 3 || h0 < PPR - 1) dst[3] = w[3];
                    continue;
                }
                const bool first = h0 == 0 && ((fl & 2) || (j0 == 0 && (fl & 4)));  // its first dword is the unit lane's
                if (j0 == 0 && (fl & 1)) continue;                                     // the whole row is
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (h0 == PPR - 1 && d >= LASTD) continue;                          // padding
                    if (d == 0 && first) continue;
                    dst[d] = w[d];
                }
            }
        }
    }
    // ---- the unit lanes place what the pieces left (rare: a goal in or next to the cell, a crossing), from what they fetched above
#pragma unroll
    for (int iu = 0; iu < ITU; ++iu) {
        if (uo[iu] < 0 || !(uo[iu] >> 24)) continue;
        const int o = uo[iu] & 0xffffff, fl = uo[iu] >> 24;
        const uint32_t cb = pcb[iu];
        if (fl & 1) {
#pragma unroll
            for (int d = 0; d < UD; ++d) {
                uint32_t w = prow[iu][d];
                if (d == 0 && (fl & 2)) w = (w & ~0xffu) | (cb & 0xffu);
                s_out[o + d] = w;
            }
        }
        if (fl & 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j == 0 ? (fl & 1) != 0 : !(fl & 2)) continue;
                s_out[o + j * (O / 4)] = (pcol[iu][j] & ~0xffu) | ((cb >> (8 * j)) & 0xffu);
            }
        }
    }
    __syncthreads();
    uint4 *obs4 = reinterpret_cast<uint4 *>(p.obs);
    const float scale = (float)(1 / 255.0);
    bool any_skip = false;                                                  // (uniform: a scalar branch)
    if (skip_term) for (int i = 0; i < ne; ++i) any_skip |= (s_env[i] >> 27 & 1u) != 0;
    if (CTX1 && ES == 1 && !any_skip && nc == SPAN) {
        // the usual workgroup -- a whole span of back-to-back uint8 frames, nobody skipped -- under ONE scalar branch: PER LDS reads
        // and PER stores per lane, no per-chunk tests (round 4)
        u32x4 *dst = reinterpret_cast<u32x4 *>(obs4 + ((size_t)e0 * cpf + cr)) + tid;
#pragma unroll
        for (int kk = 0; kk < PER; ++kk) {
            const uint4 val = s_out4[GB / 16 + kk * BS + tid];
            u32x4 nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, dst + kk * BS);
        }
        return;
    }
#pragma unroll
    for (int kk = 0; kk < PER; ++kk) {
        const int c = kk * BS + tid;
        if (c >= nc) break;
        uint4 val;
        if (ES == 4) {
            const uint32_t b = s_out[GB / 4 + c];
            val = make_uint4(__float_as_uint((float)(b & 255u) * scale), __float_as_uint((float)((b >> 8) & 255u) * scale),
                             __float_as_uint((float)((b >> 16) & 255u) * scale), __float_as_uint((float)(b >> 24) * scale));
        } else {
            val = s_out4[GB / 16 + c];
        }
        if (CTX1 && !any_skip) {
            u32x4 nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs4 + ((size_t)e0 * cpf + cr) + c));   // frames are back to back
            continue;
        }
        const unsigned cq = cr + (unsigned)c, le = cq / (unsigned)cpf, cc = cq - le * cpf;
        const uint32_t we = s_env[le];
        if (skip_term && (we >> 27 & 1u)) continue;
        uint4 *frame0 = obs4 + ((size_t)e0 + le) * p.context * cpf;
        if (CTX1) {
            u32x4 nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(frame0 + cc));
        } else {
            xw_store_chunk(frame0, (int)cc, cpf, p.context, flag_all >= 0 ? flag_all : (int)((we >> 28) & 3u), val);
        }
    }
}

template <int CH, int R, bool CTX1, int ES, int PER>
__global__ __launch_bounds__(EGO_BS) void xw_ego_gather_kernel(XwParams p, int skip_term, int publish) {
    typedef EgoSpanGeom<CH, R, ES, PER> G;
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 7, p.sig_epoch);      // the listed frames are out
    // (chunk indices fit 32 bits: the launcher checks)
    const unsigned n_chunks = (unsigned)p.n * G::cpf, c_lo = blockIdx.x * G::SPAN;
    const unsigned e0 = c_lo / G::cpf;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#ifdef XWB_EGO_PROF
        atomicAdd(&g_ego_prof[9], (unsigned long long)*p.ego_miss_count); atomicAdd(&g_ego_prof[10], 1ull);     // (tools/lab/ego_stats.py)
#endif
        *p.ego_miss_count = 0;                                               // the kernels before this one consumed the list
    }
    EGO_GATHER_LDS(G, R, lds);
    ego_gather_span<CH, R, CTX1, ES, PER>(p, lds, e0, c_lo - e0 * G::cpf, (int)(n_chunks - c_lo < (unsigned)G::SPAN ? n_chunks - c_lo : G::SPAN), skip_term, -1);
}

// the frames of the listed envs, from what the front kernels left of them (terminal frames: p.list_flag = 1)
template <int CH, int R, bool CTX1, int ES>
__global__ __launch_bounds__(EGO_BS) void xw_ego_gather_list_kernel(XwParams p, const int32_t *count_now, int publish) {
    typedef EgoSpanGeom<CH, R, ES, 2> G;
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 6, p.sig_epoch);      // the evaluation kernel is through
    const int cnt = *count_now, part = blockIdx.x % G::SPE;
    EGO_GATHER_LDS(G, R, lds);
    for (int item = blockIdx.x / G::SPE; item < cnt; item += gridDim.x / G::SPE) {
        const int e = p.done_list[item], cr = part * G::SPAN;
        __syncthreads();
        ego_gather_span<CH, R, CTX1, ES, 2>(p, lds, (unsigned)e, (unsigned)cr, G::cpf - cr < G::SPAN ? G::cpf - cr : G::SPAN, 0, p.list_flag);
        // (as the list render of the other path: the first frame of a new episode consumes fresh[] and, where the reset left
        // that to the render, the done code)
        if (part == 0 && threadIdx.x == 0 && p.list_flag == 2) { p.fresh[e] = 0; if (p.auto_reset == 2) p.done[e] = 0; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.ego_miss_count = 0;          // (the list this path's cells kernel filled is consumed)
}


// The warped 64x64 image of every goal of the listed envs (XItem::get_item_image, xitem.cpp:46-60): cv::warpAffine with
// the goal's inverse matrix, INTER_LINEAR, BORDER_CONSTANT white.  A goal keeps its pose for the whole episode, so this
// runs once per reset (~0.4 % of the envs per step) and the render reads goal pixels like any other icon.
// Four workgroups per goal, four pixels per lane with all sixteen icon reads in flight together: beside a machine-filling
// render this kernel is as slow as its chain of dependent reads (16 pixels one after the other: 108 us measured).
template <bool LIST>
__device__ __forceinline__ void warp_goals_body(const XwParams &p, const uint32_t *atlas4, const int32_t *count_now, int bid, int nblocks) {
    constexpr int PARTS = 4, PPL = 4096 / PARTS / 256;
    const int G = p.num_goals, D = p.max_dim;
    const int n_items = (LIST ? *count_now : p.n) * G * PARTS;
    for (int item = bid; item < n_items; item += nblocks) {
        const int part = item % PARTS, ig = item / PARTS, ei = ig / G, slot = ig - ei * G;
        const int e = LIST ? p.done_list[ei] : ei;
        const int cell = p.goal_cells[(size_t)e * XW_MAX_GOALS + slot];
        uint32_t *out = p.goal_img + ((size_t)e * G + slot) * 4096;
        // new poses: whatever the render cached of this env's goal cells is stale
        if (slot == 0 && part == 0 && p.ego_cache_valid)
            for (int q = threadIdx.x; q < (int)p.ego_cache_words; q += 256) p.ego_cache_valid[(size_t)e * p.ego_cache_words + q] = 0;
        if (cell == 0xff) continue;
        const int icon = (int)(p.grid[(size_t)e * D * D + cell] & CELL_ICON_MASK) - 1;
        if (icon < 0) continue;
        const double *M = p.goal_warp + ((size_t)e * XW_MAX_GOALS + slot) * 6;
        const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5];
        const uint32_t *img = atlas4 + (uint32_t)icon * 4096u;
        int fxs[PPL], fys[PPL];
        bool inside[PPL];
        uint32_t t[PPL][4];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int q = part * (4096 / PARTS) + j * 256 + threadIdx.x;
            const int px = q & 63, py = q >> 6;
            const int X0 = __double2int_rn((m1 * py + m2) * 1024) + 16, Y0 = __double2int_rn((m4 * py + m5) * 1024) + 16;
            const int X = (X0 + __double2int_rn(m0 * px * 1024)) >> 5, Y = (Y0 + __double2int_rn(m3 * px * 1024)) >> 5;
            const int ix = X >> 5, iy = Y >> 5;
            fxs[j] = X & 31; fys[j] = Y & 31;
            inside[j] = !(ix >= 64 || ix + 1 < 0 || iy >= 64 || iy + 1 < 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int qx = ix + (k & 1), qy = iy + (k >> 1);
                const bool in = (unsigned)qx < 64u && (unsigned)qy < 64u;
                const uint32_t v = img[in ? qy * 64 + qx : 0];           // (no branch around the read)
                t[j][k] = in ? v : 0xffffffu;
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int fx = fxs[j], fy = fys[j];
            uint32_t res = 0xffffffu;
            if (inside[j]) {
                int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
                if (w[0] == 32768) { w[0] = 32767; w[3] = 1; }     // BilinearTab_i: saturated entry and its compensation
                res = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    int acc = 1 << 14;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc += (int)((t[j][k] >> (8 * ch)) & 255u) * w[k];
                    res |= (uint32_t)(acc >> 15) << (8 * ch);
                }
            }
            out[part * (4096 / PARTS) + j * 256 + threadIdx.x] = res;
        }
    }
}

template <bool LIST>
__global__ __launch_bounds__(256) void xw_warp_goals_kernel(XwParams p, const uint32_t *atlas4, const int32_t *count_now) {
    warp_goals_body<LIST>(p, atlas4, count_now, (int)blockIdx.x, (int)gridDim.x);
}

// xwb_reset_done on the span path: the first two things the new episodes' first frames need -- the goal images of the reset envs
// (read by the evaluation kernel that follows) and their cell tables (which only need the new grids) -- in ONE launch, side by
// side: as two kernels in the reset's queue they ran one after the other, each as slow as its chain of dependent reads beside
// the whole-batch gather (33 + 37 us), and made that queue longer than the gather it runs beside.  Blocks [0, nb_cells): cell
// tables of the listed envs (p: the list's own source words / goal-cell list); the rest: goal images.
template <int R>
__global__ __launch_bounds__(256) void xw_ego_list_front_kernel(XwParams p, const uint8_t *map, const uint32_t *atlas4, const int32_t *count_now, int nb_cells) {
    extern __shared__ uint4 smem4[];
    if ((int)blockIdx.x < nb_cells) ego_cells_body<R, true, true>(p, map, 0, count_now, (int)blockIdx.x, smem4);
    else warp_goals_body<true>(p, atlas4, count_now, (int)blockIdx.x - nb_cells, (int)gridDim.x - nb_cells);
}

hipError_t launch_xw_warp_goals(const XwParams &p, bool list, hipStream_t s) {
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    if (list) hipLaunchKernelGGL((xw_warp_goals_kernel<true>), dim3(4096), dim3(256), 0, s, p, a4, (const int32_t *)p.done_count);
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
hipError_t xw_ego_tables(int r, int max_dim, int out_dim, EgoTap **dev_out, int *fast_out, int *cell_edge_out, int *span_out) {
    int cell_edge = 1;
    // span path: the frame as r x r squares of U = O / r pixels.  cmap: [heading][fy * r + fx] -> view cell; then, per heading,
    // [r] which border row (its place in the layout's list) frame row fy * U is, 0xff: none, and the same for columns.
    // Possible when the only rows / columns that straddle two cells are first rows / columns of a square.
    const int U = out_dim / r;
    bool span = (r == 3 || r == 5 || r == 7) && out_dim == r * (84 / r) && U % 4 == 0;
    std::vector<uint8_t> cmap((size_t)((8 * r * r + 8 * r + 15) & ~15), 0xff);     // ... then the inverse of cmap: [heading][view cell] -> square
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
        if (span) {
            for (int axis = 0; axis < 2; ++axis) {
                int nb = 0;
                for (int o = 0; o < O; ++o) {
                    if (cell_of[axis][o] < 0) {
                        if (o % U != 0 || o == 0) span = false;
                        else cmap[(size_t)4 * r * r + (size_t)(axis * 4 + dir) * r + o / U] = (uint8_t)nb;
                        nb++;
                    } else if (cell_of[axis][o] != cell_of[axis][(o / U) * U + U / 2]) {
                        span = false;
                    }
                }
            }
            for (int fy = 0; fy < r && span; ++fy)
                for (int fx = 0; fx < r; ++fx) {
                    const int rc = cell_of[0][fy * U + U / 2], cc = cell_of[1][fx * U + U / 2];
                    cmap[(size_t)dir * r * r + fy * r + fx] = (uint8_t)(row_is_y ? rc * r + cc : cc * r + rc);
                }
        }
        hd[2] = (uint16_t)cw;
        if (cw > cell_edge) cell_edge = cw;
    }
    const size_t tap_bytes = all.size() * sizeof(EgoTap), lay_bytes = lay.size() * sizeof(uint16_t);
    uint8_t *d = nullptr;
    for (int i = 0; i < 4 * r * r; ++i) {
        if (cmap[i] == 0xff || cmap[i] >= r * r) { span = false; continue; }     // (a permutation per heading, or no span path)
        cmap[(size_t)4 * r * r + 8 * r + (size_t)(i / (r * r)) * r * r + cmap[i]] = (uint8_t)(i % (r * r));
    }
    // the composed taps of an output row / column (ego_compose_taps: the two intermediate indices' taps and the output tap), so
    // that a kernel whose workgroups live for one chain of dependent reads gets them in ONE read instead of two
    std::vector<EgoTap> comp((size_t)6 * O);
    for (int i = 0; i < O; ++i) {
        comp[(size_t)3 * i + 0] = v1[v2[i].s0]; comp[(size_t)3 * i + 1] = v1[v2[i].s1]; comp[(size_t)3 * i + 2] = v2[i];
        comp[(size_t)3 * (O + i) + 0] = h1[h2[i].s0]; comp[(size_t)3 * (O + i) + 1] = h1[h2[i].s1]; comp[(size_t)3 * (O + i) + 2] = h2[i];
    }
    hipError_t err = hipMalloc(&d, tap_bytes + lay_bytes + cmap.size() + comp.size() * sizeof(EgoTap));
    if (err != hipSuccess) return err;
    err = hipMemcpy(d, all.data(), tap_bytes, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + tap_bytes, lay.data(), lay_bytes, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + tap_bytes + lay_bytes, cmap.data(), cmap.size(), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(d + tap_bytes + lay_bytes + cmap.size(), comp.data(), comp.size() * sizeof(EgoTap), hipMemcpyHostToDevice);
    *dev_out = reinterpret_cast<EgoTap *>(d);
    *fast_out = fast ? 1 : 0;
    if (span_out) *span_out = fast && span ? 1 : 0;
    if (cell_edge_out) *cell_edge_out = cell_edge;
    return err;
}

namespace {
struct EgoTables { const EgoTap *h1, *v1, *h2, *v2; const uint16_t *lut; const uint8_t *map; const EgoTap *comp; };   // comp: [2][O][3] rows, columns
EgoTables ego_tables_of(const XwParams &p) {
    const int P = 64 * p.max_dim, O = p.out_dim;
    EgoTables t;
    t.h1 = reinterpret_cast<const EgoTap *>(p.ego_taps); t.v1 = t.h1 + P; t.h2 = t.v1 + P; t.v2 = t.h2 + O;
    t.lut = reinterpret_cast<const uint16_t *>(t.v2 + O);
    t.map = reinterpret_cast<const uint8_t *>(t.lut + (size_t)4 * ego_layout_words((O + 3) & ~3, p.visible_radius));
    const int r = p.visible_radius;
    t.comp = reinterpret_cast<const EgoTap *>(t.map + (size_t)((8 * r * r + 8 * r + 15) & ~15));
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

namespace {
template <int CH, int R>
hipError_t ego_span_render_list(const XwParams &p0, const EgoTables &t, hipStream_t s, int parts);
// mode 0: every env; 2: every env the last step did not finish (a reset runs beside this: their state is in flux);
// 4: a step's frames -- every env, the finished ones first and from the list (p.list_flag says how their context moves),
//    ev_cells recorded once nothing reads the grids and agents any more (a reset's map generator may start), ev_front once
//    nothing reads the goal images either (they may be redrawn), ev_list once the listed frames are out
template <int CH, int R>
hipError_t ego_span_render(const XwParams &p, const EgoTables &t, int mode, hipStream_t s, hipEvent_t ev_front, hipEvent_t ev_list, hipEvent_t ev_cells) {
    constexpr int U = 84 / R, FB = CH * (R * U) * (R * U);
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    const size_t cells = (size_t)p.max_dim * p.max_dim;
    const int skip_front = mode == 2, skip_gather = mode != 0;
    // mode 4 without events: the hand-overs to the reset's queue are epochs, published by the kernel that FOLLOWS the producer
    const int publish = mode == 4 && !ev_front && p.sig_epoch != 0;
    const size_t cells_lds = 64 * cells * 3 + ((p.n_icons + 15) & ~15) + ((p.n_icons + 2 + 15) & ~15);
    hipLaunchKernelGGL((xw_ego_cells_kernel<R, false>), dim3((p.n + 63) / 64), dim3(64 * EgoCellsWaves<false>::NW), cells_lds, s, p, t.map, skip_front, nullptr, mode == 2 && p.sig_epoch != 0);
    if (ev_cells) { const hipError_t e = hipEventRecord(ev_cells, s); if (e != hipSuccess) return e; }
    const int nb_miss = p.dbg_ego_miss_blocks ? p.dbg_ego_miss_blocks : 4096;    // (a multiple of 4: up to four workgroups per goal cell)
    hipLaunchKernelGGL((xw_ego_eval_kernel<CH, R>), dim3(nb_miss), dim3(256), 0, s, p, a4, t.lut, t.map, publish, t.comp);
    if (ev_front) { const hipError_t e = hipEventRecord(ev_front, s); if (e != hipSuccess) return e; }
    const int es = p.obs_f32 ? 4 : 1;
    const unsigned long long n_chunks = (unsigned long long)p.n * (FB / (16 / es));
    const int32_t *cnt = (const int32_t *)p.done_count;
    const unsigned list_blocks = (unsigned)(p.n < 2048 ? p.n : 2048);
    // A/B switches (xwb_config.debug_ego_per / debug_ego_pad): 16-byte chunks per lane (2 | 4 | 8), bytes of LDS a workgroup asks for on top of its own.
    // Default padding: 13 workgroups per CU instead of 16 -- the kernels of a reset_done on the other queue (map generator,
    // goal images, list render: 256-thread groups, up to 31 KB of LDS) otherwise never find room beside this one and run
    // after it (0.292 -> 0.271 ms per step on the C4-sized batch).
    const int per = p.dbg_ego_per ? p.dbg_ego_per : 4;
    const int pad_env = p.dbg_ego_pad - 1;
#define EGO_PAD(ESV, PERV) (pad_env >= 0 ? pad_env : (163840 / 13 - EgoSpanGeom<CH, R, ESV, PERV>::LDS > 0 ? 163840 / 13 - EgoSpanGeom<CH, R, ESV, PERV>::LDS : 0))
#define EGO_GATHER_BIG(CTXV, ESV, PERV) hipLaunchKernelGGL((xw_ego_gather_kernel<CH, R, CTXV, ESV, PERV>), dim3((unsigned)((n_chunks + EGO_BS * PERV - 1) / (EGO_BS * PERV))), dim3(EGO_BS), EGO_PAD(ESV, PERV), s, p, skip_gather, publish)
#define EGO_GATHER(CTXV, ESV) do { \
        if (mode == 4) { \
            hipLaunchKernelGGL((xw_ego_gather_list_kernel<CH, R, CTXV, ESV>), dim3(list_blocks * EgoSpanGeom<CH, R, ESV, 2>::SPE), dim3(EGO_BS), 0, s, p, cnt, publish); \
            if (ev_list) { const hipError_t e = hipEventRecord(ev_list, s); if (e != hipSuccess) return e; } \
        } \
        if (per == 2) EGO_GATHER_BIG(CTXV, ESV, 2); else if (per == 8) EGO_GATHER_BIG(CTXV, ESV, 8); else EGO_GATHER_BIG(CTXV, ESV, 4); \
    } while (0)
    if (p.context == 1) { if (es == 4) EGO_GATHER(true, 4); else EGO_GATHER(true, 1); }
    else { if (es == 4) EGO_GATHER(false, 4); else EGO_GATHER(false, 1); }
#undef EGO_GATHER_BIG
#undef EGO_PAD
#undef EGO_GATHER
    return hipGetLastError();
}
}  // namespace

size_t xw_ego_square_tab_bytes(const XwParams &p) {
    const int r = p.visible_radius, U = 84 / r, UP = 4 * ((U / 4 + 3) & ~3);
    return (size_t)4 * p.ego_ncls * p.ego_ncls * p.ego_ncls * p.channels * r * r * U * UP;
}

// bytes of one cache entry on the span path (EgoEntry): a square in EgoSq's layout, all channels, and the lines next to it
size_t xw_ego_square_entry_bytes(const XwParams &p) {
    const int r = p.visible_radius, U = 84 / r, UP = 4 * ((U / 4 + 3) & ~3);
    return (size_t)((p.channels * U * UP + 2 * p.channels * U + 4 + 15) & ~15);
}

size_t xw_ego_xtab_bytes(const XwParams &p) {
    const size_t nc = (size_t)p.ego_ncls;
    return 4 * nc * nc * nc * nc * p.visible_radius * p.visible_radius * sizeof(uint32_t);
}

// the span path's tables: ego_tab3 (squares), ego_xtab (crossing pixels), ego_clsimg -- once per batch
hipError_t launch_xw_ego_build_squares(const XwParams &p, hipStream_t s) {
    const EgoTables t = ego_tables_of(p);
    const int r = p.visible_radius, nc = p.ego_ncls;
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    const unsigned blocks = (unsigned)(4 * nc * nc * nc * r * r), xblocks = (unsigned)(4 * nc * nc * nc * nc);
    uint8_t *tab3 = const_cast<uint8_t *>(p.ego_tab3);
    uint32_t *xtab = const_cast<uint32_t *>(p.ego_xtab);
    hipLaunchKernelGGL(xw_ego_build_clsimg_kernel, dim3(1), dim3(64), 0, s, p, a4, const_cast<uint2 *>(p.ego_clsimg));
#define EGO_SQ(CHV, RV) do { \
        static_assert(EgoEntry<CHV, RV>::BYTES == ((CHV * (84 / RV) * EgoSq<RV>::UP + 2 * CHV * (84 / RV) + 4 + 15) & ~15), "xw_ego_square_entry_bytes"); \
        hipLaunchKernelGGL((xw_ego_build_squares_kernel<CHV, RV>), dim3(blocks), dim3(256), 0, s, p, a4, t.h1, t.v1, t.h2, t.v2, t.map, tab3); \
        hipLaunchKernelGGL((xw_ego_build_xtab_kernel<CHV, RV>), dim3(xblocks), dim3(64), 0, s, p, a4, t.comp, t.lut, t.map, xtab); \
    } while (0)
    if (p.channels == 3) { if (r == 3) EGO_SQ(3, 3); else if (r == 5) EGO_SQ(3, 5); else EGO_SQ(3, 7); }
    else { if (r == 3) EGO_SQ(1, 3); else if (r == 5) EGO_SQ(1, 5); else EGO_SQ(1, 7); }
#undef EGO_SQ
    return hipGetLastError();
}

bool xw_ego_span(const XwParams &p) {
    // (the gather counts 16-byte chunks in 32 bits)
    return p.visible_radius && p.ego_span && p.ego_cellinfo && (unsigned long long)p.n * p.channels * p.out_dim * p.out_dim < (1ull << 32);
}

namespace {
// the frames of the done list's envs on the span path (new episodes: on the reset's queue, beside the whole-batch gather): the
// same three stages over the list, with their own source words and goal-cell list (XwParams::ego_cellsrc_list, ...) -- the
// whole-batch gather may still be reading the batch's
// `parts`: 1 = the two front kernels (they write the list's own source words, the goal-cell cache and the border rows: nothing
// the caller reads), 2 = the gather (the frames), 3 = both
template <int CH, int R>
hipError_t ego_span_render_list(const XwParams &p0, const EgoTables &t, hipStream_t s, int parts) {
    XwParams p = p0;
    p.ego_cellsrc = p0.ego_cellsrc_list; p.ego_miss = p0.ego_miss_list; p.ego_miss_count = p0.ego_miss_count_list;
    const uint32_t *a4 = reinterpret_cast<const uint32_t *>(p.atlas64);
    const size_t cells = (size_t)p.max_dim * p.max_dim;
    const int32_t *cnt = (const int32_t *)p.done_count;
    constexpr int EPW = EgoCellsGeom<true>::EPW;
    const size_t cells_lds = EPW * cells * 3 + ((p.n_icons + 15) & ~15) + ((p.n_icons + 2 + 15) & ~15);
    const int n_cap = p.n < 16384 ? p.n : 16384;               // (workgroups beyond the list leave at once)
    if (parts & 1) {
        // (parts & 4: the goal images of these envs are still to be redrawn -- launch_xw_reset with defer_warp -- in the same launch)
        const int nb_cells = (p.n + EPW - 1) / EPW;
        if (parts & 4) hipLaunchKernelGGL((xw_ego_list_front_kernel<R>), dim3(nb_cells + 4096), dim3(256), cells_lds, s, p, t.map, a4, cnt, nb_cells);
        else hipLaunchKernelGGL((xw_ego_cells_kernel<R, true>), dim3(nb_cells), dim3(256), cells_lds, s, p, t.map, 0, cnt, 0);
        hipLaunchKernelGGL((xw_ego_eval_kernel<CH, R>), dim3(1024), dim3(256), 0, s, p, a4, t.lut, t.map, 0, t.comp);
    }
    if (!(parts & 2)) return hipGetLastError();
    const int es = p.obs_f32 ? 4 : 1;
    const unsigned list_blocks = (unsigned)(n_cap < 2048 ? n_cap : 2048);
#define EGO_LIST(CTXV, ESV) hipLaunchKernelGGL((xw_ego_gather_list_kernel<CH, R, CTXV, ESV>), dim3(list_blocks * EgoSpanGeom<CH, R, ESV, 2>::SPE), dim3(EGO_BS), 0, s, p, cnt, 0)
    if (p.context == 1) { if (es == 4) EGO_LIST(true, 4); else EGO_LIST(true, 1); }
    else { if (es == 4) EGO_LIST(false, 4); else EGO_LIST(false, 1); }
#undef EGO_LIST
    return hipGetLastError();
}
}  // namespace

hipError_t launch_xw_render_ego(const XwParams &p, int indexed, hipStream_t s, hipEvent_t ev_front, hipEvent_t ev_list, hipEvent_t ev_cells) {
    const EgoTables t = ego_tables_of(p);
    const int r = p.visible_radius, O = p.out_dim, O4 = (O + 3) & ~3, D = p.max_dim;
    const int CH = p.channels;
    if (indexed != 1 && indexed < 5 && xw_ego_span(p)) {
        const int m = indexed;
        if (CH == 3) return r == 3 ? ego_span_render<3, 3>(p, t, m, s, ev_front, ev_list, ev_cells) : (r == 5 ? ego_span_render<3, 5>(p, t, m, s, ev_front, ev_list, ev_cells) : ego_span_render<3, 7>(p, t, m, s, ev_front, ev_list, ev_cells));
        return r == 3 ? ego_span_render<1, 3>(p, t, m, s, ev_front, ev_list, ev_cells) : (r == 5 ? ego_span_render<1, 5>(p, t, m, s, ev_front, ev_list, ev_cells) : ego_span_render<1, 7>(p, t, m, s, ev_front, ev_list, ev_cells));
    }
    // (5 / 6: the front kernels / the gather of the list render alone -- xwb_reset_done runs them on two queues; 7: as 5, with
    // the goal images of the listed envs redrawn in the same launch as their cell tables; 8: as 1, with that launch)
    if ((indexed == 1 || (indexed >= 5 && indexed <= 8)) && xw_ego_span(p) && p.ego_cellsrc_list) {
        const int parts = indexed == 5 ? 1 : (indexed == 6 ? 2 : (indexed == 7 ? 5 : (indexed == 8 ? 7 : 3)));
        if (CH == 3) return r == 3 ? ego_span_render_list<3, 3>(p, t, s, parts) : (r == 5 ? ego_span_render_list<3, 5>(p, t, s, parts) : ego_span_render_list<3, 7>(p, t, s, parts));
        return r == 3 ? ego_span_render_list<1, 3>(p, t, s, parts) : (r == 5 ? ego_span_render_list<1, 5>(p, t, s, parts) : ego_span_render_list<1, 7>(p, t, s, parts));
    }
    if (indexed >= 4) return hipErrorInvalidValue;             // (only the span path draws a step's terminal frames itself)
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
extern "C" __attribute__((visibility("default"))) int xwb_debug_ego_prof(unsigned long long *out) {
    unsigned long long z[12] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(xwb::g_ego_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(xwb::g_ego_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int xwb_debug_ego_prof2(unsigned long long *out) {
    unsigned long long z[12] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(xwb::g_ego_prof2), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(xwb::g_ego_prof2), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
