// tools/render_lab.hip -- A/B laboratory for the XWorld2D render kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/render_lab.hip -o tools/render_lab   (built here, run on the GPU box)
// Synthetic C4 batch (32 768 envs, 7x7, 3 channels, 348-entry table); prints the average time of each variant.
#include "../xworld_amd/csrc/kernels_xworld.hip"

#include <cstdio>
#include <vector>

using namespace xwb;

namespace xwb {
// device properties for the v1 launch plan
static int g_num_cus = 0;
static size_t g_max_lds = 0;

hipError_t xw_render_prepare(int device) {
    hipDeviceProp_t prop;
    hipError_t err = hipGetDeviceProperties(&prop, device);
    if (err != hipSuccess) return err;
    g_num_cus = prop.multiProcessorCount;
    g_max_lds = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    return hipSuccess;
}

// ===== render_all v1 (the product's kernel until the one-shot kernel replaced it): persistent LDS-table workgroups =====
// all envs: persistent 1024-thread workgroups (one per CU: the table fills the LDS), tile table resident in
// LDS, env tiles staged in LDS.  Four consecutive dwords of a frame touch at most two cells -- the cell of
// dword 0 and the cell of dword 3 (cells change every 3 dwords; a row or channel wrap coincides with a cell
// change) -- so a chunk needs two cell-code reads, not four; two chunks are in flight per lane so that the
// second chunk's LDS reads overlap the first one's.  tools/render_lab.hip holds the A/B history: this shape
// is ~13 % faster than one code read per dword and beats the position-major / segment-major variants.
template <int DIM_T, int CH>
__device__ __forceinline__ uint4 xw_expand_chunk2(const uint32_t *atlas, const uint16_t *g, int cc, int dim_rt) {
    const int D = DIM_T ? DIM_T : dim_rt;
    const int RD = XW_TILE_DW * D, RH = XW_TILE * D;
    const int d0 = cc * 4;
    int ch = d0 / (RH * RD);
    const int rem = d0 - ch * (RH * RD);
    int y = rem / RD;
    int dx = rem - y * RD;
    int cidx[4], aoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cy = y / XW_TILE, py = y - cy * XW_TILE;
        const int cx = dx / XW_TILE_DW, kk = dx - cx * XW_TILE_DW;
        cidx[k] = cy * D + cx;
        aoff[k] = ch * 36 + py * 3 + kk;
        dx += 1;
        if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
    }
    const uint32_t ca = g[cidx[0]], cb = g[cidx[3]];     // staged codes: the target bit is already stripped
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t code = cidx[k] == cidx[0] ? ca : cb;
        out[k] = atlas[code * (CH * 36) + aoff[k]];                 // tile 0 = empty cell (white)
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

template <int DIM_T, int CH, bool CTX1>
__global__ __launch_bounds__(1024) void xw_render_all_v1_kernel(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    const int D = DIM_T ? DIM_T : p.max_dim;
    const int cells = D * D;
    uint8_t *s_fresh = reinterpret_cast<uint8_t *>(s_grid + (tile_envs + 1) * cells);
    const int tid = threadIdx.x;
    const int ctx = CTX1 ? 1 : p.context;
    const int cpf = CH * 9 * cells;                       // 16-byte chunks per frame: C*144*D*D/16
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.atlas);
        for (int i = tid; i < atlas_dw / 4; i += 1024) smem4[i] = src[i];
    }
    // Each workgroup owns one contiguous range of the batch's 16-byte chunks [g_lo, g_hi), cut at 1 KiB
    // boundaries (64 chunks = one wavefront store) and balanced to +-1 KiB for ANY workgroup count.  Ranges
    // ignore env boundaries on purpose: an env frame is 16-byte but not 128-byte aligned (7x7x3: 21 168 B),
    // and wave stores that straddle cache lines cost ~20 % of the write bandwidth (measured: 181 vs 148 us).
    const long long total_chunks = (long long)p.n * cpf;
    const long long units = (total_chunks + 63) / 64;
    const long long g_lo = units * blockIdx.x / gridDim.x * 64;
    long long g_hi = units * (blockIdx.x + 1) / gridDim.x * 64;
    if (g_hi > total_chunks) g_hi = total_chunks;
    const long long win = (long long)tile_envs * cpf;
    for (long long w0 = g_lo; w0 < g_hi; w0 += win) {
        const long long w1 = w0 + win < g_hi ? w0 + win : g_hi;
        const int e_first = (int)(w0 / cpf), e_last = (int)((w1 - 1) / cpf);
        const int ne = e_last - e_first + 1;                           // <= tile_envs + 1
        __syncthreads();
        const uint16_t *gsrc = p.grid + (size_t)e_first * cells;
        for (int i = tid; i < ne * cells; i += 1024) s_grid[i] = gsrc[i] & CELL_ICON_MASK;   // drop the target bit
        if (!CTX1 && tid < ne) s_fresh[tid] = p.fresh[e_first + tid];  // rewritten by the next step kernel
        __syncthreads();
        const unsigned base = (unsigned)(w0 - (long long)e_first * cpf);   // chunk offset of w0 inside env e_first
        const int span = (int)(w1 - w0);
        uint4 *win_obs = reinterpret_cast<uint4 *>(p.obs) + w0;             // CTX1: frames are back to back
        for (int c0 = tid; c0 < span; c0 += 2048) {
            const int c1 = c0 + 1024;
            const bool has1 = c1 < span;
            const unsigned a0 = base + (unsigned)c0, a1 = base + (unsigned)(has1 ? c1 : c0);
            const int le0 = (int)(a0 / (unsigned)cpf), cc0 = (int)(a0 - (unsigned)le0 * (unsigned)cpf);
            const int le1 = (int)(a1 / (unsigned)cpf), cc1 = (int)(a1 - (unsigned)le1 * (unsigned)cpf);
            const uint4 v0 = xw_expand_chunk2<DIM_T, CH>(s_atlas, s_grid + le0 * cells, cc0, D);
            const uint4 v1 = xw_expand_chunk2<DIM_T, CH>(s_atlas, s_grid + le1 * cells, cc1, D);
            if (CTX1) {
                u32x4 n0 = {v0.x, v0.y, v0.z, v0.w};
                __builtin_nontemporal_store(n0, reinterpret_cast<u32x4 *>(win_obs + c0));
                if (has1) {
                    u32x4 n1 = {v1.x, v1.y, v1.z, v1.w};
                    __builtin_nontemporal_store(n1, reinterpret_cast<u32x4 *>(win_obs + c1));
                }
            } else {
                uint4 *obs4 = reinterpret_cast<uint4 *>(p.obs);
                xw_store_chunk(obs4 + (size_t)(e_first + le0) * ctx * cpf, cc0, cpf, ctx, s_fresh[le0], v0);
                if (has1) xw_store_chunk(obs4 + (size_t)(e_first + le1) * ctx * cpf, cc1, cpf, ctx, s_fresh[le1], v1);
            }
        }
    }
}

// launch configuration shared by both render_all variants: how many grids fit next to the table in LDS,
// and how many persistent workgroups the chip holds (LDS-limited: 1 per CU for the colour NAV palette)
struct RenderPlan { int tile_envs, n_tiles, atlas_dw, n_blocks; size_t lds; };

template <int CH>
static hipError_t plan_render(const XwParams &p, int tile_cap, RenderPlan &r) {
    const int cells = p.max_dim * p.max_dim;
    r.atlas_dw = (p.n_icons + 1) * CH * 36;
    const size_t atlas_bytes = (size_t)r.atlas_dw * 4;
    const size_t lds_cap = g_max_lds ? g_max_lds : 65536;
    const size_t per_env = (size_t)cells * 2 + 1;          // cell codes + fresh flag
    if (atlas_bytes + per_env + 64 > lds_cap) return hipErrorInvalidValue;
    r.tile_envs = (int)((lds_cap - atlas_bytes - 64) / per_env) - 1;
    if (r.tile_envs > tile_cap) r.tile_envs = tile_cap;
    r.n_tiles = (p.n + r.tile_envs - 1) / r.tile_envs;
    r.lds = atlas_bytes + (size_t)(r.tile_envs + 1) * per_env + 16;      // a chunk window can overlap tile_envs + 1 envs
    const int cus = g_num_cus ? g_num_cus : 256;
    int per_cu = (int)(lds_cap / r.lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 1) per_cu = 1;                            // 1024-thread groups: one per CU keeps the tile split even
    // Leave one CU per XCD without a render workgroup: the reset kernel's few latency-bound wavefronts run
    // beside this kernel (side stream) and are 3.5x slower when they must share a CU with 16 render waves
    // (workgroup b is placed on XCD b % 8, so cus - 8 groups leave exactly one free CU in every XCD).
    int want = cus * per_cu;
    if (want >= 64) want -= 8;
    if (const char *ev = getenv("XWB_RENDER_BLOCKS")) { const int v = atoi(ev); if (v > 0) want = v; }
    const int n_env_groups = (p.n + 3) / 4;                // at least ~4 envs per workgroup
    r.n_blocks = n_env_groups < want ? n_env_groups : want;
    return hipSuccess;
}

template <typename K>
static hipError_t allow_big_lds(K kern, size_t lds, size_t &configured) {
    if (lds > 65536 && configured < lds) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(g_max_lds ? g_max_lds : lds));
        if (err != hipSuccess) return err;
        configured = g_max_lds ? g_max_lds : lds;
    }
    return hipSuccess;
}

template <int DIM_T, int CH>
static hipError_t render_all_v1(const XwParams &p, hipStream_t s) {
    RenderPlan r;
    hipError_t err = plan_render<CH>(p, 16, r);
    if (err != hipSuccess) return err;
    if (p.context == 1) {
        auto kern = xw_render_all_v1_kernel<DIM_T, CH, true>;
        static size_t configured = 0;
        if ((err = allow_big_lds(kern, r.lds, configured)) != hipSuccess) return err;
        hipLaunchKernelGGL(kern, dim3(r.n_blocks), dim3(1024), r.lds, s, p, r.tile_envs, r.n_tiles, r.atlas_dw);
    } else {
        auto kern = xw_render_all_v1_kernel<DIM_T, CH, false>;
        static size_t configured = 0;
        if ((err = allow_big_lds(kern, r.lds, configured)) != hipSuccess) return err;
        hipLaunchKernelGGL(kern, dim3(r.n_blocks), dim3(1024), r.lds, s, p, r.tile_envs, r.n_tiles, r.atlas_dw);
    }
    return hipGetLastError();
}


// (moved out of the product after losing the A/B: position-major render)
// ---- render_all, position-major variant (compile-time D) ------------------------------------------------
// Where a 16-byte chunk sits inside a frame (channel, pixel row, the cells its 4 dwords fall in, the dword
// offsets inside the tile) does not depend on the env.  Each lane therefore owns P fixed chunk positions,
// decodes them ONCE into 2 packed registers per position (4 cell indices, 4 in-tile dword offsets), and then
// walks the envs of the staged tile: per env and position it is 4 cell-code reads + 4 table reads from LDS,
// 4 multiply-adds and one non-temporal 16-byte store.  NT is chosen so that NT * P covers the frame's chunk
// count with < 6 % idle lanes (7x7x3: 1323 chunks = 704 lanes x 2).
template <int D, int CH>
struct RenderGeom {
    static constexpr int cells = D * D;
    static constexpr int cpf = CH * 9 * cells;                       // 16-byte chunks per frame
    static constexpr int P = (cpf + 1023) / 1024;
    static constexpr int NT = 64 * ((cpf + 64 * P - 1) / (64 * P));
};

template <int D, int CH>
__global__ __launch_bounds__((RenderGeom<D, CH>::NT)) void xw_render_all_v2_kernel(XwParams p, int tile_envs, int n_tiles,
                                                                               int atlas_dw) {
    using G = RenderGeom<D, CH>;
    constexpr int NT = G::NT, P = G::P, cells = G::cells, cpf = G::cpf;
    constexpr int RD = XW_TILE_DW * D, RH = XW_TILE * D;
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    uint8_t *s_fresh = reinterpret_cast<uint8_t *>(s_grid + tile_envs * cells);
    const int tid = threadIdx.x;
    const int ctx = p.context;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.atlas);
        for (int i = tid; i < atlas_dw / 4; i += NT) smem4[i] = src[i];
    }
    // decode this lane's chunk positions once
    uint32_t cell4[P], off4[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int pos = tid + q * NT;
        const int d0 = (pos < cpf ? pos : 0) * 4;
        int ch = d0 / (RH * RD);
        const int rem = d0 - ch * (RH * RD);
        int y = rem / RD;
        int dx = rem - y * RD;
        uint32_t c4 = 0, o4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cy = y / XW_TILE, py = y - cy * XW_TILE;
            const int cx = dx / XW_TILE_DW, kk = dx - cx * XW_TILE_DW;
            c4 |= (uint32_t)(cy * D + cx) << (8 * k);
            o4 |= (uint32_t)(ch * 36 + py * 3 + kk) << (8 * k);
            dx += 1;
            if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
        }
        cell4[q] = c4; off4[q] = o4;
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs;
        const int ne = min(tile_envs, p.n - e0);
        __syncthreads();
        const uint16_t *gsrc = p.grid + (size_t)e0 * cells;
        for (int i = tid; i < ne * cells; i += NT) s_grid[i] = gsrc[i];
        if (ctx > 1 && tid < ne) s_fresh[tid] = p.fresh[e0 + tid];
        __syncthreads();
        if (ctx > 1 && tid < ne) p.fresh[e0 + tid] = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int pos = tid + q * NT;
            if (pos >= cpf) continue;
            const uint32_t c4 = cell4[q], o4 = off4[q];
            uint4 *dst = reinterpret_cast<uint4 *>(p.obs) + (size_t)e0 * ctx * cpf;
#pragma unroll 4
            for (int le = 0; le < ne; ++le) {
                const uint16_t *g = s_grid + le * cells;
                uint4 v;
                v.x = s_atlas[(uint32_t)g[c4 & 0xff] * (CH * 36) + (o4 & 0xff)];
                v.y = s_atlas[(uint32_t)g[(c4 >> 8) & 0xff] * (CH * 36) + ((o4 >> 8) & 0xff)];
                v.z = s_atlas[(uint32_t)g[(c4 >> 16) & 0xff] * (CH * 36) + ((o4 >> 16) & 0xff)];
                v.w = s_atlas[(uint32_t)g[c4 >> 24] * (CH * 36) + (o4 >> 24)];
                xw_store_chunk(dst + (size_t)le * ctx * cpf, pos, cpf, ctx, ctx > 1 ? s_fresh[le] != 0 : false, v);
            }
        }
    }
}

}  // namespace xwb

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// ---- variant C: persistent 1024-thread workgroups holding the 150 KB LDS allocation, trivial payload ----
__global__ __launch_bounds__(1024) void lab_store_only(uint4 *obs, int cpf, int tile_envs, int n_tiles, int n, int nt) {
    extern __shared__ uint4 smem4[];
    if (threadIdx.x == 0) smem4[0] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    const uint4 v = smem4[0];
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs;
        const int ne = min(tile_envs, n - e0);
        const int total = ne * cpf;
        uint4 *dst = obs + (size_t)e0 * cpf;
        for (int c = threadIdx.x; c < total; c += 1024) {
            u32x4 nv = {v.x + c, v.y, v.z, v.w};
            if (nt) __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(dst + c));
            else dst[c] = make_uint4(v.x + c, v.y, v.z, v.w);
        }
    }
}

// ---- variant B: full index math + grid reads, but the table read replaced by arithmetic ----
template <int D, int CH>
__global__ __launch_bounds__(1024) void lab_no_table(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    const int cells = D * D, tid = threadIdx.x, cpf = CH * 9 * cells;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs, ne = min(tile_envs, p.n - e0);
        __syncthreads();
        for (int i = tid; i < ne * cells; i += 1024) s_grid[i] = p.grid[(size_t)e0 * cells + i];
        __syncthreads();
        for (int c = tid; c < ne * cpf; c += 1024) {
            const int le = c / cpf, cc = c - le * cpf;
            const int RD = 3 * D, RH = 12 * D;
            int d0 = cc * 4, ch = d0 / (RH * RD), rem = d0 - ch * (RH * RD), y = rem / RD, dx = rem - y * RD;
            uint32_t out[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
                const uint32_t code = s_grid[le * cells + cy * D + cx];
                out[k] = code * (CH * 36) + ch * 36 + py * 3 + kk;
                dx += 1;
                if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
            }
            u32x4 nv = {out[0], out[1], out[2], out[3]};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(reinterpret_cast<uint4 *>(p.obs) + (size_t)(e0 + le) * cpf + cc));
        }
    }
}

// ---- variant E: position-major, fully unrolled over a compile-time tile of 16 envs, loads batched before stores ----
template <int D, int CH, int TILE>
__global__ __launch_bounds__((RenderGeom<D, CH>::NT)) void lab_v2_batched(XwParams p, int n_tiles, int atlas_dw) {
    using G = RenderGeom<D, CH>;
    constexpr int NT = G::NT, P = G::P, cells = G::cells, cpf = G::cpf, RD = 3 * D, RH = 12 * D;
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    const int tid = threadIdx.x;
    for (int i = tid; i < atlas_dw / 4; i += NT) smem4[i] = reinterpret_cast<const uint4 *>(p.atlas)[i];
    uint32_t cell4[P], off4[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int pos = tid + q * NT;
        const int d0 = (pos < cpf ? pos : 0) * 4;
        int ch = d0 / (RH * RD);
        const int rem = d0 - ch * (RH * RD);
        int y = rem / RD, dx = rem - y * RD;
        uint32_t c4 = 0, o4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
            c4 |= (uint32_t)(cy * D + cx) << (8 * k);
            o4 |= (uint32_t)(ch * 36 + py * 3 + kk) << (8 * k);
            dx += 1;
            if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
        }
        cell4[q] = c4; off4[q] = o4;
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * TILE;
        __syncthreads();
        for (int i = tid; i < TILE * cells; i += NT) s_grid[i] = p.grid[(size_t)e0 * cells + i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int pos = tid + q * NT;
            if (pos >= cpf) continue;
            const uint32_t c4 = cell4[q], o4 = off4[q];
            uint4 *dst = reinterpret_cast<uint4 *>(p.obs) + (size_t)e0 * cpf + pos;
#pragma unroll
            for (int h = 0; h < TILE; h += 8) {
                uint32_t code[8][4];
#pragma unroll
                for (int le = 0; le < 8; ++le) {
                    const uint16_t *g = s_grid + (h + le) * cells;
                    code[le][0] = g[c4 & 0xff]; code[le][1] = g[(c4 >> 8) & 0xff];
                    code[le][2] = g[(c4 >> 16) & 0xff]; code[le][3] = g[c4 >> 24];
                }
                uint32_t v[8][4];
#pragma unroll
                for (int le = 0; le < 8; ++le) {
                    v[le][0] = s_atlas[code[le][0] * (CH * 36) + (o4 & 0xff)];
                    v[le][1] = s_atlas[code[le][1] * (CH * 36) + ((o4 >> 8) & 0xff)];
                    v[le][2] = s_atlas[code[le][2] * (CH * 36) + ((o4 >> 16) & 0xff)];
                    v[le][3] = s_atlas[code[le][3] * (CH * 36) + (o4 >> 24)];
                }
#pragma unroll
                for (int le = 0; le < 8; ++le) {
                    u32x4 nv = {v[le][0], v[le][1], v[le][2], v[le][3]};
                    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(dst + (size_t)(h + le) * cpf));
                }
            }
        }
    }
}


// ---- variant F: segment-major.  Work item = one 12-byte tile row segment (3 dwords, contiguous in the table);
// a lane owns P fixed segment positions (channel, pixel row, cell column) and walks the envs of the tile:
// per env 1 cell-code read + ds_read2_b32 + ds_read_b32 + one 12-byte store.  Consecutive lanes write
// consecutive 12-byte segments (frame rows are D segments and rows are contiguous in the planar frame).
template <int D, int CH>
struct SegGeom {
    static constexpr int cells = D * D;
    static constexpr int spf = CH * 12 * D * D;                       // 12-byte segments per frame
    static constexpr int P = (spf + 1023) / 1024;
    static constexpr int NT = 64 * ((spf + 64 * P - 1) / (64 * P));
};
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <int D, int CH, int TILE, int BATCH, bool SKIP_EMPTY>
__global__ __launch_bounds__((SegGeom<D, CH>::NT)) void lab_seg(XwParams p, int n_tiles, int atlas_dw) {
    using G = SegGeom<D, CH>;
    constexpr int NT = G::NT, P = G::P, cells = G::cells, spf = G::spf, RH = 12 * D;
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    const int tid = threadIdx.x;
    for (int i = tid; i < atlas_dw / 4; i += NT) smem4[i] = reinterpret_cast<const uint4 *>(p.atlas)[i];
    int cellq[P], offq[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int pos = min(tid + q * NT, spf - 1);
        const int ch = pos / (RH * D), rem = pos - ch * (RH * D);
        const int y = rem / D, cx = rem - y * D;
        const int cy = y / 12, py = y - cy * 12;
        cellq[q] = cy * D + cx;
        offq[q] = ch * 36 + py * 3;
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * TILE;
        __syncthreads();
        for (int i = tid; i < TILE * cells; i += NT) s_grid[i] = p.grid[(size_t)e0 * cells + i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int pos = tid + q * NT;
            if (pos >= spf) continue;
            uint32_t *dst = reinterpret_cast<uint32_t *>(p.obs) + ((size_t)e0 * spf + pos) * 3;
#pragma unroll
            for (int h = 0; h < TILE; h += BATCH) {
                uint32_t code[BATCH];
#pragma unroll
                for (int le = 0; le < BATCH; ++le) code[le] = s_grid[(h + le) * cells + cellq[q]];
                u32x3 v[BATCH];
#pragma unroll
                for (int le = 0; le < BATCH; ++le) {
                    if (SKIP_EMPTY && code[le] == 0) {
                        v[le] = (u32x3){0xffffffffu, 0xffffffffu, 0xffffffffu};
                    } else {
                        const uint32_t *t = s_atlas + code[le] * (CH * 36) + offq[q];
                        v[le] = (u32x3){t[0], t[1], t[2]};
                    }
                }
#pragma unroll
                for (int le = 0; le < BATCH; ++le)
                    __builtin_nontemporal_store(v[le], reinterpret_cast<u32x3 *>(dst + (size_t)(h + le) * spf * 3));
            }
        }
    }
}

__global__ __launch_bounds__(256) void lab_s1(uint4 *obs, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 nv = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + i)); }
}
template <int BS>
__global__ __launch_bounds__(BS) void lab_s2(uint4 *obs, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BS + threadIdx.x; i < n; i += (size_t)gridDim.x * BS) {
        u32x4 nv = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + i));
    }
}
// persistent, each workgroup owns a contiguous span, UNR stores in flight per lane
template <int BS, int UNR>
__global__ __launch_bounds__(BS) void lab_s3(uint4 *obs, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)BS * UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const size_t j = i + (size_t)u * BS;
            if (j < hi) { u32x4 nv = {(uint32_t)j, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + j)); }
        }
    }
}

template <int K> __device__ __forceinline__ void wait_vm();
template <> __device__ __forceinline__ void wait_vm<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<1>() { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<2>() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
// linear sweep by persistent workgroups, at most K+1 stores outstanding per wavefront
template <int BS, int K>
__global__ __launch_bounds__(BS) void lab_s4(uint4 *obs, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BS + threadIdx.x; i < n; i += (size_t)gridDim.x * BS) {
        u32x4 nv = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + i));
        wait_vm<K>();
    }
}

// ---- variant H: S1-shaped render: short-lived 256-thread workgroups in dispatch order, one 16-byte chunk per
// lane, cell codes and table dwords through L1/L2 (no LDS, no barriers) ----
template <int D, int CH, int BS, int PER>
__global__ __launch_bounds__(BS) void lab_h(XwParams p, size_t n_chunks) {
    constexpr int cells = D * D, cpf = CH * 9 * cells, RD = 3 * D, RH = 12 * D;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const size_t g = ((size_t)blockIdx.x * PER + u) * BS + threadIdx.x;
        if (g >= n_chunks) return;
        const int e = (int)(g / cpf), cc = (int)(g - (size_t)e * cpf);
        const uint16_t *grid = p.grid + (size_t)e * cells;
        int d0 = cc * 4, ch = d0 / (RH * RD), rem = d0 - ch * (RH * RD), y = rem / RD, dx = rem - y * RD;
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
            const uint32_t code = grid[cy * D + cx];
            out[k] = p.atlas[code * (CH * 36) + ch * 36 + py * 3 + kk];
            dx += 1;
            if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
        }
        u32x4 nv = {out[0], out[1], out[2], out[3]};
        __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(reinterpret_cast<uint4 *>(p.obs) + g));
    }
}

// persistent workgroups pulling 16 KB blocks (TILE_CH chunks) from a global ticket counter
template <int BS>
__global__ __launch_bounds__(BS) void lab_s5(uint4 *obs, size_t n, int tile_ch, int *ticket) {
    __shared__ int s_t;
    const int n_tiles = (int)((n + tile_ch - 1) / tile_ch);
    while (true) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1);
        __syncthreads();
        const int t = s_t;
        __syncthreads();
        if (t >= n_tiles) break;
        const size_t lo = (size_t)t * tile_ch, hi = lo + tile_ch < n ? lo + tile_ch : n;
        for (size_t i = lo + threadIdx.x; i < hi; i += BS) {
            u32x4 nv = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + i));
        }
    }
}

// ---- variants of the chunk-major product kernel ----
template <int D, int CH, bool SKIP, bool TWO_CODES, int UNR>
__global__ __launch_bounds__(1024) void lab_a(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    constexpr int cells = D * D, cpf = CH * 9 * cells, RD = 3 * D, RH = 12 * D;
    const int tid = threadIdx.x;
    for (int i = tid; i < atlas_dw / 4; i += 1024) smem4[i] = reinterpret_cast<const uint4 *>(p.atlas)[i];
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs, ne = min(tile_envs, p.n - e0);
        __syncthreads();
        for (int i = tid; i < ne * cells; i += 1024) s_grid[i] = p.grid[(size_t)e0 * cells + i];
        __syncthreads();
        const int total = ne * cpf;
        for (int c0 = tid; c0 < total; c0 += 1024 * UNR) {
            uint32_t out[UNR][4];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 1024;
                if (c >= total) break;
                const int le = c / cpf, cc = c - le * cpf;
                const uint16_t *g = s_grid + le * cells;
                int d0 = cc * 4, ch = d0 / (RH * RD), rem = d0 - ch * (RH * RD), y = rem / RD, dx = rem - y * RD;
                int cidx[4], aoff[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
                    cidx[k] = cy * D + cx;
                    aoff[k] = ch * 36 + py * 3 + kk;
                    dx += 1;
                    if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
                }
                uint32_t code[4];
                if (TWO_CODES) {
                    const uint32_t ca = g[cidx[0]], cb = g[cidx[3]];
#pragma unroll
                    for (int k = 0; k < 4; ++k) code[k] = cidx[k] == cidx[0] ? ca : cb;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) code[k] = g[cidx[k]];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (SKIP) out[u][k] = code[k] ? s_atlas[code[k] * (CH * 36) + aoff[k]] : 0xffffffffu;
                    else out[u][k] = s_atlas[code[k] * (CH * 36) + aoff[k]];
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 1024;
                if (c >= total) break;
                const int le = c / cpf, cc = c - le * cpf;
                u32x4 nv = {out[u][0], out[u][1], out[u][2], out[u][3]};
                __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(reinterpret_cast<uint4 *>(p.obs) + (size_t)(e0 + le) * cpf + cc));
            }
        }
    }
}

// ---- A9: cell codes staged as overlapping pairs (code[c] | code[next(c)] << 16): one LDS read serves the (at most
// two) cells a 16-byte chunk touches ----
template <int D, int CH, int UNR>
__global__ __launch_bounds__(1024) void lab_a9(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint32_t *s_pair = s_atlas + atlas_dw;
    constexpr int cells = D * D, cpf = CH * 9 * cells, RD = 3 * D, RH = 12 * D;
    const int tid = threadIdx.x;
    for (int i = tid; i < atlas_dw / 4; i += 1024) smem4[i] = reinterpret_cast<const uint4 *>(p.atlas)[i];
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs, ne = min(tile_envs, p.n - e0);
        __syncthreads();
        for (int i = tid; i < ne * cells; i += 1024) {
            const int le = i / cells, c = i - le * cells;
            const uint16_t *g = p.grid + (size_t)(e0 + le) * cells;
            s_pair[i] = (uint32_t)g[c] | ((uint32_t)g[c + 1 == cells ? 0 : c + 1] << 16);
        }
        __syncthreads();
        const int total = ne * cpf;
        for (int c0 = tid; c0 < total; c0 += 1024 * UNR) {
            uint32_t out[UNR][4];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 1024;
                if (c >= total) break;
                const int le = c / cpf, cc = c - le * cpf;
                int d0 = cc * 4, ch = d0 / (RH * RD), rem = d0 - ch * (RH * RD), y = rem / RD, dx = rem - y * RD;
                int cidx[4], aoff[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
                    cidx[k] = cy * D + cx;
                    aoff[k] = ch * 36 + py * 3 + kk;
                    dx += 1;
                    if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
                }
                const uint32_t pr = s_pair[le * cells + cidx[0]];
                const uint32_t ca = pr & 0xffff, cb = pr >> 16;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t code = cidx[k] == cidx[0] ? ca : cb;
                    out[u][k] = s_atlas[code * (CH * 36) + aoff[k]];
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = c0 + u * 1024;
                if (c >= total) break;
                const int le = c / cpf, cc = c - le * cpf;
                u32x4 nv = {out[u][0], out[u][1], out[u][2], out[u][3]};
                __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(reinterpret_cast<uint4 *>(p.obs) + (size_t)(e0 + le) * cpf + cc));
            }
        }
    }
}

// ---- variant J: S1-shaped render.  Short-lived workgroups in dispatch order, each owning EPB whole envs.  The
// workgroup copies only the tiles its envs need (one slot per occupied cell + one white slot) from the L2-resident
// table into LDS with 16-byte loads, then expands its envs' frames chunk-major exactly like the product kernel.
template <int D, int CH, int BS, int EPB>
__global__ __launch_bounds__(BS) void lab_j(XwParams p) {
    constexpr int cells = D * D, cpf = CH * 9 * cells, TDW = CH * 36, RD = 3 * D, RH = 12 * D;
    __shared__ uint4 s_tiles4[(EPB * cells + 1) * TDW / 4];      // slot 0 = white, slot 1 + le*cells + c = cell c of env le
    __shared__ uint16_t s_code[EPB * cells];
    uint32_t *s_tiles = reinterpret_cast<uint32_t *>(s_tiles4);
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * EPB;
    const int ne = min(EPB, p.n - e0);
    for (int i = tid; i < ne * cells; i += BS) s_code[i] = p.grid[(size_t)e0 * cells + i];
    for (int i = tid; i < TDW / 4; i += BS) s_tiles4[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    __syncthreads();
    // tile copy: (cell, quad) items, 27 uint4 per tile for CH = 3
    constexpr int QPT = TDW / 4;
    const uint4 *atlas4 = reinterpret_cast<const uint4 *>(p.atlas);
    for (int i = tid; i < ne * cells * QPT; i += BS) {
        const int c = i / QPT, q = i - c * QPT;
        const uint32_t code = s_code[c];
        if (code) s_tiles4[(1 + c) * QPT + q] = atlas4[code * QPT + q];
    }
    __syncthreads();
    const int total = ne * cpf;
    uint4 *obs = reinterpret_cast<uint4 *>(p.obs) + (size_t)e0 * cpf;
    for (int c0 = tid; c0 < total; c0 += BS) {
        const int le = c0 / cpf, cc = c0 - le * cpf;
        int d0 = cc * 4, ch = d0 / (RH * RD), rem = d0 - ch * (RH * RD), y = rem / RD, dx = rem - y * RD;
        int cidx[4], aoff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cy = y / 12, py = y - cy * 12, cx = dx / 3, kk = dx - cx * 3;
            cidx[k] = le * cells + cy * D + cx;
            aoff[k] = ch * 36 + py * 3 + kk;
            dx += 1;
            if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
        }
        const uint32_t sa = s_code[cidx[0]] ? 1 + cidx[0] : 0, sb = s_code[cidx[3]] ? 1 + cidx[3] : 0;
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = s_tiles[(cidx[k] == cidx[0] ? sa : sb) * TDW + aoff[k]];
        u32x4 nv = {out[0], out[1], out[2], out[3]};
        __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + c0));
    }
}

// ---- variant K: one-shot workgroups in dispatch order (the S1 store structure).  The span's bytes are assembled
// in LDS in OUTPUT order from 12-byte tile rows gathered through L2 (one 12-byte load per tile row instead of H's four
// dword gathers per chunk), then leave as one 1 KiB-aligned wavefront store of 16-byte chunks.
template <int D, int CH, int BS, int PER, bool LOADS_FIRST = false>
__global__ __launch_bounds__(BS) void lab_k(XwParams p, size_t n_chunks) {
    constexpr int cells = D * D, FB = CH * 144 * cells, PB = 144 * cells, RB = 12 * D;
    constexpr int SPAN = BS * PER;                                  // chunks per workgroup
    __shared__ uint4 s_out4[SPAN + 2];
    __shared__ uint16_t s_code[3 * cells];
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_out4);
    const int tid = threadIdx.x;
    const size_t c_lo = (size_t)blockIdx.x * SPAN;
    const size_t c_hi = c_lo + SPAN < n_chunks ? c_lo + SPAN : n_chunks;
    const size_t b_lo = c_lo * 16, b_hi = c_hi * 16;
    const int e0 = (int)(b_lo / FB), e1 = (int)((b_hi - 1) / FB);
    const int ncode = (e1 - e0 + 1) * cells;
    for (int i = tid; i < ncode; i += BS) s_code[i] = p.grid[(size_t)e0 * cells + i] & 0x7fff;
    __syncthreads();
    const size_t u0 = b_lo / 12, u1 = (b_hi + 11) / 12;
    const int nu = (int)(u1 - u0);
    const int shift = 4 - (int)(b_lo - u0 * 12) / 4;                // dword index of unit u0 (chunk 0 sits at dword 4)
    if (LOADS_FIRST) {
        constexpr int IT = ((SPAN * 16 + 11) / 12 + 1 + BS - 1) / BS;
        uint32_t va[IT], vb[IT], vc[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = it * BS + tid;
            const size_t B = (u0 + (i < nu ? i : 0)) * 12;
            const int e = (int)(B / FB);
            const int r = (int)(B - (size_t)e * FB);
            const int ch = r / PB, r2 = r - ch * PB, y = r2 / RB, cx = (r2 - y * RB) / 12, cy = y / 12, py = y - cy * 12;
            const uint32_t code = e < p.n ? s_code[(e - e0) * cells + cy * D + cx] : 0;
            const uint32_t *src = p.atlas + code * (CH * 36) + ch * 36 + py * 3;
            va[it] = src[0]; vb[it] = src[1]; vc[it] = src[2];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = it * BS + tid;
            if (i < nu) { const int o = shift + 3 * i; s_out[o] = va[it]; s_out[o + 1] = vb[it]; s_out[o + 2] = vc[it]; }
        }
    } else {
    for (int i = tid; i < nu; i += BS) {
        const size_t B = (u0 + i) * 12;
        const int e = (int)(B / FB);
        const int r = (int)(B - (size_t)e * FB);
        const int ch = r / PB, r2 = r - ch * PB, y = r2 / RB, cx = (r2 - y * RB) / 12, cy = y / 12, py = y - cy * 12;
        const uint32_t code = e < p.n ? s_code[(e - e0) * cells + cy * D + cx] : 0;
        const uint32_t *src = p.atlas + code * (CH * 36) + ch * 36 + py * 3;
        const uint32_t a = src[0], b = src[1], c = src[2];
        const int o = shift + 3 * i;
        s_out[o] = a; s_out[o + 1] = b; s_out[o + 2] = c;
    }
    }
    __syncthreads();
    const int nc = (int)(c_hi - c_lo);
    uint4 *obs = reinterpret_cast<uint4 *>(p.obs) + c_lo;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = k * BS + tid;
        if (c < nc) {
            const uint4 v = s_out4[1 + c];
            u32x4 nv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + c));
        }
    }
}

__global__ __launch_bounds__(256) void lab_checksum(const uint32_t *obs, size_t n_dw, unsigned long long *out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dw; i += (size_t)gridDim.x * 256)
        acc += (unsigned long long)obs[i] * (2 * i + 1);
    atomicAdd(out, acc);
}

// one-shot workgroups of BS threads, PER adjacent 16-byte stores per lane (workgroup covers BS*PER*16 contiguous bytes)
template <int BS, int PER>
__global__ __launch_bounds__(BS) void lab_s6(uint4 *obs, size_t n) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const size_t i = ((size_t)blockIdx.x * PER + u) * BS + threadIdx.x;
        if (i < n) { u32x4 nv = {(uint32_t)i, 1, 2, 3}; __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs + i)); }
    }
}

int main() {
    const int N = 32768, D = 7, CH = 3, NI = 347;
    const int cells = D * D, cpf = CH * 9 * cells;
    const size_t obs_bytes = (size_t)N * cpf * 16;
    CK(xw_render_prepare(0));
    std::vector<uint16_t> grid((size_t)N * cells);
    uint32_t seed = 12345;
    for (auto &g : grid) { seed = seed * 1664525u + 1013904223u; uint32_t r = seed >> 8; uint32_t q = r % 100; g = q < 57 ? 0 : (q < 90 ? (uint16_t)2 : (uint16_t)(1 + (r / 100) % NI)); }  // 57% empty, 33% brick, 10% random icons (NAV 7x7: 28 / 16 / 5 of 49)
    std::vector<uint32_t> atlas((size_t)(NI + 1) * CH * 36);
    for (size_t i = 0; i < atlas.size(); ++i) atlas[i] = (uint32_t)(i * 2654435761u);
    XwParams p{};
    p.n = N; p.context = 1; p.max_dim = D; p.dim = D; p.channels = CH; p.n_icons = NI;
    uint16_t *d_grid; uint32_t *d_atlas; uint8_t *d_obs, *d_fresh;
    CK(hipMalloc(&d_grid, grid.size() * 2)); CK(hipMalloc(&d_atlas, atlas.size() * 4));
    CK(hipMalloc(&d_obs, obs_bytes)); CK(hipMalloc(&d_fresh, N));
    CK(hipMemcpy(d_grid, grid.data(), grid.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_atlas, atlas.data(), atlas.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_fresh, 0, N));
    p.grid = d_grid; p.atlas = d_atlas; p.obs = d_obs; p.fresh = d_fresh;
    RenderPlan r;
    CK(plan_render<CH>(p, 16, r));
    printf("plan: tile_envs %d n_tiles %d lds %zu blocks %d cpf %d\n", r.tile_envs, r.n_tiles, r.lds, r.n_blocks, cpf);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_it = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipGetLastError());
        printf("%-46s %8.1f us  %7.1f GB/s\n", name, ms * 50.0, obs_bytes / (ms * 50.0) / 1e3);
        return 0;
    };
    size_t cfg = 0;
    CK(allow_big_lds(lab_store_only, r.lds, cfg));
    time_it("hipMemsetAsync", [&] { (void)hipMemsetAsync(d_obs, 1, obs_bytes, 0); });
    for (int blocks : {256, 512}) {
        char nm[96];
        snprintf(nm, sizeof nm, "C store-only nt, %d x 1024 thr, 150KB LDS", blocks);
        time_it(nm, [&] { hipLaunchKernelGGL(lab_store_only, dim3(blocks), dim3(1024), blocks == 256 ? r.lds : 1024, 0, reinterpret_cast<uint4 *>(d_obs), cpf, 16, r.n_tiles, N, 1); });
    }
    time_it("C store-only plain, 256 x 1024", [&] { hipLaunchKernelGGL(lab_store_only, dim3(256), dim3(1024), r.lds, 0, reinterpret_cast<uint4 *>(d_obs), cpf, 16, r.n_tiles, N, 0); });
    time_it("C store-only nt, 2048 x 1024 thr, small LDS", [&] { hipLaunchKernelGGL(lab_store_only, dim3(2048), dim3(1024), 1024, 0, reinterpret_cast<uint4 *>(d_obs), cpf, 16, r.n_tiles, N, 1); });
    cfg = 0; CK(allow_big_lds(lab_no_table<7, 3>, r.lds, cfg));
    time_it("B index math + grid reads, no table", [&] { hipLaunchKernelGGL((lab_no_table<7, 3>), dim3(256), dim3(1024), r.lds, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(xw_render_all_v1_kernel<7, 3, true>, r.lds, cfg));
    time_it("A product v1 (chunk-major)", [&] { hipLaunchKernelGGL((xw_render_all_v1_kernel<7, 3, true>), dim3(256), dim3(1024), r.lds, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(xw_render_all_v2_kernel<7, 3>, r.lds, cfg));
    time_it("D v2 (position-major, unroll 4)", [&] { hipLaunchKernelGGL((xw_render_all_v2_kernel<7, 3>), dim3(256), dim3(RenderGeom<7, 3>::NT), r.lds, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_v2_batched<7, 3, 16>, r.lds, cfg));
    time_it("E v2 batched 8 envs (loads before stores)", [&] { hipLaunchKernelGGL((lab_v2_batched<7, 3, 16>), dim3(256), dim3(RenderGeom<7, 3>::NT), r.lds, 0, p, N / 16, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_seg<7, 3, 16, 8, false>, r.lds, cfg));
    time_it("F segment-major 12B, batch 8", [&] { hipLaunchKernelGGL((lab_seg<7, 3, 16, 8, false>), dim3(256), dim3(SegGeom<7, 3>::NT), r.lds, 0, p, N / 16, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_seg<7, 3, 16, 4, false>, r.lds, cfg));
    time_it("F segment-major 12B, batch 4", [&] { hipLaunchKernelGGL((lab_seg<7, 3, 16, 4, false>), dim3(256), dim3(SegGeom<7, 3>::NT), r.lds, 0, p, N / 16, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_seg<7, 3, 16, 16, false>, r.lds, cfg));
    time_it("F segment-major 12B, batch 16", [&] { hipLaunchKernelGGL((lab_seg<7, 3, 16, 16, false>), dim3(256), dim3(SegGeom<7, 3>::NT), r.lds, 0, p, N / 16, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_seg<7, 3, 16, 8, true>, r.lds, cfg));
    time_it("G segment-major 12B, batch 8, skip empty", [&] { hipLaunchKernelGGL((lab_seg<7, 3, 16, 8, true>), dim3(256), dim3(SegGeom<7, 3>::NT), r.lds, 0, p, N / 16, r.atlas_dw); });
    printf("-- tile size sweep (store-only nt | product v1) --\n");
    for (int te : {1, 2, 4, 8, 16, 32}) {
        char nm[96];
        const int ntl = (N + te - 1) / te;
        snprintf(nm, sizeof nm, "C store-only, tile %d envs", te);
        time_it(nm, [&] { hipLaunchKernelGGL(lab_store_only, dim3(256), dim3(1024), r.lds, 0, reinterpret_cast<uint4 *>(d_obs), cpf, te, ntl, N, 1); });
        if (te <= 16) {
            snprintf(nm, sizeof nm, "A product v1, tile %d envs", te);
            time_it(nm, [&] { hipLaunchKernelGGL((xw_render_all_v1_kernel<7, 3, true>), dim3(256), dim3(1024), r.lds, 0, p, te, ntl, r.atlas_dw); });
        }
    }
    printf("-- pure store structures --\n");
    const size_t nch = obs_bytes / 16;
    time_it("S1 one uint4 per thread, 256-thr blocks", [&] { hipLaunchKernelGGL(lab_s1, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S2 grid-stride, 2048 x 256", [&] { hipLaunchKernelGGL((lab_s2<256>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S2 grid-stride, 256 x 1024", [&] { hipLaunchKernelGGL((lab_s2<1024>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S2 grid-stride, 512 x 1024", [&] { hipLaunchKernelGGL((lab_s2<1024>), dim3(512), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S3 spans, 256 x 1024, 1 in flight", [&] { hipLaunchKernelGGL((lab_s3<1024, 1>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S3 spans, 256 x 1024, 4 in flight", [&] { hipLaunchKernelGGL((lab_s3<1024, 4>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S3 spans, 1024 x 256, 4 in flight", [&] { hipLaunchKernelGGL((lab_s3<256, 4>), dim3(1024), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S3 spans, 2048 x 256, 8 in flight", [&] { hipLaunchKernelGGL((lab_s3<256, 8>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    printf("-- throttled linear sweep (persistent 256 x 1024) --\n");
    time_it("S4 vmcnt(0)", [&] { hipLaunchKernelGGL((lab_s4<1024, 0>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 vmcnt(1)", [&] { hipLaunchKernelGGL((lab_s4<1024, 1>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 vmcnt(2)", [&] { hipLaunchKernelGGL((lab_s4<1024, 2>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 vmcnt(4)", [&] { hipLaunchKernelGGL((lab_s4<1024, 4>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 vmcnt(8)", [&] { hipLaunchKernelGGL((lab_s4<1024, 8>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 vmcnt(16)", [&] { hipLaunchKernelGGL((lab_s4<1024, 16>), dim3(256), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 512x1024 vmcnt(0)", [&] { hipLaunchKernelGGL((lab_s4<1024, 0>), dim3(512), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 512x1024 vmcnt(1)", [&] { hipLaunchKernelGGL((lab_s4<1024, 1>), dim3(512), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 2048x256 vmcnt(0)", [&] { hipLaunchKernelGGL((lab_s4<256, 0>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    time_it("S4 2048x256 vmcnt(1)", [&] { hipLaunchKernelGGL((lab_s4<256, 1>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); });
    printf("-- S1-shaped render (table through L1/L2) --\n");
    time_it("H 256 thr, 1 chunk/lane", [&] { hipLaunchKernelGGL((lab_h<7, 3, 256, 1>), dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, 0, p, nch); });
    time_it("H 256 thr, 2 chunks/lane", [&] { hipLaunchKernelGGL((lab_h<7, 3, 256, 2>), dim3((unsigned)((nch + 511) / 512)), dim3(256), 0, 0, p, nch); });
    time_it("H 256 thr, 4 chunks/lane", [&] { hipLaunchKernelGGL((lab_h<7, 3, 256, 4>), dim3((unsigned)((nch + 1023) / 1024)), dim3(256), 0, 0, p, nch); });
    time_it("H 512 thr, 1 chunk/lane", [&] { hipLaunchKernelGGL((lab_h<7, 3, 512, 1>), dim3((unsigned)((nch + 511) / 512)), dim3(512), 0, 0, p, nch); });
    time_it("H 1024 thr, 1 chunk/lane", [&] { hipLaunchKernelGGL((lab_h<7, 3, 1024, 1>), dim3((unsigned)((nch + 1023) / 1024)), dim3(1024), 0, 0, p, nch); });
    printf("-- dynamic ticket queue (persistent) --\n");
    int *d_ticket; CK(hipMalloc(&d_ticket, 4));
    for (int tch : {1024, 4096, 21168}) {
        for (int blocks : {256, 512}) {
            char nm[96]; snprintf(nm, sizeof nm, "S5 ticket, %d x 1024, tile %d chunks", blocks, tch);
            time_it(nm, [&] { (void)hipMemsetAsync(d_ticket, 0, 4, 0); hipLaunchKernelGGL((lab_s5<1024>), dim3(blocks), dim3(1024), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch, tch, d_ticket); });
        }
    }
    time_it("S5 ticket, 2048 x 256, tile 1024 chunks", [&] { (void)hipMemsetAsync(d_ticket, 0, 4, 0); hipLaunchKernelGGL((lab_s5<256>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch, 1024, d_ticket); });
    time_it("S5 ticket, 2048 x 256, tile 256 chunks", [&] { (void)hipMemsetAsync(d_ticket, 0, 4, 0); hipLaunchKernelGGL((lab_s5<256>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch, 256, d_ticket); });
    printf("-- chunk-major variants --\n");
#define LABA(S, T, U, NAME) { cfg = 0; CK(allow_big_lds(lab_a<7, 3, S, T, U>, r.lds, cfg)); \
    time_it(NAME, [&] { hipLaunchKernelGGL((lab_a<7, 3, S, T, U>), dim3(256), dim3(1024), r.lds, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); }); }
    LABA(false, false, 1, "A0 = product structure")
    LABA(true, false, 1, "A1 skip empty")
    LABA(false, true, 1, "A2 two code reads")
    LABA(true, true, 1, "A3 skip empty + two codes")
    LABA(false, false, 2, "A5 unroll 2")
    LABA(true, true, 2, "A6 skip + two codes + unroll 2")
    LABA(false, true, 2, "A7 two codes + unroll 2")
    LABA(false, true, 4, "A8 two codes + unroll 4")
    cfg = 0; CK(allow_big_lds(lab_a9<7, 3, 1>, r.lds + 2048, cfg));
    time_it("A9 pair codes, unroll 1", [&] { hipLaunchKernelGGL((lab_a9<7, 3, 1>), dim3(256), dim3(1024), r.lds + 2048, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_a9<7, 3, 2>, r.lds + 2048, cfg));
    time_it("A9 pair codes, unroll 2", [&] { hipLaunchKernelGGL((lab_a9<7, 3, 2>), dim3(256), dim3(1024), r.lds + 2048, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    cfg = 0; CK(allow_big_lds(lab_a9<7, 3, 3>, r.lds + 2048, cfg));
    time_it("A9 pair codes, unroll 3", [&] { hipLaunchKernelGGL((lab_a9<7, 3, 3>), dim3(256), dim3(1024), r.lds + 2048, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw); });
    printf("-- S1-shaped render with per-workgroup tile staging --\n");
    time_it("J 256 thr, 1 env / group", [&] { hipLaunchKernelGGL((lab_j<7, 3, 256, 1>), dim3(N), dim3(256), 0, 0, p); });
    time_it("J 256 thr, 2 envs / group", [&] { hipLaunchKernelGGL((lab_j<7, 3, 256, 2>), dim3(N / 2), dim3(256), 0, 0, p); });
    time_it("J 512 thr, 2 envs / group", [&] { hipLaunchKernelGGL((lab_j<7, 3, 512, 2>), dim3(N / 2), dim3(512), 0, 0, p); });
    time_it("J 512 thr, 1 env / group", [&] { hipLaunchKernelGGL((lab_j<7, 3, 512, 1>), dim3(N), dim3(512), 0, 0, p); });
    time_it("J 1024 thr, 2 envs / group", [&] { hipLaunchKernelGGL((lab_j<7, 3, 1024, 2>), dim3(N / 2), dim3(1024), 0, 0, p); });
    time_it("PRODUCT one-shot render_all (256 x 4)", [&] { (void)launch_xw_render(p, 0, 0); });
    printf("-- K: one-shot, 12-byte tile rows gathered through L2 into LDS in output order --\n");
#define LABK(BS, PER) { char nm[64]; snprintf(nm, sizeof nm, "K one-shot, %d thr, %d chunks/lane", BS, PER); \
    time_it(nm, [&] { hipLaunchKernelGGL((lab_k<7, 3, BS, PER>), dim3((unsigned)((nch + (size_t)BS * PER - 1) / ((size_t)BS * PER))), dim3(BS), 0, 0, p, nch); }); }
    LABK(256, 1) LABK(256, 2) LABK(256, 4) LABK(512, 2) LABK(512, 4) LABK(1024, 1) LABK(1024, 2) LABK(128, 4) LABK(64, 4)
    LABK(128, 2) LABK(128, 3) LABK(128, 6) LABK(128, 8) LABK(128, 16) LABK(192, 4) LABK(256, 3) LABK(256, 8) LABK(64, 8) LABK(64, 16)
#define LABK2(BS, PER) { char nm[64]; snprintf(nm, sizeof nm, "K2 loads first, %d thr, %d chunks/lane", BS, PER); \
    time_it(nm, [&] { hipLaunchKernelGGL((lab_k<7, 3, BS, PER, true>), dim3((unsigned)((nch + (size_t)BS * PER - 1) / ((size_t)BS * PER))), dim3(BS), 0, 0, p, nch); }); }
    LABK2(128, 2) LABK2(128, 4) LABK2(128, 8) LABK2(256, 2) LABK2(256, 4) LABK2(64, 4) LABK2(64, 8)
    {   // K against the product kernel: checksum of every output dword
        unsigned long long *d_ck, ck[3] = {0, 0, 0};
        CK(hipMalloc(&d_ck, 24)); CK(hipMemset(d_ck, 0, 24));
        cfg = 0; CK(allow_big_lds(xw_render_all_v1_kernel<7, 3, true>, r.lds, cfg));
        CK(hipMemset(d_obs, 0, obs_bytes));
        hipLaunchKernelGGL((xw_render_all_v1_kernel<7, 3, true>), dim3(248), dim3(1024), r.lds, 0, p, r.tile_envs, r.n_tiles, r.atlas_dw);
        hipLaunchKernelGGL(lab_checksum, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(d_obs), obs_bytes / 4, d_ck);
        CK(hipMemset(d_obs, 0, obs_bytes));
        hipLaunchKernelGGL((lab_k<7, 3, 128, 4>), dim3((unsigned)((nch + 511) / 512)), dim3(128), 0, 0, p, nch);
        hipLaunchKernelGGL(lab_checksum, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(d_obs), obs_bytes / 4, d_ck + 1);
        CK(hipMemset(d_obs, 0, obs_bytes));
        hipLaunchKernelGGL((lab_k<7, 3, 256, 2, true>), dim3((unsigned)((nch + 511) / 512)), dim3(256), 0, 0, p, nch);
        hipLaunchKernelGGL(lab_checksum, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint32_t *>(d_obs), obs_bytes / 4, d_ck + 2);
        CK(hipMemcpy(ck, d_ck, 24, hipMemcpyDeviceToHost));
        printf("checksum product %016llx  K %016llx  K2 %016llx  %s\n", ck[0], ck[1], ck[2], ck[0] == ck[1] && ck[0] == ck[2] ? "EQUAL" : "DIFFERENT");
    }
    printf("-- one-shot workgroups: size and stores per lane --\n");
#define S6(BS, PER) { char nm[64]; snprintf(nm, sizeof nm, "S6 one-shot, %d thr, %d stores/lane", BS, PER); \
    time_it(nm, [&] { hipLaunchKernelGGL((lab_s6<BS, PER>), dim3((unsigned)((nch + (size_t)BS * PER - 1) / ((size_t)BS * PER))), dim3(BS), 0, 0, reinterpret_cast<uint4 *>(d_obs), nch); }); }
    S6(64, 1) S6(128, 1) S6(256, 1) S6(512, 1) S6(1024, 1)
    S6(256, 2) S6(256, 4) S6(256, 8) S6(256, 16) S6(256, 64)
    S6(1024, 2) S6(1024, 4) S6(1024, 8) S6(1024, 21)
    return 0;
}
