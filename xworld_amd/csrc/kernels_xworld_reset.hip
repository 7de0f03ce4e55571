// kernels_xworld_reset.hip -- XWorld2D reset path (map generation + teacher idle stage) for gfx950.
//
// Replaces, per environment of the compacted "done" list (or of the whole batch):
//   XWorld::reset (xworld/xworld.cpp:109-151), XWorldEnv.reset / __instantiate_entities / __padding_walls /
//   cpp_get_entities (maps/xworld_env.py:95-101,376-384,412-493), XWorldNav._configure (maps/XWorldNav.py:16-67),
//   XWorldWalls._configure (maps/XWorldWalls.py:14-36), spanning_tree_maze_generator (python/maze2d.py:74-114),
//   XWorld3DNavTarget.idle (xworld3d/tasks/XWorld3DNavTarget.py:28-43) with _reachable / bfs
//   (xworld3d_task.py:328-342, maze2d.py:43-71), XWorldSimulator::reset_game (xworld_simulator.cpp:143-157).
//
// One lane generates one map (decision order "xwb-mapgen-v1").  Only ~0.35 % of the envs
// finish per step, so this kernel is latency-bound: every per-cell set (maze walls, free cells, flooded
// cells) is a bit mask held in registers (NW x 64 bits for D*D cells), "k-th free cell in row-major order"
// is a rank-select on the mask, the flood fill is shift-and-mask on whole rows, and the only indexed
// storage -- the DFS stack, the shuffled wall list and a few per-goal words -- lives in LDS laid out
// [index][lane] so the 64 lanes of the wavefront never share a bank row entry.  Cells are written to the
// env's grid row in HBM with fire-and-forget stores.
// one out-of-line copy of the Philox block function: this kernel runs on a couple of wavefronts whose
// instruction fetches miss all the way to L2 while render_all saturates the memory system
#define XWB_PHILOX_ATTR __noinline__
#include "xwb_common.h"
#include "xw_device.h"

namespace xwb {

template <int NW>
struct Mask {
    uint64_t w[NW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = 0;
    }
    __device__ __forceinline__ bool test(int b) const {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) v = ((b >> 6) == i) ? w[i] : v;
        return (v >> (b & 63)) & 1ull;
    }
    __device__ __forceinline__ void set(int b) {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] |= ((b >> 6) == i) ? (1ull << (b & 63)) : 0ull;
    }
    __device__ __forceinline__ void reset(int b) {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] &= ((b >> 6) == i) ? ~(1ull << (b & 63)) : ~0ull;
    }
    __device__ __forceinline__ bool any() const {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) v |= w[i];
        return v != 0;
    }
    __device__ __forceinline__ bool equals(const Mask &o) const {
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) v |= w[i] ^ o.w[i];
        return v == 0;
    }
    // index of the k-th (0-based) set bit in ascending bit order
    __device__ __forceinline__ int select(int k) const {
        int base = 0;
        uint64_t word = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = __popcll(w[i]);
            if (!found) {
                if (k < c) { word = w[i]; base = i * 64; found = true; }
                else k -= c;
            }
        }
        int pos = 0;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const int c = __popcll(word & ((1ull << s) - 1ull));
            if (k >= c) { k -= c; word >>= s; pos += s; }
        }
        return base + pos;
    }
    __device__ __forceinline__ Mask shl(int s) const {      // 1 <= s <= 63
        Mask r;
#pragma unroll
        for (int i = NW - 1; i >= 0; --i) r.w[i] = (w[i] << s) | (i > 0 ? (w[i - 1] >> (64 - s)) : 0ull);
        return r;
    }
    __device__ __forceinline__ Mask shr(int s) const {
        Mask r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = (w[i] >> s) | (i + 1 < NW ? (w[i + 1] << (64 - s)) : 0ull);
        return r;
    }
    __device__ __forceinline__ Mask operator&(const Mask &o) const {
        Mask r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = w[i] & o.w[i];
        return r;
    }
    __device__ __forceinline__ Mask operator|(const Mask &o) const {
        Mask r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = w[i] | o.w[i];
        return r;
    }
    __device__ __forceinline__ Mask andnot(const Mask &o) const {      // this & ~o
        Mask r;
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = w[i] & ~o.w[i];
        return r;
    }
};

// the four neighbours of every cell of m (cells = y*D + x), clipped to the D x D board
template <int NW>
__device__ __forceinline__ Mask<NW> neighbours(const Mask<NW> &m, int D, const Mask<NW> &col0, const Mask<NW> &colN,
                                               const Mask<NW> &valid) {
    Mask<NW> r = m.andnot(colN).shl(1) | m.andnot(col0).shr(1) | m.shl(D) | m.shr(D);
    return r & valid;
}

struct IconTables {
    const int16_t *first[3];
    const int16_t *variants;
    __device__ __forceinline__ int nv(int type, int name) const { return first[type][name + 1] - first[type][name]; }
    __device__ __forceinline__ int icon(int type, int name, int k) const { return variants[first[type][name] + k]; }
};

// per-lane indexed storage in LDS: element i of lane l at [i * 64 + l]
struct LaneLds {
    uint32_t *stack;     // [64]  DFS frames: node | perm << 8 | next << 16
    uint8_t *blk;        // [D*D] shuffled '#' cells
    uint16_t *gname;     // [XW_MAX_GOALS]
    uint8_t *gcell;      // [XW_MAX_GOALS]
    uint16_t *ov_idx;    // [XW_MAX_GOALS]
    uint16_t *ov_val;    // [XW_MAX_GOALS]
    int lane;
    __device__ __forceinline__ int at(int i) const { return i * 64 + lane; }
};

// maze2d.spanning_tree_maze_generator: bit c set = '#'.  Randomised DFS over the n x n node lattice with an
// explicit stack; each node shuffles [(-1,0),(1,0),(0,1),(0,-1)] by Fisher-Yates i = 3..1, j = below(i+1).
template <int NW>
__device__ __forceinline__ Mask<NW> xw_maze(Stream &s, int D, const LaneLds &L) {
    int X = D;
    const bool pad = (X % 2) == 0;
    if (pad) X -= 1;
    const int n = (X + 1) / 2;
    Mask<NW> mz;
    mz.clear();
    for (int y = 0; y < X; ++y)
        for (int x = 0; x < X; ++x)
            if (!(x % 2 == 0 && y % 2 == 0)) mz.set(y * D + x);
    // The k-th *visited* node consumes draws 3k..3k+2 whatever the DFS path is, so all n*n shuffles are drawn
    // up front in a loop every lane runs in lock step (the DFS below is divergent in time across lanes and
    // must stay cheap per iteration).  perm table: L.blk is free until the '#' list is built.
    for (int k = 0; k < n * n; ++k) {
        int m0 = 0, m1 = 1, m2 = 2, m3 = 3;
        {   // i = 3
            const int j = (int)s.below(4u);
            const int vj = j == 0 ? m0 : (j == 1 ? m1 : (j == 2 ? m2 : m3));
            const int vi = m3;
            if (j == 0) m0 = vi; else if (j == 1) m1 = vi; else if (j == 2) m2 = vi;
            m3 = vj;
        }
        {   // i = 2
            const int j = (int)s.below(3u);
            const int vj = j == 0 ? m0 : (j == 1 ? m1 : m2);
            const int vi = m2;
            if (j == 0) m0 = vi; else if (j == 1) m1 = vi;
            m2 = vj;
        }
        {   // i = 1
            const int j = (int)s.below(2u);
            const int vj = j == 0 ? m0 : m1;
            const int vi = m1;
            if (j == 0) m0 = vi;
            m1 = vj;
        }
        L.blk[L.at(k)] = (uint8_t)(m0 | (m1 << 2) | (m2 << 4) | (m3 << 6));
    }
    uint64_t visited = 0;
    int sp = 1, n_visited = 0;
    L.stack[L.at(0)] = 0u | (0xffu << 16);
    while (sp > 0) {
        const int top = sp - 1;
        uint32_t f = L.stack[L.at(top)];
        const int node = f & 0xff;
        int perm = (f >> 8) & 0xff, next = (f >> 16) & 0xff;
        const int cx = node % n, cy = node / n;
        if (next == 0xff) {
            visited |= 1ull << node;
            perm = L.blk[L.at(n_visited++)];
            next = 0;
        }
        if (next >= 4) { sp--; continue; }
        const int m = (perm >> (2 * next)) & 3;
        next += 1;
        L.stack[L.at(top)] = (uint32_t)node | ((uint32_t)perm << 8) | ((uint32_t)next << 16);
        const int dx = m == 0 ? -1 : (m == 1 ? 1 : 0);
        const int dy = m == 2 ? 1 : (m == 3 ? -1 : 0);
        const int nx = cx + dx, ny = cy + dy;
        if (nx >= 0 && nx < n && ny >= 0 && ny < n && !((visited >> (ny * n + nx)) & 1ull)) {
            mz.reset((cy + ny) * D + (cx + nx));                 // open the wall between the two nodes
            L.stack[L.at(sp)] = (uint32_t)(ny * n + nx) | (0xffu << 16);
            sp++;
        }
    }
    if (pad) {
        for (int i = 0; i < X; ++i) if (i % 2) mz.set(X * D + i);
        for (int i = 0; i < D; ++i) if (i % 2) mz.set(i * D + X);
    }
    return mz;
}

template <int NW, int KIND>
__device__ void xw_reset_env(const XwParams &p, const IconTables &T, const LaneLds &L, int e, bool keep_done) {
    const int MD = p.max_dim, D = p.dim, off = (MD - D) / 2;
    const uint32_t ep = p.episode[e] + 1;
    p.episode[e] = ep;
    Stream s;
    s.init(p.seed, p.env_gid0 + (uint32_t)e, ep, 0);

    // board masks
    Mask<NW> valid, col0, colN;
    valid.clear(); col0.clear(); colN.clear();
    for (int y = 0; y < D; ++y) { col0.set(y * D); colN.set(y * D + D - 1); }
    for (int c = 0; c < D * D; ++c) valid.set(c);

    // grid row: brick padding outside the actual dims, empty inside; entity cells are overwritten below
    // (same lane, program order).  cpp_get_entities shifts by the padding offset, __padding_walls adds bricks.
    const uint16_t brick = (uint16_t)(T.icon(1, 0, 0) + 1);      // self.items["block"]["brick"][0]
    uint16_t *g = p.grid + (size_t)e * MD * MD;
    for (int y = 0; y < MD; ++y)
        for (int x = 0; x < MD; ++x) {
            const int lx = x - off, ly = y - off;
            g[y * MD + x] = (lx >= 0 && ly >= 0 && lx < D && ly < D) ? (uint16_t)0 : brick;
        }
    auto put = [&](int c, int icon) { g[(c / D + off) * MD + (c % D + off)] = (uint16_t)(icon + 1); };

    const int ng = p.num_goals;
    Mask<NW> avail, occupied;
    occupied.clear();
    int na, agent_cell;

    if constexpr (KIND == 0) {
        // ---- XWorldNav: distinct goal names (shuffle + pop), maze, shuffled '#' list, placement ----
        const int M = p.n_names[0];
        int n_ov = 0;
        for (int i = 0; i < ng; ++i) {
            const int j = (int)s.below((uint32_t)(M - i));
            int vj = j, vl = M - 1 - i, at_j = -1;
            for (int k = 0; k < n_ov; ++k) {
                const int idx = L.ov_idx[L.at(k)];
                if (idx == j) { vj = L.ov_val[L.at(k)]; at_j = k; }
                if (idx == M - 1 - i) vl = L.ov_val[L.at(k)];
            }
            L.gname[L.at(i)] = (uint16_t)vj;
            if (at_j >= 0) L.ov_val[L.at(at_j)] = (uint16_t)vl;               // names[j] = names[M-1-i]
            else { L.ov_idx[L.at(n_ov)] = (uint16_t)j; L.ov_val[L.at(n_ov)] = (uint16_t)vl; n_ov++; }
        }
        const Mask<NW> mz = xw_maze<NW>(s, D, L);
        int nb = 0;
#pragma unroll
        for (int wi = 0; wi < NW; ++wi) {                     // '#' cells in row-major order
            uint64_t v = mz.w[wi];
            while (v) {
                const int b = __ffsll((long long)v) - 1;
                L.blk[L.at(nb++)] = (uint8_t)(wi * 64 + b);
                v &= v - 1;
            }
        }
        avail = valid.andnot(mz);
        na = D * D - nb;
        for (int i = nb - 1; i >= 1; --i) {                   // random.shuffle(blocks)
            const int j = (int)s.below((uint32_t)(i + 1));
            const uint8_t a = L.blk[L.at(i)], b = L.blk[L.at(j)];
            L.blk[L.at(i)] = b; L.blk[L.at(j)] = a;
        }
        for (int i = 0; i < ng; ++i) {
            const int c = avail.select((int)s.below((uint32_t)na));
            avail.reset(c); na--;
            const int nm = L.gname[L.at(i)];
            const int v = (int)s.below((uint32_t)T.nv(0, nm));
            put(c, T.icon(0, nm, v));
            occupied.set(c);
            L.gcell[L.at(i)] = (uint8_t)c;
        }
        for (int i = 0; i < p.num_blocks; ++i) {
            const int c = L.blk[L.at(--nb)];                   // blocks.pop()
            const int nm = (int)s.below((uint32_t)p.n_names[1]);
            const int v = (int)s.below((uint32_t)T.nv(1, nm));
            put(c, T.icon(1, nm, v));
            occupied.set(c);
        }
        {
            const int c = avail.select((int)s.below((uint32_t)na));
            avail.reset(c); na--;
            const int nm = (int)s.below((uint32_t)p.n_names[2]);
            const int v = (int)s.below((uint32_t)T.nv(2, nm));
            put(c, T.icon(2, nm, v));
            agent_cell = c;
        }
    } else {
        // ---- XWorldWalls: one full brick row, a partial brick column, then agent, goals, blocks ----
        avail = valid;
        int nb = 0;
        int n_blocks = p.num_blocks;
        const int row = (int)s.below((uint32_t)D);
        const int first = n_blocks < D ? n_blocks : D;
        for (int i = 0; i < first; ++i) L.blk[L.at(nb++)] = (uint8_t)(row * D + i);
        n_blocks -= first;
        const int column = (int)s.below((uint32_t)D);
        const int lim = n_blocks < D - 1 ? n_blocks : D - 1;
        for (int i = 0, j = 0; j < lim; ++i) if (i != row) { L.blk[L.at(nb++)] = (uint8_t)(i * D + column); j++; }
        for (int i = 0; i < nb; ++i) avail.reset(L.blk[L.at(i)]);
        na = D * D - nb;
        {   // agent
            const int c = avail.select((int)s.below((uint32_t)na));
            avail.reset(c); na--;
            const int nm = (int)s.below((uint32_t)p.n_names[2]);
            const int v = (int)s.below((uint32_t)T.nv(2, nm));
            put(c, T.icon(2, nm, v));
            agent_cell = c;
        }
        for (int i = 0; i < ng; ++i) {
            const int c = avail.select((int)s.below((uint32_t)na));
            avail.reset(c); na--;
            const int nm = (int)s.below((uint32_t)p.n_names[0]);
            const int v = (int)s.below((uint32_t)T.nv(0, nm));
            put(c, T.icon(0, nm, v));
            occupied.set(c);
            L.gcell[L.at(i)] = (uint8_t)c;
            L.gname[L.at(i)] = (uint16_t)nm;
        }
        for (int i = 0; i < nb; ++i) {
            const int c = L.blk[L.at(i)];
            const int nm = (int)s.below((uint32_t)p.n_names[1]);
            const int v = (int)s.below((uint32_t)T.nv(1, nm));
            put(c, T.icon(1, nm, v));
            occupied.set(c);
        }
    }

    // ---- XWorld3DNavTarget.idle: goals reachable from the agent with blocks and the other goals as obstacles.
    // Flood the empty cells from the agent by whole-board shifts; a goal is reachable iff one of its
    // 4-neighbours is flooded (a path's interior holds neither blocks nor goals).
    Mask<NW> free_cells = valid.andnot(occupied);          // agent cell included: it is the seed
    Mask<NW> reach;
    reach.clear();
    reach.set(agent_cell);
    for (int it = 0; it < D * D; ++it) {
        const Mask<NW> grown = reach | (neighbours<NW>(reach, D, col0, colN, valid) & free_cells);
        if (grown.equals(reach)) break;
        reach = grown;
    }
    int nc = 0;
    uint32_t cand_bits = 0;                                 // goal i is a candidate
    for (int i = 0; i < ng; ++i) {
        Mask<NW> gm;
        gm.clear();
        gm.set(L.gcell[L.at(i)]);
        if ((neighbours<NW>(gm, D, col0, colN, valid) & reach).any()) { cand_bits |= 1u << i; nc++; }
    }
    int target = -1;                                        // reference asserts nc > 0 ("map too crowded?")
    if (nc > 0) {
        int k = (int)s.below((uint32_t)nc);                 // random.choice(targets)
        int pick = 0;
        for (int i = 0; i < ng; ++i)
            if ((cand_bits >> i) & 1u) { if (k == 0) { pick = i; break; } k--; }
        target = L.gname[L.at(pick)];
    }

    p.agent_xy[e] = (agent_cell % D + off) | ((agent_cell / D + off) << 16);
    p.task_state[e] = pack_task(target, STAGE_NAV, EV_NONE);
    p.task_steps[e] = 0;
    p.num_steps[e] = 0;
    p.fresh[e] = 2;                                       // render: init_screen (zero the older context frames)
    if (!keep_done) p.done[e] = (uint8_t)done_code(p, 0, EV_NONE);
}

template <int NW, int KIND>
__global__ __launch_bounds__(64) void xw_reset_kernel(XwParams p, int mode, int keep_done, const int32_t *count_now) {
    extern __shared__ uint32_t lds32[];
    // a handful of latency-bound wavefronts that run beside render_all's 16 waves per CU
    __builtin_amdgcn_s_setprio(3);
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int total = mode == MODE_RESET_ALL ? p.n : *count_now;
    if (blockIdx.x * 64 >= total) return;                              // whole wavefront idle
    LaneLds L;
    L.lane = threadIdx.x;
    L.stack = lds32;                                                   // 64 x 64 x 4 B
    L.gname = reinterpret_cast<uint16_t *>(lds32 + 64 * 64);           // 16 x 64 x 2 B
    L.ov_idx = L.gname + XW_MAX_GOALS * 64;
    L.ov_val = L.ov_idx + XW_MAX_GOALS * 64;
    L.gcell = reinterpret_cast<uint8_t *>(L.ov_val + XW_MAX_GOALS * 64);   // 16 x 64 B
    L.blk = L.gcell + XW_MAX_GOALS * 64;                               // D*D x 64 B
    // name -> icon-variant tables staged in LDS once per wavefront: every lookup afterwards is an LDS read
    // instead of a dependent chain of global loads queued behind render_all's write stream
    int16_t *t_first = reinterpret_cast<int16_t *>(L.blk + p.dim * p.dim * 64);
    int16_t *t_var = t_first + ((p.name_first_len + 1) & ~1);
    for (int k = threadIdx.x; k < p.name_first_len; k += 64) t_first[k] = p.name_first[k];
    for (int k = threadIdx.x; k < p.name_variants_len; k += 64) t_var[k] = p.name_variants[k];
    __syncthreads();
    if (i >= total) return;
    const int e = mode == MODE_RESET_ALL ? i : p.done_list[i];
    IconTables T;
    T.first[0] = t_first + p.name_first_off[0];
    T.first[1] = t_first + p.name_first_off[1];
    T.first[2] = t_first + p.name_first_off[2];
    T.variants = t_var;
    xw_reset_env<NW, KIND>(p, T, L, e, keep_done != 0);
}

template <int NW>
static void launch_reset_nw(const XwParams &p, int mode, dim3 grid, size_t lds, hipStream_t s) {
    const int32_t *cnt = p.done_count;
    if (p.map_kind == 0) hipLaunchKernelGGL((xw_reset_kernel<NW, 0>), grid, dim3(64), lds, s, p, mode, p.auto_reset, cnt);
    else hipLaunchKernelGGL((xw_reset_kernel<NW, 1>), grid, dim3(64), lds, s, p, mode, p.auto_reset, cnt);
}

hipError_t launch_xw_reset(const XwParams &p, int mode, hipStream_t s) {
    dim3 grid((p.n + 63) / 64);
    const int cells = p.dim * p.dim;
    const size_t lds = 64 * 64 * 4 + 3 * XW_MAX_GOALS * 64 * 2 + XW_MAX_GOALS * 64 + (size_t)cells * 64 +
                       2 * (size_t)(p.name_first_len + 2 + p.name_variants_len);
    if (lds > 65536) return hipErrorInvalidValue;
    if (cells <= 64) launch_reset_nw<1>(p, mode, grid, lds, s);
    else if (cells <= 128) launch_reset_nw<2>(p, mode, grid, lds, s);
    else launch_reset_nw<4>(p, mode, grid, lds, s);
    return hipGetLastError();
}

}  // namespace xwb
