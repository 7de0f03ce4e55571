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
#include "../../include/xwb_trig.h"

namespace xwb {

#ifdef XWB_RESET_PROF
__device__ unsigned long long g_reset_prof[16];   // [2k] = sum of phase k (100 MHz ticks), [2k+1] = max
#define RP_T0() unsigned long long rp_last = wall_clock64()
#define RP_T(k) do { const unsigned long long now = wall_clock64(); atomicAdd(&g_reset_prof[2 * (k)], now - rp_last); atomicMax(&g_reset_prof[2 * (k) + 1], now - rp_last); rp_last = now; } while (0)
#else
#define RP_T0()
#define RP_T(k)
#endif

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
    uint16_t *gicon;     // [XW_MAX_GOALS] (aliases ov_idx: the name overrides are dead once the goals are placed)
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

// The idle stage of an XWorld3DNav* task on one env's board (decision order "xwb-taskgen-v1"): shared by the reset
// kernel (the episode's first teach()) and by xw_idle3d_kernel (exclusive group scheduling: an idle XWorld3DNav* group picked
// in mid-episode, teacher.cpp:209-220).  The board is what the caller holds in registers / its LDS columns: cells are
// indices of the actual D x D dims, L.gcell / L.gname / L.gicon the goal slots in entity order.
template <int NW>
struct Idle3d {                                            // (references to the caller's registers: nothing is copied)
    const int D, MD, off, ng, agent_icon;
    const Mask<NW> &valid, &col0, &colN;
    Mask<NW> &occupied;                                    // in / out: blocks and goals
    int &agent_cell;                                       // in / out
    uint32_t &target_bits;                                 // out: goal slot i belongs to self.target
    int &sent_a, &sent_b;                                  // out: names bound into the teacher's grammar (G / G1, G2)
    int &between;                                          // out: NavTargetBetween's middle cell (actual-dim index)
};

// REORDER: a 2-D-native group runs beside this one -- a later idle stage enumerates the goals in env.entities order
template <int NW, bool REORDER>
__device__ __forceinline__ void xw_idle_stage_3d(const XwParams &p, int e, Stream &s, const LaneLds &L, uint16_t *g, const Idle3d<NW> &c,
                                                 int kind, int &tf) {
    const int D = c.D, MD = c.MD, off = c.off, ng = c.ng, agent_icon = c.agent_icon;
    const Mask<NW> &valid = c.valid, &col0 = c.col0, &colN = c.colN;
    Mask<NW> &occupied = c.occupied;
    int &agent_cell = c.agent_cell, &sent_a = c.sent_a, &sent_b = c.sent_b, &between = c.between;
    uint32_t &target_bits = c.target_bits;
    auto put = [&](int cell, int icon) { g[(cell / D + off) * MD + (cell % D + off)] = (uint16_t)(icon + 1); };
    if (kind == TASK_TARGET || kind == TASK_AVOID) {
        // goals reachable from the agent with blocks and the other goals as obstacles: flood the empty cells from
        // the agent by whole-board shifts; a goal is reachable iff one of its 4-neighbours is flooded
        const Mask<NW> free_cells = valid.andnot(occupied);          // agent cell included: it is the seed
        Mask<NW> reach;
        reach.clear();
        reach.set(agent_cell);
        for (int it = 0; it < D * D; ++it) {
            const Mask<NW> grown = reach | (neighbours<NW>(reach, D, col0, colN, valid) & free_cells);
            if (grown.equals(reach)) break;
            reach = grown;
        }
        int nc = 0;
        uint32_t cand_bits = 0;
        for (int i = 0; i < ng; ++i) {
            Mask<NW> gm;
            gm.clear();
            gm.set(L.gcell[L.at(i)]);
            if ((neighbours<NW>(gm, D, col0, colN, valid) & reach).any()) { cand_bits |= 1u << i; nc++; }
        }
        if (nc > 0) {                                                // else: assert targets, "map too crowded?"
            int k = (int)s.below((uint32_t)nc);                      // sel_goal = random.choice(targets)
            int pick = 0;
            for (int i = 0; i < ng; ++i)
                if ((cand_bits >> i) & 1u) { if (k == 0) { pick = i; break; } k--; }
            const int selname = L.gname[L.at(pick)];
            if (kind == TASK_TARGET) {
                tf = selname;
                sent_a = selname;
                for (int i = 0; i < ng; ++i) if (L.gname[L.at(i)] == selname) target_bits |= 1u << i;
            } else {
                int nr = 0;
                for (int i = 0; i < ng; ++i) if (L.gname[L.at(i)] != selname) nr++;
                if (nr > 0) {                                        // else: assert referents
                    int r = (int)s.below((uint32_t)nr);              // referent = random.choice(referents)
                    int refname = 0;
                    for (int i = 0; i < ng; ++i)
                        if (L.gname[L.at(i)] != selname) { if (r == 0) { refname = L.gname[L.at(i)]; break; } r--; }
                    for (int i = 0; i < ng; ++i) if (L.gname[L.at(i)] != refname) target_bits |= 1u << i;
                    sent_a = refname;
                }
            }
        }
    } else if (ng >= 2) {
        // ---- Near / Between / Direction: delete the agent and two goals, put the goals on a tile, re-place the agent
        Mask<NW> A = valid.andnot(occupied);                         // available_grids after _delete_entity(agent)
        const int d0 = (int)s.below((uint32_t)ng);                   // random.shuffle(goals); g1, g2 = goals[:2]
        const int d1 = (int)s.below((uint32_t)(ng - 1));
        const int g1 = d0, g2 = d1 < d0 ? d1 : d1 + 1;
        const int c1o = L.gcell[L.at(g1)], c2o = L.gcell[L.at(g2)];
        A.set(c1o); A.set(c2o);
        auto from_right = [&](const Mask<NW> &m) { return m.andnot(col0).shr(1); };   // bit c = m[c+1], x < D-1
        const Mask<NW> Nl = A.andnot(colN).shl(1) & valid, Nr = from_right(A), Nu = A.shl(D) & valid, Nd = A.shr(D);
        Mask<NW> M[6];
        int nm = 0;
        if (kind == TASK_NEAR) {                                     // _get_p_tiles
            const Mask<NW> C1 = Nl | Nr | Nu | Nd;
            const Mask<NW> C2 = (Nl & Nr) | (Nl & Nu) | (Nl & Nd) | (Nr & Nu) | (Nr & Nd) | (Nu & Nd);
            const Mask<NW> Hb = A & Nr, Vb = A & Nd, Db = A & from_right(A.shr(D));
            M[0] = Hb & from_right(C2); M[1] = Hb & C2;
            M[2] = Vb & C2.shr(D);      M[3] = Vb & C2;
            M[4] = Db & from_right(C1.shr(D)); M[5] = Db & C1;
            nm = 6;
        } else if (kind == TASK_BETWEEN) {                           // _get_t_tiles
            M[0] = A & Nl & Nr & (Nu | Nd);
            M[1] = A & Nu & Nd & (Nl | Nr);
            nm = 2;
        } else {                                                     // _get_l_tiles
            const Mask<NW> Tv = A & Nd & A.shr(2 * D);
            const Mask<NW> Th = A & Nr & from_right(Nr);
            M[0] = Tv; M[1] = Tv; M[2] = Th; M[3] = Th;
            nm = 4;
        }
        int nt = 0;
        for (int m = 0; m < 6; ++m) if (m < nm)
#pragma unroll
            for (int wi = 0; wi < NW; ++wi) nt += __popcll(M[m].w[wi]);
        bool ok = nt > 0;                                            // assert tiles, "map too crowded?"
        int l1 = 0, l2 = 0, al = 0, direction = 0, tgt = g1, ref = g2;
        if (ok) {
            int t0 = (int)s.below((uint32_t)nt);                     // random.shuffle(tiles); tiles[0]
            if (nt >= 2) (void)s.below((uint32_t)(nt - 1));
            // tiles are listed cell-major, the nm kinds in order inside a cell: find the cell, then the kind
            int tc = 0, tm = 0;
            {
                int lo = 0, hi = D * D;                              // smallest c with prefix(c + 1) > t0
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    int pre = 0;
                    for (int m = 0; m < 6; ++m) if (m < nm)
#pragma unroll
                        for (int wi = 0; wi < NW; ++wi) {
                            const int b = mid + 1 - wi * 64;
                            const uint64_t lowmask = b <= 0 ? 0ull : (b >= 64 ? ~0ull : ((1ull << b) - 1ull));
                            pre += __popcll(M[m].w[wi] & lowmask);
                        }
                    if (pre > t0) hi = mid; else lo = mid + 1;
                }
                tc = lo;
                int pre = 0;
                for (int m = 0; m < 6; ++m) if (m < nm)
#pragma unroll
                    for (int wi = 0; wi < NW; ++wi) {
                        const int b = tc - wi * 64;
                        const uint64_t lowmask = b <= 0 ? 0ull : (b >= 64 ? ~0ull : ((1ull << b) - 1ull));
                        pre += __popcll(M[m].w[wi] & lowmask);
                    }
                int r = t0 - pre;
                for (int m = 0; m < 6; ++m) if (m < nm && M[m].test(tc)) { if (r == 0) { tm = m; break; } r--; }
            }
            if (kind == TASK_NEAR) {
                const int other = tm < 2 ? tc + 1 : (tm < 4 ? tc + D : tc + D + 1);
                l1 = (tm & 1) ? other : tc; l2 = (tm & 1) ? tc : other;
            } else if (kind == TASK_BETWEEN) {
                l1 = tm == 0 ? tc - 1 : tc - D; l2 = tm == 0 ? tc + 1 : tc + D;
            } else {
                const int st = tm < 2 ? D : 1;
                l1 = (tm & 1) ? tc + st : tc; l2 = (tm & 1) ? tc + 2 * st : tc + st;
            }
            occupied.reset(c1o); occupied.reset(c2o);
            occupied.set(l1); occupied.set(l2);                      // _set_entity_inst(g1), (g2)
            A.reset(l1); A.reset(l2);
            int seed = l2;
            bool inclusive = false;
            if (kind == TASK_BETWEEN) {
                seed = (l1 + l2) / 2;                                // _middle_loc: same row or same column
            } else if (kind == TASK_DIRECTION) {
                Mask<NW> one;
                one.clear(); one.set(l1);
                Mask<NW> Ne = neighbours<NW>(one, D, col0, colN, valid) & A;   // empty 4-neighbours of g1 ...
                if (!Ne.any()) { one.clear(); one.set(l2); Ne = neighbours<NW>(one, D, col0, colN, valid) & A; tgt = g2; ref = g1; }
                int ne = 0;
#pragma unroll
                for (int wi = 0; wi < NW; ++wi) ne += __popcll(Ne.w[wi]);
                if (ne == 0) ok = false;                             // assert empty_grids
                else {
                    const int ec = Ne.select((int)s.below((uint32_t)ne));    // random.choice(empty_grids), row-major
                    const int tl = tgt == g1 ? l1 : l2, rl = ref == g1 ? l1 : l2;
                    // __compute_triple_direction(target, referent, e): view = e -> target, v2 = target -> referent
                    const int v1x = tl % D - ec % D, v1y = tl / D - ec / D;
                    const int v2x = rl % D - tl % D, v2y = rl / D - tl / D;
                    const int c = v1x * v2x + v1y * v2y, sn = v1y * v2x - v1x * v2y;
                    direction = c > 0 ? DIR_FRONT : (c < 0 ? DIR_BEHIND : (sn > 0 ? DIR_RIGHT : DIR_LEFT));
                    seed = ec; inclusive = true;                     // _propagate_agent([e], inclusive=True)
                }
            }
            if (ok) {
                // _propagate_agent: flood fill from the seed over cells that hold neither blocks nor goals
                const Mask<NW> open = valid.andnot(occupied);
                Mask<NW> fl;
                fl.clear(); fl.set(seed);
                for (int it = 0; it < D * D; ++it) {
                    const Mask<NW> grown = fl | (neighbours<NW>(fl, D, col0, colN, valid) & open);
                    if (grown.equals(fl)) break;
                    fl = grown;
                }
                int na = inclusive ? 0 : -1;                         // the seed itself only counts when inclusive
#pragma unroll
                for (int wi = 0; wi < NW; ++wi) na += __popcll(fl.w[wi]);
                if (na <= 0) ok = false;                             // assert new_a
                else {
                    int ka = (int)s.below((uint32_t)na);             // agent.loc, _ = random.choice(new_a)
                    al = seed;
                    if (!(inclusive && ka == 0)) {
                        // new_a is in BFS discovery order (moves left, right, up, down): replay the BFS up to entry ka
                        const int want = inclusive ? ka - 1 : ka;
                        Mask<NW> seen;
                        seen.clear(); seen.set(seed);
                        int head = 0, tail = 0, count = 0;
                        L.blk[L.at(tail++)] = (uint8_t)seed;
                        bool found = false;
                        while (head < tail && !found) {
                            const int c = L.blk[L.at(head++)];
                            const int cx = c % D, cy = c / D;
                            for (int m = 0; m < 4 && !found; ++m) {
                                const int nx = cx + (m == 0 ? -1 : (m == 1 ? 1 : 0)), ny = cy + (m == 2 ? -1 : (m == 3 ? 1 : 0));
                                if (nx < 0 || ny < 0 || nx >= D || ny >= D) continue;
                                const int nc2 = ny * D + nx;
                                if (seen.test(nc2) || occupied.test(nc2)) continue;
                                seen.set(nc2);
                                L.blk[L.at(tail++)] = (uint8_t)nc2;
                                if (count == want) { al = nc2; found = true; }
                                count++;
                            }
                        }
                    }
                }
            }
        }
        if (ok) {
            // the env changed: XWorld::reset(false).  Clear the three old cells, then write the new ones.
            auto clear_cell = [&](int c) { g[(c / D + off) * MD + (c % D + off)] = 0; };
            clear_cell(c1o); clear_cell(c2o); clear_cell(agent_cell);
            L.gcell[L.at(g1)] = (uint8_t)l1; L.gcell[L.at(g2)] = (uint8_t)l2;
            put(al, agent_icon);
            agent_cell = al;
            sent_a = L.gname[L.at(kind == TASK_DIRECTION ? ref : g1)];
            if (kind == TASK_BETWEEN) sent_b = L.gname[L.at(g2)];
            if (REORDER) {
                // (only where a later idle stage enumerates the goals: a batch with a 2-D-native group beside this one)
                // env.entities: g1 and g2 were deleted and set again, so they now follow the other goals, in that order
                // (xworld_env.py _delete_entity / _set_entity_inst).  A later idle stage that enumerates the goals -- the
                // 2-D-native group's random.choice(targets) -- sees that order, so the goal slots take it too; the
                // egocentric poses travel with their goals.
                const uint8_t c1 = L.gcell[L.at(g1)], c2 = L.gcell[L.at(g2)];
                const uint16_t i1 = L.gicon[L.at(g1)], i2 = L.gicon[L.at(g2)], n1 = L.gname[L.at(g1)], n2 = L.gname[L.at(g2)];
                double *gw = p.visible_radius ? p.goal_warp + (size_t)e * XW_MAX_GOALS * 6 : nullptr;
                double w1[6], w2[6];
                if (gw) for (int q = 0; q < 6; ++q) { w1[q] = gw[g1 * 6 + q]; w2[q] = gw[g2 * 6 + q]; }
                int k = 0;
                for (int i = 0; i < ng; ++i) {
                    if (i == g1 || i == g2) continue;
                    if (k != i) {
                        L.gcell[L.at(k)] = L.gcell[L.at(i)]; L.gicon[L.at(k)] = L.gicon[L.at(i)]; L.gname[L.at(k)] = L.gname[L.at(i)];
                        if (gw) for (int q = 0; q < 6; ++q) gw[k * 6 + q] = gw[i * 6 + q];
                    }
                    ++k;
                }
                L.gcell[L.at(k)] = c1; L.gicon[L.at(k)] = i1; L.gname[L.at(k)] = n1;
                L.gcell[L.at(k + 1)] = c2; L.gicon[L.at(k + 1)] = i2; L.gname[L.at(k + 1)] = n2;
                if (gw) for (int q = 0; q < 6; ++q) { gw[k * 6 + q] = w1[q]; gw[(k + 1) * 6 + q] = w2[q]; }
            }
            if (kind == TASK_NEAR) {
                // _get_surrounding_goals(refer=g1.loc): dist < 1.5 + 1e-3 = the 8-neighbourhood, goals AT g1.loc skipped
                for (int i = 0; i < ng; ++i) {
                    const int c = L.gcell[L.at(i)];
                    const int ddx = c % D - l1 % D, ddy = c / D - l1 / D;
                    if (c != l1 && ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1) target_bits |= 1u << i;
                }
            } else if (kind == TASK_BETWEEN) {
                between = (l1 + l2) / 2;
            } else {
                // navigation_reward: a reached goal g wins iff direction(g, referent) seen along the agent's constant
                // yaw 1.5707963 (heading +y) equals `direction` and g is within 1.0 + 1e-3 of the referent
                // (the step kernel evaluates the same test with the heading at that time -- it changes in egocentric
                // mode; the bits below are the answer for the heading at reset)
                const int rl = ref == g1 ? l1 : l2;
                const int hd = p.visible_radius ? p.agent_dir[e] : 1;
                const int hx = hd == 0 ? 1 : (hd == 2 ? -1 : 0), hy = hd == 1 ? 1 : (hd == 3 ? -1 : 0);
                for (int i = 0; i < ng; ++i) {
                    const int c = L.gcell[L.at(i)];
                    const int v2x = rl % D - c % D, v2y = rl / D - c / D;
                    if (v2x * v2x + v2y * v2y != 1) continue;         // dist == 0 -> False; dist > 1.001 -> far
                    const int cs = hx * v2x + hy * v2y, sn = hy * v2x - hx * v2y;
                    const int dir = cs > 0 ? DIR_FRONT : (cs < 0 ? DIR_BEHIND : (sn > 0 ? DIR_RIGHT : DIR_LEFT));
                    if (dir == direction) target_bits |= 1u << i;
                }
                tf = ((rl / D + off) * MD + (rl % D + off)) | (direction << 8);
            }
        }
    }
        if (kind == TASK_BETWEEN && between >= 0) tf = (between / D + off) * MD + (between % D + off);
}

// GM (task groups, compile time so that the usual one-group batch carries none of the other paths): 0 = one XWorld3DNav*
// group, 1 = one 2-D-native group, 2 = two groups
template <int NW, int KIND, int GM>
__device__ void xw_reset_env(const XwParams &p, const IconTables &T, const LaneLds &L, int e, bool keep_done,
                             const uint4 *pre_draws, uint32_t n_pre_draws, int mode_all) {
    int level_dim = p.dim, level_goals = p.num_goals, level_blocks = p.num_blocks;
    if (KIND == 0 && p.curriculum != 0) {
        // XWorldNav._configure: level -> dims, goals, blocks (XWorldNav.py:27-34)
        const int level = curriculum_configure(p, e);
        level_dim = 3 + level;
        level_goals = level < 3 ? 2 : 4;
        level_blocks = level == 5 ? 16 : 3 * level;
    }
    const int MD = p.max_dim, D = level_dim, off = (MD - D) / 2;
    RP_T0();
    // shadow (a pre-generated episode): the one after the newest this env already holds -- sh_ep counts them, so the number
    // does not depend on whether the live counter has been bumped yet by whoever installs the previous one; written into
    // shadow slot (episode & 1) of the swapped-in arrays (index ew), the live counters and flags are left alone
    const uint32_t ep = (p.shadow && mode_all == 0 ? p.sh_ep[e] : p.episode[e]) + 1;
    if (!p.shadow) p.episode[e] = ep;
    else p.sh_ep[e] = ep;
    const size_t ew = p.shadow ? (size_t)(ep & 1u) * (size_t)p.n + (size_t)e : (size_t)e;
    Stream s;
    s.init(p.seed, p.env_gid0 + (uint32_t)e, ep, 0);
    s.pre = (Stream::lds_block_ptr)pre_draws; s.npre = n_pre_draws;

    // board masks
    Mask<NW> valid, col0, colN;
    valid.clear(); col0.clear(); colN.clear();
    for (int y = 0; y < D; ++y) { col0.set(y * D); colN.set(y * D + D - 1); }
    for (int c = 0; c < D * D; ++c) valid.set(c);

    // grid row: brick padding outside the actual dims, empty inside; entity cells are overwritten below
    // (same lane, program order).  cpp_get_entities shifts by the padding offset, __padding_walls adds bricks.
    const uint16_t brick = (uint16_t)(T.icon(1, 0, 0) + 1);      // self.items["block"]["brick"][0]
    uint16_t *g = p.grid + ew * MD * MD;
    for (int y = 0; y < MD; ++y)
        for (int x = 0; x < MD; ++x) {
            const int lx = x - off, ly = y - off;
            g[y * MD + x] = (lx >= 0 && ly >= 0 && lx < D && ly < D) ? (uint16_t)0 : brick;
        }
    auto put_code = [&](int c, uint16_t code) { g[(c / D + off) * MD + (c % D + off)] = code; };
    auto put = [&](int c, int icon) { put_code(c, (uint16_t)(icon + 1)); };

    const int ng = level_goals;
    Mask<NW> avail, occupied;
    occupied.clear();
    int na, agent_cell, agent_icon;

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
        RP_T(0);
        const Mask<NW> mz = xw_maze<NW>(s, D, L);
        RP_T(1);
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
            const int ic = T.icon(0, nm, v);
            put(c, ic);
            occupied.set(c);
            L.gcell[L.at(i)] = (uint8_t)c;
            L.gicon[L.at(i)] = (uint16_t)ic;
            if (p.visible_radius) {
                // xworld_env.py:211-223: yaw ~ U[0, 4 * PI_2), scale ~ U[0.5, 1], offset ~ U[0, 1 - scale]; random.uniform
                // (a, b) = a + (b - a) * random() with random() = unit().  XItem::get_item_image (xitem.cpp:33-63) then
                // warps the icon by getRotationMatrix2D(centre, 90 - yaw * 180 / M_PI, scale) plus the translation
                // (offset + scale / 2 - 0.5) * 64; cv::warpAffine inverts that matrix: the inverse is what the
                // egocentric render needs, so it is stored.
                const double u0 = (double)s.unit(), u1 = (double)s.unit(), u2 = (double)s.unit();
                const double yaw = 0 + (1.5707963 * 4 - 0) * u0;
                const double scale = 0.5 + (1 - 0.5) * u1;
                const double offset = 0 + ((1 - scale) - 0) * u2;
                const double angle = (90 - yaw * 180 / 3.14159265358979323846) * 3.1415926535897932384626433832795 / 180;
                double sn, cs;                                  // include/xwb_trig.h: the same bits on the host's checker
                xwb_sincos(angle, &sn, &cs);
                const double alpha = cs * scale, beta = sn * scale;
                double M[6] = {alpha, beta, (1 - alpha) * 32.0 - beta * 32.0, -beta, alpha, beta * 32.0 + (1 - alpha) * 32.0};
                M[2] += (offset + scale / 2 - 0.5) * 64;
                M[5] += (offset + scale / 2 - 0.5) * 64;
                double Dt = M[0] * M[4] - M[1] * M[3];
                Dt = Dt != 0 ? 1. / Dt : 0;
                const double A11 = M[4] * Dt, A22 = M[0] * Dt;
                M[0] = A11; M[1] *= -Dt; M[3] *= -Dt; M[4] = A22;
                const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
                M[2] = b1; M[5] = b2;
                double *gw = p.goal_warp + ((size_t)e * XW_MAX_GOALS + i) * 6;
                for (int k = 0; k < 6; ++k) gw[k] = M[k];
            }
        }
        for (int i = 0; i < level_blocks; ++i) {
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
            agent_icon = T.icon(2, nm, v);
            put(c, agent_icon);
            agent_cell = c;
            // xworld_env.py:208-210: yaw = random.choice(range(-1, 3)) * PI_2 -> heading up, right, down, left
            if (p.visible_radius) p.agent_dir[e] = (uint8_t)((s.below(4u) + 3u) & 3u);
        }
    } else {
        // ---- XWorldWalls: one full brick row, a partial brick column, then agent, goals, blocks ----
        avail = valid;
        int nb = 0;
        int n_blocks = level_blocks;
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
            agent_icon = T.icon(2, nm, v);
            put(c, agent_icon);
            agent_cell = c;
        }
        for (int i = 0; i < ng; ++i) {
            const int c = avail.select((int)s.below((uint32_t)na));
            avail.reset(c); na--;
            const int nm = (int)s.below((uint32_t)p.n_names[0]);
            const int v = (int)s.below((uint32_t)T.nv(0, nm));
            const int ic = T.icon(0, nm, v);
            put(c, ic);
            occupied.set(c);
            L.gcell[L.at(i)] = (uint8_t)c;
            L.gname[L.at(i)] = (uint16_t)nm;
            L.gicon[L.at(i)] = (uint16_t)ic;
        }
        for (int i = 0; i < nb; ++i) {
            const int c = L.blk[L.at(i)];
            const int nm = (int)s.below((uint32_t)p.n_names[1]);
            const int v = (int)s.below((uint32_t)T.nv(1, nm));
            put(c, T.icon(1, nm, v));
            occupied.set(c);
        }
    }

    RP_T(2);
    // ---- teacher idle stage (TaskGroup::run_stage samples one task of the group per episode, then its idle()):
    // decision order "xwb-taskgen-v1" (DESIGN.md).  Nothing is written to the grid before the stage has
    // succeeded, so the "map too crowded?" cases (the reference asserts) simply keep the generated map.
    // One or two task groups (conf order), each: TaskGroup::run_stage draws a task, Task::reset, its idle stage.  The
    // 3-D-family stage may rearrange the map; the 2-D-family stage only reads it (its candidate tables are refreshed below
    // when it ran before a rearrangement: every later idle stage of the episode sees the final map).
    uint32_t target_bits = 0;                              // goal slot i belongs to self.target
    int sent_a = 0xffff, sent_b = 0xffff;                  // names bound into the teacher's grammar (G / G1, G2)
    int between = -1;                                      // NavTargetBetween: the middle cell (actual-dim index)

    auto idle_stage_2d = [&](int kind, bool draw, int &tf, int &st0) {
        // ---- the 2-D-native group (rule D14b).  XWorldTask._reachable: bfs with the BLOCKS as the only obstacles;
        // the agent never leaves its component and nothing else moves, so the candidate sets of every later idle
        // stage of this episode are fixed here: goal_cells + cand2d are what the step kernel's idle stage reads.
        Mask<NW> goalm;
        goalm.clear();
        for (int i = 0; i < ng; ++i) goalm.set(L.gcell[L.at(i)]);
        const Mask<NW> open2 = valid.andnot(occupied.andnot(goalm));
        Mask<NW> r2;
        r2.clear();
        r2.set(agent_cell);
        for (int it = 0; it < D * D; ++it) {
            const Mask<NW> grown = r2 | (neighbours<NW>(r2, D, col0, colN, valid) & open2);
            if (grown.equals(r2)) break;
            r2 = grown;
        }
        uint32_t cand = 0;
        for (int i = 0; i < ng; ++i) {
            const int c = L.gcell[L.at(i)];
            if (r2.test(c)) cand |= (1u << i) | (p.icon_colored[L.gicon[L.at(i)]] ? (1u << (16 + i)) : 0u);
        }
        uint8_t *gc = p.goal_cells + ew * XW_MAX_GOALS;
        for (int i = 0; i < XW_MAX_GOALS; ++i) {
            const int c = i < ng ? L.gcell[L.at(i)] : 0;
            gc[i] = i < ng ? (uint8_t)((c / D + off) * MD + (c % D + off)) : (uint8_t)0xff;
        }
        p.cand2d[ew] = cand;
        if (draw) {
            int tsteps0;
            idle_2d(kind, cand, gc, [&](uint32_t n) { return s.below(n); }, tf, st0, tsteps0);
        }
    };
    auto idle_stage_3d = [&](int kind, int &tf) {
        const Idle3d<NW> c{D, MD, off, ng, agent_icon, valid, col0, colN, occupied, agent_cell, target_bits, sent_a, sent_b, between};
        xw_idle_stage_3d<NW, GM == 2>(p, e, s, L, g, c, kind, tf);
    };
    int kindv[2] = {TASK_TARGET, TASK_TARGET}, tfv[2] = {-1, -1}, st0v[2] = {STAGE_NAV, STAGE_NAV};
    const bool first_2d = GM == 1 || (GM == 2 && p.group2d);
    if (GM == 2 && p.exclusive) {
        // Teacher::teach's exclusive branch at reset (teacher.cpp:209-220 after reset_after_game_reset): the groups are
        // re-sorted, nobody is busy, so the group that now heads the list runs its idle stage -- the other one stays idle
        // until a later teach() picks it (an XWorld3DNav* group then rearranges the map in mid-episode: xw_idle3d_kernel).
        const int pick = xw_sort_groups(p, e, ep, 0u, p.grp_order[e] & 1), other = pick ^ 1;
        const bool pick_2d = pick == 0 ? first_2d : !first_2d;
        const int tsel = pick ? sample_task<1>(p, s, e) : sample_task<0>(p, s, e);
        kindv[pick] = pick ? task_at<1>(p, tsel) : task_at<0>(p, tsel);
        kindv[other] = TASK_TARGET; tfv[other] = -1; st0v[other] = STAGE_IDLE;       // TaskGroup::reset: no busy task
        if (pick_2d) {
            idle_stage_2d(kindv[pick], true, tfv[pick], st0v[pick]);
        } else {
            idle_stage_3d(kindv[pick], tfv[pick]);
            int tf_unused, st_unused;
            idle_stage_2d(TASK2D_TARGET, false, tf_unused, st_unused);            // the 2-D group's candidate tables, from the final map
        }
        p.grp_order[e] = (uint8_t)(pick | (pick << 1));
    } else {
        if (p.exclusive && GM != 2 && p.minstd) {         // one group: the sort still draws once from the reference's engine
            uint32_t x = p.minstd[e];
            (void)xwb_minstd_rand_range_state(&x, (float)p.group_weight[0]);
            p.minstd[e] = x;
        }
        {
            const int tsel = sample_task<0>(p, s, e);
            kindv[0] = p.n_tasks > 0 ? task_at<0>(p, tsel) : TASK_TARGET;
            if (GM != 0 && first_2d) idle_stage_2d(kindv[0], true, tfv[0], st0v[0]);
            if (GM != 1 && !first_2d) idle_stage_3d(kindv[0], tfv[0]);
        }
        if (GM == 2) {
            const int tsel = sample_task<1>(p, s, e);
            kindv[1] = task_at<1>(p, tsel);
            if (!first_2d) idle_stage_2d(kindv[1], true, tfv[1], st0v[1]);
            else idle_stage_3d(kindv[1], tfv[1]);
            if (first_2d) { int tf_unused, st_unused; idle_stage_2d(kindv[0], false, tf_unused, st_unused); }
        }
    }
    const int kind = kindv[0];
    RP_T(3);
    // goal cells carry bit 15 when the goal belongs to the target set (the step kernel's whole reward rule)
    for (int i = 0; i < ng; ++i)
        put_code(L.gcell[L.at(i)], (uint16_t)((L.gicon[L.at(i)] + 1) | (((target_bits >> i) & 1u) ? 0x8000u : 0u)));
    if (GM == 0) {                                        // goal slot -> cell (the egocentric render finds a goal's pose by it)
        uint8_t *gc = p.goal_cells + ew * XW_MAX_GOALS;
        for (int i = 0; i < XW_MAX_GOALS; ++i) {
            const int c = i < ng ? L.gcell[L.at(i)] : 0;
            gc[i] = i < ng ? (uint8_t)((c / D + off) * MD + (c % D + off)) : (uint8_t)0xff;
        }
    }

    p.agent_xy[ew] = (agent_cell % D + off) | ((agent_cell / D + off) << 16);
    p.task_state[ew] = pack_task(tfv[0], st0v[0], EV_NONE, kind);
    if (GM == 2) { p.task_state2[ew] = pack_task(tfv[1], st0v[1], EV_NONE, kindv[1]); if (!p.shadow) p.task_steps2[e] = 0; }
    p.sent_names[ew] = (uint32_t)sent_a | ((uint32_t)sent_b << 16);
    if (p.shadow) return;                                 // a pre-generated episode: installed later (xw_step_kernel / the list render)
    p.task_steps[e] = 0;
    p.num_steps[e] = 0;
    p.fresh[e] = 2;                                       // render: init_screen (zero the older context frames)
    atomicAdd(p.perf + 36, 1ull);                         // games reset
    if (!keep_done) p.done[e] = (uint8_t)done_code(p, 0, EV_NONE);
    RP_T(4);
}

template <int NW, int KIND, int GM>
__global__ __launch_bounds__(64) void xw_reset_kernel(XwParams p, int mode, int keep_done, const int32_t *count_now) {
    extern __shared__ uint32_t lds32[];
    // a handful of latency-bound wavefronts that run beside render_all's 16 waves per CU
    __builtin_amdgcn_s_setprio(3);
    // Envs per wavefront: map generation is data-dependent serial code and a wavefront runs the union of its lanes' paths,
    // so the list is spread as thinly as the grid allows -- one env per wavefront for the usual fraction of a percent of
    // the batch (C4 loop: 64 / 16 / 4 / 1 envs per wavefront = 0.1346 / 0.1296 / 0.1264 / 0.1243 ms per step), more lanes
    // per wavefront when many envs finish together (fixed-length episodes), 64 when the whole batch is reset.
    const int total = mode == MODE_RESET_ALL ? p.n : *count_now;
    int per_wave = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    per_wave = per_wave < 1 ? 1 : (per_wave > 64 ? 64 : per_wave);
    if ((int)blockIdx.x * per_wave < total) {                          // else: whole wavefront idle
        LaneLds L;
        L.lane = threadIdx.x;
        L.stack = lds32;                                                   // 64 x 64 x 4 B
        L.gname = reinterpret_cast<uint16_t *>(lds32 + 64 * 64);           // 16 x 64 x 2 B
        L.ov_idx = L.gname + XW_MAX_GOALS * 64;
        L.ov_val = L.ov_idx + XW_MAX_GOALS * 64;
        L.gicon = L.ov_idx;
        L.gcell = reinterpret_cast<uint8_t *>(L.ov_val + XW_MAX_GOALS * 64);   // 16 x 64 B
        L.blk = L.gcell + XW_MAX_GOALS * 64;                               // D*D x 64 B
        // name -> icon-variant tables staged in LDS once per wavefront: every lookup afterwards is an LDS read
        // instead of a dependent chain of global loads queued behind render_all's write stream
        const int lds_dim = p.curriculum != 0 ? p.max_dim : p.dim;          // a curriculum env may be at any level
        int16_t *t_first = reinterpret_cast<int16_t *>(L.blk + lds_dim * lds_dim * 64);
        int16_t *t_var = t_first + ((p.name_first_len + 1) & ~1);
        for (int k = threadIdx.x; k < p.name_first_len; k += 64) t_first[k] = p.name_first[k];
        for (int k = threadIdx.x; k < p.name_variants_len; k += 64) t_var[k] = p.name_variants[k];
        __syncthreads();
        IconTables T;
        T.first[0] = t_first + p.name_first_off[0];
        T.first[1] = t_first + p.name_first_off[1];
        T.first[2] = t_first + p.name_first_off[2];
        T.variants = t_var;
        // One env per wavefront (the usual case): 63 lanes would idle while one walks the serial map generation, a large
        // part of whose instructions are Philox rounds.  The blocks of a counter-based stream are independent: every lane
        // computes one block of this env's reset stream up front, the serial lane then reads its draws from LDS.
        const bool solo = per_wave == 1;
        uint4 *s_pre = reinterpret_cast<uint4 *>((reinterpret_cast<uintptr_t>(t_var + p.name_variants_len) + 15) & ~(uintptr_t)15);
        // the grid is capped (a short list should not cost the dispatch of one workgroup per env of the batch): loop
        for (int base = blockIdx.x * per_wave; base < total; base += gridDim.x * per_wave) {
            const int i = base + (solo ? 0 : (int)threadIdx.x);
            const bool mine = (solo ? threadIdx.x == 0 : (int)threadIdx.x < per_wave) && i < total;
            const int e = i < total ? (mode == MODE_RESET_ALL ? i : p.done_list[i]) : 0;
            if (solo) {
                const uint32_t ep = (p.shadow && mode != MODE_RESET_ALL ? p.sh_ep[e] : p.episode[e]) + 1;   // (lane 0 bumps it below; read before that)
                __builtin_amdgcn_wave_barrier();
                s_pre[threadIdx.x] = philox4x32_10(threadIdx.x, ep, 0u, 0u, p.seed, p.env_gid0 + (uint32_t)e);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            if (mine) xw_reset_env<NW, KIND, GM>(p, T, L, e, keep_done != 0, solo ? s_pre : nullptr, solo ? 64u : 0u, mode == MODE_RESET_ALL);
            if (solo) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    }
    // (The epoch that tells the other queue's list render "every env of this launch is regenerated" is published by a
    // one-thread kernel queued behind this one.  Publishing it from here -- a release fence per writing wavefront, the last
    // one through stores the epoch -- saves that kernel's 5 us but the L2 write-backs cost the render running beside it 6 %:
    // 0.122 -> 0.127 ms per step on C4.)
}

// ---- exclusive scheduling of two task groups: an idle XWorld3DNav* group picked in mid-episode (teacher.cpp:209-220) ----
// TaskGroup::run_stage (teaching_task.cpp:204-222) for the envs the step kernel listed: draw a task, run its idle stage on the
// env's CURRENT map -- the board is rebuilt from the grid row, the goal slots from goal_cells (entity order) --, write the
// rearranged map back.  Decisions: the successive words of stream 5 | num_steps << 8 of the episode ("xwb-taskgen-v1").
// One env per wavefront (lane 0): the list holds a handful of envs per step at most.
template <int NW>
__device__ void xw_idle3d_env(const XwParams &p, const LaneLds &L, int e) {
    const int MD = p.max_dim, D = p.curriculum != 0 ? 3 + p.cur_level[e] : p.dim, off = (MD - D) / 2;
    const int G3 = p.group2d ? 1 : 0;                      // conf index of the XWorld3DNav* group
    uint16_t *g = p.grid + (size_t)e * MD * MD;
    Mask<NW> valid, col0, colN, occupied;
    valid.clear(); col0.clear(); colN.clear(); occupied.clear();
    for (int y = 0; y < D; ++y) { col0.set(y * D); colN.set(y * D + D - 1); }
    for (int c = 0; c < D * D; ++c) valid.set(c);
    const int axy = p.agent_xy[e];
    int agent_cell = ((axy >> 16) - off) * D + ((axy & 0xffff) - off);
    const int agent_icon = (int)(g[(axy >> 16) * MD + (axy & 0xffff)] & CELL_ICON_MASK) - 1;
    for (int c = 0; c < D * D; ++c)
        if (c != agent_cell && (g[(c / D + off) * MD + (c % D + off)] & CELL_ICON_MASK)) occupied.set(c);
    uint8_t *gc = p.goal_cells + (size_t)e * XW_MAX_GOALS;
    int ng = 0;
    for (int i = 0; i < XW_MAX_GOALS; ++i) {
        const int mc = gc[i];
        if (mc == 0xff) break;
        const int icon = (int)(g[mc] & CELL_ICON_MASK) - 1;
        if (icon < 0) break;                               // (cannot happen: the table lists cells that hold goals)
        L.gcell[L.at(i)] = (uint8_t)((mc / MD - off) * D + (mc % MD - off));
        L.gicon[L.at(i)] = (uint16_t)icon;
        L.gname[L.at(i)] = (uint16_t)p.icon_name[icon];
        ng++;
    }
    Stream s;
    s.init(p.seed, p.env_gid0 + (uint32_t)e, p.episode[e], 5u | ((uint32_t)p.num_steps[e] << 8));
    const int tsel = G3 ? sample_task<1>(p, s, e) : sample_task<0>(p, s, e);
    const int kind = G3 ? task_at<1>(p, tsel) : task_at<0>(p, tsel);
    uint32_t target_bits = 0;
    int sent_a = 0xffff, sent_b = 0xffff, between = -1, tf = -1;
    const Idle3d<NW> c{D, MD, off, ng, agent_icon, valid, col0, colN, occupied, agent_cell, target_bits, sent_a, sent_b, between};
    xw_idle_stage_3d<NW, true>(p, e, s, L, g, c, kind, tf);
    for (int i = 0; i < ng; ++i) {
        const int cell = L.gcell[L.at(i)];
        g[(cell / D + off) * MD + (cell % D + off)] = (uint16_t)((L.gicon[L.at(i)] + 1) | (((target_bits >> i) & 1u) ? 0x8000u : 0u));
        gc[i] = (uint8_t)((cell / D + off) * MD + (cell % D + off));
    }
    p.agent_xy[e] = (agent_cell % D + off) | ((agent_cell / D + off) << 16);
    (G3 ? p.task_state2 : p.task_state)[e] = pack_task(tf, STAGE_NAV, EV_NONE, kind);
    (G3 ? p.task_steps2 : p.task_steps)[e] = 0;
    p.sent_names[e] = (uint32_t)sent_a | ((uint32_t)sent_b << 16);
    // the step that picked the group may also have ended the game (FLAGS_max_steps): its terminal frame shows the new map
    if (!p.visible_radius && p.term_flag[e]) {
        uint16_t *t = p.term_grid + (size_t)e * MD * MD;
        for (int k = 0; k < MD * MD; ++k) t[k] = g[k];
    }
}

template <int NW>
__global__ __launch_bounds__(64) void xw_idle3d_kernel(XwParams p, const int32_t *count_now) {
    extern __shared__ uint32_t lds32[];
    const int total = *count_now;
    if ((int)blockIdx.x >= total) return;
    LaneLds L;
    L.lane = threadIdx.x;
    L.stack = lds32;                                                        // (unused: the maze generator's)
    L.gname = reinterpret_cast<uint16_t *>(lds32);
    L.ov_idx = L.gname + XW_MAX_GOALS * 64;
    L.ov_val = L.ov_idx + XW_MAX_GOALS * 64;
    L.gicon = L.ov_idx;
    L.gcell = reinterpret_cast<uint8_t *>(L.ov_val + XW_MAX_GOALS * 64);
    L.blk = L.gcell + XW_MAX_GOALS * 64;
    for (int i = blockIdx.x; i < total; i += gridDim.x)
        if (threadIdx.x == 0) xw_idle3d_env<NW>(p, L, p.idle_list[i]);
}

hipError_t launch_xw_idle3d(const XwParams &p, hipStream_t s) {
    const int lds_dim = p.curriculum != 0 ? p.max_dim : p.dim;
    const int cells = lds_dim * lds_dim;
    const size_t lds = 3 * XW_MAX_GOALS * 64 * 2 + XW_MAX_GOALS * 64 + (size_t)cells * 64;
    dim3 grid(p.n < 256 ? p.n : 256);
    const int32_t *cnt = p.idle_count;
    if (cells <= 64) hipLaunchKernelGGL((xw_idle3d_kernel<1>), grid, dim3(64), lds, s, p, cnt);
    else if (cells <= 128) hipLaunchKernelGGL((xw_idle3d_kernel<2>), grid, dim3(64), lds, s, p, cnt);
    else hipLaunchKernelGGL((xw_idle3d_kernel<4>), grid, dim3(64), lds, s, p, cnt);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    // egocentric: the goal slots of those envs were re-ordered (poses travel with their goals): their images are redrawn
    // slot by slot, which also drops the envs' cached goal cells
    if (p.visible_radius) {
        XwParams q = p;
        q.done_list = p.idle_list; q.done_count = p.idle_count;
        return launch_xw_warp_goals(q, true, s);
    }
    return hipSuccess;
}

template <int NW>
static void launch_reset_nw(const XwParams &p, int mode, dim3 grid, size_t lds, hipStream_t s) {
    const int32_t *cnt = p.done_count;
    const int gm = p.n_tasks2 > 0 ? 2 : (p.group2d ? 1 : 0);
#define XW_RESET_LAUNCH(KINDV, GMV) hipLaunchKernelGGL((xw_reset_kernel<NW, KINDV, GMV>), grid, dim3(64), lds, s, p, mode, p.auto_reset, cnt)
    if (p.map_kind == 0) { if (gm == 0) XW_RESET_LAUNCH(0, 0); else if (gm == 1) XW_RESET_LAUNCH(0, 1); else XW_RESET_LAUNCH(0, 2); }
    else { if (gm == 0) XW_RESET_LAUNCH(1, 0); else if (gm == 1) XW_RESET_LAUNCH(1, 1); else XW_RESET_LAUNCH(1, 2); }
#undef XW_RESET_LAUNCH
}

hipError_t launch_xw_reset(const XwParams &p, int mode, hipStream_t s, hipEvent_t before_warp, const uint32_t *warp_epoch_slot, uint32_t warp_epoch,
                           int defer_warp) {
    // n / 64 wavefronts in every mode (at least 256 for small batches): the whole batch = 64 envs per wavefront, a short
    // list = one env per wavefront, and the kernel fills the lanes in between as the list grows.  Whole C4 batch
    // finishing together every 8th step (tools/mass_reset.py): 0.183 ms per step with this grid, 0.323 with 2048
    // wavefronts, 0.458 with one per env -- there the machine is throughput-bound and idle lanes cost.
    const int all = (p.n + 63) / 64;
    const int want = all > 256 ? all : (p.n < 256 ? p.n : 256);
    dim3 grid(mode == MODE_RESET_ALL ? all : want);
    const int lds_dim = p.curriculum != 0 ? p.max_dim : p.dim;
    const int cells = lds_dim * lds_dim;
    const size_t lds = 64 * 64 * 4 + 3 * XW_MAX_GOALS * 64 * 2 + XW_MAX_GOALS * 64 + (size_t)cells * 64 +
                       2 * (size_t)(p.name_first_len + 2 + p.name_variants_len) + 32 + 64 * sizeof(uint4);
    if (lds > 65536) return hipErrorInvalidValue;
    if (cells <= 64) launch_reset_nw<1>(p, mode, grid, lds, s);
    else if (cells <= 128) launch_reset_nw<2>(p, mode, grid, lds, s);
    else launch_reset_nw<4>(p, mode, grid, lds, s);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    // egocentric: the goals of the reset envs got new poses; render their warped images once
    // (defer_warp: the caller redraws them itself -- launch_xw_render(p, 7, ...), beside the cell tables of the same envs)
    if (p.visible_radius && !defer_warp) {
        if (before_warp) { err = hipStreamWaitEvent(s, before_warp, 0); if (err != hipSuccess) return err; }
        if (warp_epoch_slot) { err = launch_xw_wait(warp_epoch_slot, warp_epoch, p.sync + 4, p.poison_host, s); if (err != hipSuccess) return err; }
        return launch_xw_warp_goals(p, mode != MODE_RESET_ALL, s);
    }
    return hipSuccess;
}

}  // namespace xwb

#ifdef XWB_RESET_PROF
extern "C" int xwb_debug_reset_prof(unsigned long long *out) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(xwb::g_reset_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(xwb::g_reset_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
