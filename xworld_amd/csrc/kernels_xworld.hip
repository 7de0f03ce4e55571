// kernels_xworld.hip -- XWorld2D (full observation) as lock-step data-parallel HIP for gfx950.
//
// Replaces, for a whole batch of environments per launch:
//   step    XAgent::act (xitem.cpp:89-101), XMap::move_item (xmap.cpp:76-101), XWorld::act (xworld.cpp:162-166),
//           XWorldSimulator::take_action/game_over (xworld_simulator.cpp:165-265), Teacher::teach ordering
//           (teacher.cpp:202-251, teaching_task.cpp:64-116), XWorld3DNavTarget.navigation_reward
//           (XWorld3DNavTarget.py:45-60) + _time_reward/_reach_object (xworld3d_task.py:451-482)
//   reset   XWorld::reset (xworld.cpp:109-151), XWorldEnv.reset/__instantiate_entities/__padding_walls
//           (xworld_env.py:95-101,412-493), XWorldNav._configure (XWorldNav.py:16-67), XWorldWalls._configure
//           (XWorldWalls.py:14-36), spanning_tree_maze_generator (maze2d.py:74-114), XWorld3DNavTarget.idle
//           (XWorld3DNavTarget.py:28-43) with _reachable/bfs (xworld3d_task.py:328-342, maze2d.py:43-71)
//   render  XMap::to_image (xmap.cpp:125-146), get_screen_rgb / down_sample_image
//           (xworld_simulator.cpp:287-307,508-545), make_context_screens (simulator.cpp:62-85)
//
// HBM layout (SoA, env index fastest): grid u16[N][D*D] (cell = palette icon + 1, 0 empty; one item per
// cell, which is all the nav maps ever produce), agent_xy, task_steps, task_state, num_steps, episode,
// reward, done, obs u8[N][context][C][12D][12D] planar B,G,R.
//
// The render never builds the reference's 64 px canvas: with 64 -> 12 px per cell the OpenCV bilinear
// taps of an output pixel stay inside one cell (DESIGN.md "tile table"), so the frame is a pure
// expansion  obs[n][c][12*cy+py][12*cx+px] = tile[grid[n][cy][cx]][c][py][px]  of a
// (n_icons+1) x C x 12 x 12 table that lives in LDS; HBM traffic is the output stream plus 2 B/cell.
#include "xwb_common.h"

namespace xwb {

enum : int { STAGE_IDLE = 0, STAGE_NAV = 1, STAGE_TERMINAL = 2 };
enum : int { EV_NONE = 0, EV_CORRECT = 1, EV_WRONG = 2, EV_TIMEUP = 3 };

__device__ __forceinline__ int pack_task(int target, int stage, int event) {
    return (target & 0xffff) | (stage << 16) | (event << 24);
}

__device__ __forceinline__ int done_code(const XwParams &p, int num_steps, int event) {
    // AgentSpecificSimulator::game_over = GameSimulator::game_over | XWorldSimulator::game_over
    int code = (p.max_steps > 0 && num_steps >= p.max_steps) ? MAX_STEP : ALIVE;
    if (p.task_mode == 0) {       // lang_acquisition, xworld_simulator.cpp:166-177
        if (event == EV_CORRECT) code |= SUCCESS;
        else if (event == EV_WRONG) code |= DEAD;
        else if (event == EV_TIMEUP) code |= MAX_STEP;
    }
    return code;
}

// wave-aggregated append of the lanes with `flag` set: one atomic per wavefront
__device__ __forceinline__ void wave_append(bool flag, int value, int32_t *list, int32_t *count) {
    unsigned long long m = __ballot(flag);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (flag) list[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

// ------------------------------------------------------------------- step --
__global__ __launch_bounds__(256) void xw_step_kernel(XwParams p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    int32_t *count_now = p.done_count;
    if (e == 0) *p.done_count_next = 0;        // double-buffered done counter: zero the next step's
    bool is_done = false;
    if (e < p.n) {
        int a = p.actions ? p.actions[e] : policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, p.policy_step, 4);
        p.actions_out[e] = a;
        if ((unsigned)a >= 4u) {               // CHECK_LT(action_idx, get_num_actions())
            atomicAdd(p.err_count, 1);
        } else {
            const int D = p.max_dim;
            uint16_t *g = p.grid + (size_t)e * D * D;
            const int axy = p.agent_xy[e];
            int ax = axy & 0xffff, ay = axy >> 16;
            const int steps = p.num_steps[e] + 1;          // GameSimulator::take_actions: once per call
            const uint16_t agent_code = g[ay * D + ax];
            const int ddx = a == 2 ? -1 : (a == 3 ? 1 : 0);   // MOVE_LEFT / MOVE_RIGHT
            const int ddy = a == 0 ? -1 : (a == 1 ? 1 : 0);   // MOVE_UP / MOVE_DOWN
            int hit = 0;
            bool success = false;
            for (int i = 0; i < p.act_rep; ++i) {
                const int tx = ax + ddx, ty = ay + ddy;
                success = false;
                if (tx >= 0 && ty >= 0 && tx < D && ty < D) {
                    const int code = g[ty * D + tx];
                    if (code == 0) {                        // XMap::move_item: empty cell -> move
                        g[ay * D + ax] = 0;
                        g[ty * D + tx] = agent_code;
                        ax = tx; ay = ty;
                        success = true;
                    } else {
                        hit = code;                         // contact_list -> "collision:<id>" event
                    }
                }
            }
            // Teacher::teach -> Task stage (one group, task XWorld3DNavTarget)
            const int ts = p.task_state[e];
            const int target = (int)(int16_t)(ts & 0xffff);
            int stage = (ts >> 16) & 0xff;
            int tsteps = p.task_steps[e];
            int event = EV_NONE;
            double rew = 0.0;
            if (stage == STAGE_NAV) {
                rew = -0.01;                                // time_penalty
                tsteps += 1;
                if (tsteps >= p.dim * p.dim * p.max_steps_factor) {
                    event = EV_TIMEUP;
                    stage = STAGE_TERMINAL;
                } else if (hit != 0 && a == 1 && p.icon_type[hit - 1] == 0) {
                    // _reach_object: id in collisions and |theta| < pi/4.  Full-observation entities keep
                    // yaw = 1.5707963 (heading +y), so theta = 0 only for a goal hit by MOVE_DOWN.
                    if ((int)p.icon_name[hit - 1] == target) { event = EV_CORRECT; rew += 1.0; }
                    else { event = EV_WRONG; rew += -1.0; }
                    stage = STAGE_TERMINAL;
                }
            }
            float r = 0.0f;                                 // SimulatorInterface::take_actions
            r += 0.0f;                                      // XWorldSimulator::take_action returns 0
            r = (float)((double)r + rew);                   // r += teacher_->give_reward() (double)
            const int code = done_code(p, steps, event);
            p.agent_xy[e] = ax | (ay << 16);
            p.task_state[e] = pack_task(target, stage, event);
            p.task_steps[e] = tsteps;
            p.num_steps[e] = steps;
            p.success[e] = success ? 1 : 0;
            p.reward[e] = r;
            p.done[e] = (uint8_t)code;
            is_done = code != ALIVE;
        }
    }
    wave_append(is_done, e, p.done_list, count_now);
}

hipError_t launch_xw_step(const XwParams &p, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    hipLaunchKernelGGL(xw_step_kernel, grid, block, 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- compact --
__global__ __launch_bounds__(256) void xw_compact_kernel(XwParams p, int mode, int32_t *count_now) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool flag = false;
    if (e < p.n) flag = mode == MODE_RESET_MASK ? (p.mask[e] != 0) : (p.done[e] != 0);
    wave_append(flag, e, p.done_list, count_now);
}

// ------------------------------------------------------------------ reset --
struct IconTables {
    const int16_t *first[3];
    const int16_t *variants;
    __device__ __forceinline__ int nv(int type, int name) const { return first[type][name + 1] - first[type][name]; }
    __device__ __forceinline__ int icon(int type, int name, int k) const { return variants[first[type][name] + k]; }
};

// maze2d.spanning_tree_maze_generator: mz[y*D+x] = 1 for '#'.  Randomised DFS over the node lattice
// with an explicit stack; shuffle = Fisher-Yates i = 3..1, j = below(i+1) on [(-1,0),(1,0),(0,1),(0,-1)].
__device__ void xw_maze(Stream &s, int D, uint8_t *mz) {
    int X = D;
    const bool pad = (X % 2) == 0;
    if (pad) X -= 1;
    const int n = (X + 1) / 2;
    for (int y = 0; y < X; ++y)
        for (int x = 0; x < X; ++x) mz[y * D + x] = (x % 2 == 0 && y % 2 == 0) ? 0 : 1;
    uint64_t visited = 0;                     // n*n <= 64 nodes
    uint8_t st_node[64], st_perm[64], st_next[64];
    int sp = 0;
    st_node[0] = 0; st_next[0] = 0xff; sp = 1;
    while (sp > 0) {
        const int top = sp - 1;
        const int node = st_node[top];
        const int cx = node % n, cy = node / n;
        if (st_next[top] == 0xff) {
            visited |= 1ull << node;
            int mv[4] = {0, 1, 2, 3};
            for (int i = 3; i >= 1; --i) {
                const int j = (int)s.below((uint32_t)(i + 1));
                // swap mv[i], mv[j] without dynamic register indexing
                int vi = i == 3 ? mv[3] : (i == 2 ? mv[2] : mv[1]);
                int vj = j == 0 ? mv[0] : (j == 1 ? mv[1] : (j == 2 ? mv[2] : mv[3]));
                if (j == 0) mv[0] = vi; else if (j == 1) mv[1] = vi; else if (j == 2) mv[2] = vi; else mv[3] = vi;
                if (i == 3) mv[3] = vj; else if (i == 2) mv[2] = vj; else mv[1] = vj;
            }
            st_perm[top] = (uint8_t)(mv[0] | (mv[1] << 2) | (mv[2] << 4) | (mv[3] << 6));
            st_next[top] = 0;
        }
        if (st_next[top] >= 4) { sp--; continue; }
        const int m = (st_perm[top] >> (2 * st_next[top])) & 3;
        st_next[top] += 1;
        const int dx = m == 0 ? -1 : (m == 1 ? 1 : 0);
        const int dy = m == 2 ? 1 : (m == 3 ? -1 : 0);
        const int nx = cx + dx, ny = cy + dy;
        if (nx >= 0 && nx < n && ny >= 0 && ny < n && !((visited >> (ny * n + nx)) & 1ull)) {
            mz[(cy + ny) * D + (cx + nx)] = 0;              // open the wall between the two nodes
            st_node[sp] = (uint8_t)(ny * n + nx);
            st_next[sp] = 0xff;
            sp++;
        }
    }
    if (pad) {
        for (int i = 0; i < X; ++i) mz[X * D + i] = (i % 2 == 0) ? 0 : 1;
        for (int i = 0; i < D; ++i) mz[i * D + X] = (i % 2 == 0) ? 0 : 1;
    }
}

__device__ __forceinline__ int list_take(uint8_t *list, int &n, int k) {    // order-preserving remove
    const int v = list[k];
    for (int i = k; i + 1 < n; ++i) list[i] = list[i + 1];
    n -= 1;
    return v;
}

__device__ void xw_reset_env(const XwParams &p, const IconTables &T, int e, bool keep_done) {
    const int MD = p.max_dim, D = p.dim, off = (MD - D) / 2;
    const uint32_t ep = p.episode[e] + 1;
    p.episode[e] = ep;
    Stream s;
    s.init(p.seed, p.env_gid0 + (uint32_t)e, ep, 0);

    uint16_t cells[XW_MAX_DIM * XW_MAX_DIM];
    uint8_t avail[XW_MAX_DIM * XW_MAX_DIM];
    uint8_t blk[XW_MAX_DIM * XW_MAX_DIM];
    int na = 0, nb = 0;
    int goal_cell[XW_MAX_GOALS], goal_name[XW_MAX_GOALS];
    const int ng = p.num_goals;
    int agent_cell = 0;
    for (int i = 0; i < D * D; ++i) cells[i] = 0;

    if (p.map_kind == 0) {
        // ---- XWorldNav: distinct goal names (shuffle + pop), maze, shuffled '#' list, placement ----
        const int M = p.n_names[0];
        int ov_idx[XW_MAX_GOALS], ov_val[XW_MAX_GOALS], n_ov = 0;
        for (int i = 0; i < ng; ++i) {
            const int j = (int)s.below((uint32_t)(M - i));
            int vj = j, vl = M - 1 - i;
            for (int k = 0; k < n_ov; ++k) { if (ov_idx[k] == j) vj = ov_val[k]; if (ov_idx[k] == M - 1 - i) vl = ov_val[k]; }
            goal_name[i] = vj;
            bool found = false;                     // names[j] = names[M-1-i]
            for (int k = 0; k < n_ov; ++k) if (ov_idx[k] == j) { ov_val[k] = vl; found = true; }
            if (!found) { ov_idx[n_ov] = j; ov_val[n_ov] = vl; n_ov++; }
        }
        uint8_t mz[XW_MAX_DIM * XW_MAX_DIM];
        xw_maze(s, D, mz);
        for (int c = 0; c < D * D; ++c) { if (mz[c]) blk[nb++] = (uint8_t)c; else avail[na++] = (uint8_t)c; }
        for (int i = nb - 1; i >= 1; --i) {
            const int j = (int)s.below((uint32_t)(i + 1));
            const uint8_t t = blk[i]; blk[i] = blk[j]; blk[j] = t;
        }
        for (int i = 0; i < ng; ++i) {
            const int c = list_take(avail, na, (int)s.below((uint32_t)na));
            const int v = (int)s.below((uint32_t)T.nv(0, goal_name[i]));
            cells[c] = (uint16_t)(T.icon(0, goal_name[i], v) + 1);
            goal_cell[i] = c;
        }
        for (int i = 0; i < p.num_blocks; ++i) {
            const int c = blk[--nb];
            const int nm = (int)s.below((uint32_t)p.n_names[1]);
            const int v = (int)s.below((uint32_t)T.nv(1, nm));
            cells[c] = (uint16_t)(T.icon(1, nm, v) + 1);
        }
        {
            const int c = list_take(avail, na, (int)s.below((uint32_t)na));
            const int nm = (int)s.below((uint32_t)p.n_names[2]);
            const int v = (int)s.below((uint32_t)T.nv(2, nm));
            cells[c] = (uint16_t)(T.icon(2, nm, v) + 1);
            agent_cell = c;
        }
    } else {
        // ---- XWorldWalls: one full brick row, a partial brick column, then agent, goals, blocks ----
        for (int c = 0; c < D * D; ++c) avail[na++] = (uint8_t)c;
        int n_blocks = p.num_blocks;
        const int row = (int)s.below((uint32_t)D);
        const int first = n_blocks < D ? n_blocks : D;
        for (int i = 0; i < first; ++i) blk[nb++] = (uint8_t)(row * D + i);
        n_blocks -= first;
        const int column = (int)s.below((uint32_t)D);
        const int lim = n_blocks < D - 1 ? n_blocks : D - 1;
        for (int i = 0, j = 0; j < lim; ++i) if (i != row) { blk[nb++] = (uint8_t)(i * D + column); j++; }
        for (int i = 0; i < nb; ++i) {                      // remove wall cells from the free list
            int k = 0;
            while (avail[k] != blk[i]) ++k;
            (void)list_take(avail, na, k);
        }
        {   // agent
            const int c = list_take(avail, na, (int)s.below((uint32_t)na));
            const int nm = (int)s.below((uint32_t)p.n_names[2]);
            const int v = (int)s.below((uint32_t)T.nv(2, nm));
            cells[c] = (uint16_t)(T.icon(2, nm, v) + 1);
            agent_cell = c;
        }
        for (int i = 0; i < ng; ++i) {
            const int c = list_take(avail, na, (int)s.below((uint32_t)na));
            const int nm = (int)s.below((uint32_t)p.n_names[0]);
            const int v = (int)s.below((uint32_t)T.nv(0, nm));
            cells[c] = (uint16_t)(T.icon(0, nm, v) + 1);
            goal_cell[i] = c; goal_name[i] = nm;
        }
        for (int i = 0; i < nb; ++i) {
            const int nm = (int)s.below((uint32_t)p.n_names[1]);
            const int v = (int)s.below((uint32_t)T.nv(1, nm));
            cells[blk[i]] = (uint16_t)(T.icon(1, nm, v) + 1);
        }
    }

    // ---- XWorld3DNavTarget.idle: goals reachable from the agent (blocks and the other goals are
    // obstacles).  One flood fill of the empty cells from the agent; a goal is reachable iff one of its
    // 4-neighbours is the agent cell or a flooded cell (a path's interior can hold neither blocks nor goals).
    uint8_t *queue = avail;                       // free list no longer needed
    uint64_t seen[4] = {0, 0, 0, 0};
    auto mark = [&](int c) { seen[c >> 6] |= 1ull << (c & 63); };
    auto is_marked = [&](int c) { return (seen[c >> 6] >> (c & 63)) & 1ull; };
    int qh = 0, qt = 0;
    queue[qt++] = (uint8_t)agent_cell; mark(agent_cell);
    while (qh < qt) {
        const int c = queue[qh++];
        const int cx = c % D, cy = c / D;
        if (cx > 0 && !is_marked(c - 1) && cells[c - 1] == 0) { mark(c - 1); queue[qt++] = (uint8_t)(c - 1); }
        if (cx + 1 < D && !is_marked(c + 1) && cells[c + 1] == 0) { mark(c + 1); queue[qt++] = (uint8_t)(c + 1); }
        if (cy > 0 && !is_marked(c - D) && cells[c - D] == 0) { mark(c - D); queue[qt++] = (uint8_t)(c - D); }
        if (cy + 1 < D && !is_marked(c + D) && cells[c + D] == 0) { mark(c + D); queue[qt++] = (uint8_t)(c + D); }
    }
    int cand[XW_MAX_GOALS], nc = 0;
    for (int i = 0; i < ng; ++i) {
        const int c = goal_cell[i], cx = c % D, cy = c / D;
        const bool r = (cx > 0 && is_marked(c - 1)) || (cx + 1 < D && is_marked(c + 1)) ||
                       (cy > 0 && is_marked(c - D)) || (cy + 1 < D && is_marked(c + D));
        if (r) cand[nc++] = i;
    }
    int target = -1;                              // reference asserts nc > 0 ("map too crowded?")
    if (nc > 0) {
        const int k = (int)s.below((uint32_t)nc);
        int pick = cand[0];
        for (int i = 1; i < nc; ++i) if (i == k) pick = cand[i];
        target = goal_name[0];
        for (int i = 1; i < ng; ++i) if (i == pick) target = goal_name[i];
    }

    // ---- write the env's state: cells shifted by the padding offset, brick padding walls outside ----
    const uint16_t brick = (uint16_t)(T.icon(1, 0, 0) + 1);     // self.items["block"]["brick"][0]
    uint16_t *g = p.grid + (size_t)e * MD * MD;
    for (int y = 0; y < MD; ++y)
        for (int x = 0; x < MD; ++x) {
            const int lx = x - off, ly = y - off;
            g[y * MD + x] = (lx >= 0 && ly >= 0 && lx < D && ly < D) ? cells[ly * D + lx] : brick;
        }
    p.agent_xy[e] = (agent_cell % D + off) | ((agent_cell / D + off) << 16);
    p.task_state[e] = pack_task(target, STAGE_NAV, EV_NONE);
    p.task_steps[e] = 0;
    p.num_steps[e] = 0;
    p.fresh[e] = 1;
    if (!keep_done) p.done[e] = (uint8_t)done_code(p, 0, EV_NONE);
}

__global__ __launch_bounds__(64) void xw_reset_kernel(XwParams p, int mode, int keep_done, const int32_t *count_now) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    int e;
    if (mode == MODE_RESET_ALL) {
        if (i >= p.n) return;
        e = i;
    } else {
        if (i >= *count_now) return;
        e = p.done_list[i];
    }
    IconTables T;
    T.first[0] = p.name_first + p.name_first_off[0];
    T.first[1] = p.name_first + p.name_first_off[1];
    T.first[2] = p.name_first + p.name_first_off[2];
    T.variants = p.name_variants;
    xw_reset_env(p, T, e, keep_done != 0);
}

// ----------------------------------------------------------------- render --
// One output chunk = 16 consecutive bytes of an env's planar frame = 4 dwords, each of which lies
// inside one tile row (12 px = 3 dwords, frame rows are 3*D dwords).
template <int DIM_T, int CH>
__device__ __forceinline__ uint4 xw_expand_chunk(const uint32_t *atlas, const uint16_t *g, int cc, int dim_rt) {
    const int D = DIM_T ? DIM_T : dim_rt;
    const int RD = XW_TILE_DW * D;      // dwords per frame row
    const int RH = XW_TILE * D;         // rows per channel
    const int d0 = cc * 4;
    int ch = d0 / (RH * RD);
    const int rem = d0 - ch * (RH * RD);
    int y = rem / RD;
    int dx = rem - y * RD;
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cy = y / XW_TILE, py = y - cy * XW_TILE;
        const int cx = dx / XW_TILE_DW, kk = dx - cx * XW_TILE_DW;
        const uint32_t code = g[cy * D + cx];
        out[k] = atlas[code * (CH * 36) + ch * 36 + py * 3 + kk];   // tile 0 = empty cell (white)
        dx += 1;
        if (dx == RD) { dx = 0; y += 1; if (y == RH) { y = 0; ch += 1; } }
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void xw_store_chunk(uint4 *frame0, int cc, int chunks_per_frame, int ctx, bool fresh, uint4 v) {
    uint4 *q = frame0 + cc;
    if (ctx > 1) {
        // shift_context: oldest first; init_screen: zeros.  The same lane owns offset cc in every frame.
        if (fresh) for (int f = 0; f + 1 < ctx; ++f) q[(size_t)f * chunks_per_frame] = make_uint4(0, 0, 0, 0);
        else for (int f = 0; f + 1 < ctx; ++f) q[(size_t)f * chunks_per_frame] = q[(size_t)(f + 1) * chunks_per_frame];
    }
    // streamed once, never re-read by this kernel: one non-temporal global_store_dwordx4 per lane
    u32x4 nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(&q[(size_t)(ctx - 1) * chunks_per_frame]));
}

// all envs: persistent workgroups (one per CU), tile table resident in LDS, env tiles staged in LDS
template <int DIM_T, int CH>
__global__ __launch_bounds__(1024) void xw_render_all_kernel(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
    extern __shared__ uint4 smem4[];
    uint32_t *s_atlas = reinterpret_cast<uint32_t *>(smem4);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_atlas + atlas_dw);
    const int D = DIM_T ? DIM_T : p.max_dim;
    const int cells = D * D;
    uint8_t *s_fresh = reinterpret_cast<uint8_t *>(s_grid + tile_envs * cells);
    const int tid = threadIdx.x;
    const int ctx = p.context;
    const int cpf = CH * 9 * cells;                       // 16-byte chunks per frame: C*144*D*D/16
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.atlas);
        for (int i = tid; i < atlas_dw / 4; i += 1024) smem4[i] = src[i];
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e0 = tile * tile_envs;
        const int ne = min(tile_envs, p.n - e0);
        __syncthreads();
        const uint16_t *gsrc = p.grid + (size_t)e0 * cells;
        for (int i = tid; i < ne * cells; i += 1024) s_grid[i] = gsrc[i];
        if (ctx > 1 && tid < ne) { s_fresh[tid] = p.fresh[e0 + tid]; }
        __syncthreads();
        if (ctx > 1 && tid < ne) p.fresh[e0 + tid] = 0;
        const int total = ne * cpf;
        for (int c = tid; c < total; c += 1024) {
            const int le = c / cpf, cc = c - le * cpf;
            const uint4 v = xw_expand_chunk<DIM_T, CH>(s_atlas, s_grid + le * cells, cc, D);
            uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)(e0 + le) * ctx * cpf;
            xw_store_chunk(frame0, cc, cpf, ctx, ctx > 1 ? s_fresh[le] != 0 : false, v);
        }
    }
}

// the compacted list of freshly reset envs: one env per workgroup pass, tile table through L1/L2
template <int DIM_T, int CH>
__global__ __launch_bounds__(256) void xw_render_list_kernel(XwParams p, const int32_t *count_now) {
    __shared__ uint16_t s_grid[XW_MAX_DIM * XW_MAX_DIM];
    const int D = DIM_T ? DIM_T : p.max_dim;
    const int cells = D * D;
    const int ctx = p.context;
    const int cpf = CH * 9 * cells;
    const int cnt = *count_now;
    for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
        const int e = p.done_list[i];
        __syncthreads();
        for (int k = threadIdx.x; k < cells; k += 256) s_grid[k] = p.grid[(size_t)e * cells + k];
        __syncthreads();
        uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * ctx * cpf;
        for (int cc = threadIdx.x; cc < cpf; cc += 256) {
            const uint4 v = xw_expand_chunk<DIM_T, CH>(p.atlas, s_grid, cc, D);
            xw_store_chunk(frame0, cc, cpf, ctx, true, v);
        }
        if (threadIdx.x == 0) p.fresh[e] = 0;
    }
}

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

template <int DIM_T, int CH>
static hipError_t render_all(const XwParams &p, hipStream_t s) {
    const int cells = p.max_dim * p.max_dim;
    const int atlas_dw = (p.n_icons + 1) * CH * 36;
    const size_t atlas_bytes = (size_t)atlas_dw * 4;
    const size_t lds_cap = g_max_lds ? g_max_lds : 65536;
    const size_t per_env = (size_t)cells * 2 + 1;
    if (atlas_bytes + per_env + 64 > lds_cap) return hipErrorInvalidValue;
    int tile_envs = (int)((lds_cap - atlas_bytes - 64) / per_env);
    if (tile_envs > 16) tile_envs = 16;
    // even number of cells*tile so the fresh bytes start aligned; nothing else depends on it
    const int n_tiles = (p.n + tile_envs - 1) / tile_envs;
    const size_t lds = atlas_bytes + (size_t)tile_envs * per_env + 16;
    auto kern = xw_render_all_kernel<DIM_T, CH>;
    static size_t configured = 0;
    if (lds > 65536 && configured < lds) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        if (err != hipSuccess) return err;
        configured = lds_cap;
    }
    const int cus = g_num_cus ? g_num_cus : 256;
    dim3 grid(n_tiles < cus ? n_tiles : cus), block(1024);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p, tile_envs, n_tiles, atlas_dw);
    return hipGetLastError();
}

template <int DIM_T, int CH>
static hipError_t render_list(const XwParams &p, hipStream_t s) {
    dim3 grid(512), block(256);
    hipLaunchKernelGGL((xw_render_list_kernel<DIM_T, CH>), grid, block, 0, s, p, (const int32_t *)p.done_count);
    return hipGetLastError();
}

template <int CH>
static hipError_t render_dispatch(const XwParams &p, int indexed, hipStream_t s) {
#define XW_CASE(DIMV) case DIMV: return indexed ? render_list<DIMV, CH>(p, s) : render_all<DIMV, CH>(p, s);
    switch (p.max_dim) {
        XW_CASE(7) XW_CASE(8) XW_CASE(11)
        default: return indexed ? render_list<0, CH>(p, s) : render_all<0, CH>(p, s);
    }
#undef XW_CASE
}

hipError_t launch_xw_render(const XwParams &p, int indexed, hipStream_t s) {
    return p.channels == 3 ? render_dispatch<3>(p, indexed, s) : render_dispatch<1>(p, indexed, s);
}

hipError_t launch_xw_reset(const XwParams &p, int mode, hipStream_t s) {
    dim3 grid((p.n + 63) / 64), block(64);
    hipLaunchKernelGGL(xw_reset_kernel, grid, block, 0, s, p, mode, p.auto_reset, (const int32_t *)p.done_count);
    return hipGetLastError();
}

hipError_t launch_xw_compact(const XwParams &p, int mode, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    hipLaunchKernelGGL(xw_compact_kernel, grid, block, 0, s, p, mode, p.done_count);
    return hipGetLastError();
}

}  // namespace xwb
