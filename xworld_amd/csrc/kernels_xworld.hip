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
#include <cstdlib>
#include "xw_device.h"

namespace xwb {

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
        // render hand-off flag: 0 = env untouched (leave its context ring alone), 1 = stepped, 2 = fresh (reset)
        p.fresh[e] = (unsigned)a < 4u ? 1 : 0;
        if (a == ACTION_SKIP) {
            // XWB_ACTION_SKIP: this env does not take part in the call (per-slot SimulatorInterface views)
        } else if ((unsigned)a >= 4u) {        // CHECK_LT(action_idx, get_num_actions())
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
            int target = task_target(ts), kind = task_kind(ts);
            int stage = task_stage(ts);
            int tsteps = p.task_steps[e];
            int event = EV_NONE;
            double rew = 0.0;
            if (p.group2d) {
                // rule D14b (games/xworld/tasks/xworld_task.py:184-223): the group draws a task whenever its busy
                // task is idle (teaching_task.cpp:204-222), also at step time
                if (stage == STAGE_IDLE) {
                    Stream s2;
                    s2.init(p.seed, p.env_gid0 + (uint32_t)e, p.episode[e], 2u);
                    s2.blk = (uint32_t)steps;
                    const int k2 = p.tasks[s2.below((uint32_t)p.n_tasks)];
                    kind = k2;
                    idle_2d(kind, p.cand2d[e], p.goal_cells + (size_t)e * XW_MAX_GOALS,
                            [&](uint32_t n) { return s2.below(n); }, target, stage, tsteps);
                    rew = 0.0;
                } else if (stage == STAGE_NAV) {
                    rew = -0.1;                             // time_penalty
                    if (!success) rew += -0.2;              // failed_action_penalty
                    tsteps += 1;
                    if (p.task_mode == 1 && tsteps >= D * D / 2) {           // one_channel: h*w / 2 (max dims)
                        tsteps = 0;
                        stage = STAGE_IDLE;                 // _record_failure, "S -> timeup"
                    } else if (ay * D + ax == target) {     // agent.loc == self.target
                        tsteps = 0;
                        event = EV_CORRECT; rew += 1.0;
                        stage = STAGE_IDLE;
                    }
                    // `agent.loc in goal_locs` (-1.0) cannot hold: XMap::move_item never enters an occupied cell
                }
            } else if (stage == STAGE_NAV) {
                rew = -0.01;                                // time_penalty
                tsteps += 1;
                if (tsteps >= p.dim * p.dim * p.max_steps_factor) {
                    event = EV_TIMEUP;
                    stage = STAGE_TERMINAL;
                } else if (hit != 0 && a == 1 && p.icon_type[(hit & CELL_ICON_MASK) - 1] == 0) {
                    // _reach_object: id in collisions and |theta| < pi/4.  Full-observation entities keep
                    // yaw = 1.5707963 (heading +y), so theta = 0 only for a goal hit by MOVE_DOWN.
                    // Target / Near / Direction / Avoid: the reached goal is in self.target (cell bit 15, set by
                    // the idle stage) -> correct, else wrong.  Between: any reached goal is wrong.
                    if (kind != TASK_BETWEEN && (hit & CELL_TARGET_BIT)) { event = EV_CORRECT; rew += 1.0; }
                    else { event = EV_WRONG; rew += -1.0; }
                    stage = STAGE_TERMINAL;
                } else if (kind == TASK_BETWEEN && ay * D + ax == target) {
                    // XWorld3DNavTargetBetween.navigation_reward: dist(agent, middle) < threshold / 2
                    event = EV_CORRECT; rew += 1.0;
                    stage = STAGE_TERMINAL;
                }
            }
            float r = 0.0f;                                 // SimulatorInterface::take_actions
            r += 0.0f;                                      // XWorldSimulator::take_action returns 0
            r = (float)((double)r + rew);                   // r += teacher_->give_reward() (double)
            const int code = done_code(p, steps, event);
            p.agent_xy[e] = ax | (ay << 16);
            p.task_state[e] = pack_task(target, stage, event, kind);
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

// flag: 0 = env untouched by this call (nothing to do), 1 = stepped (ring shift), 2 = fresh (init_screen)
__device__ __forceinline__ void xw_store_chunk(uint4 *frame0, int cc, int chunks_per_frame, int ctx, int flag, uint4 v) {
    uint4 *q = frame0 + cc;
    if (ctx > 1) {
        if (flag == 0) return;
        const bool fresh = flag == 2;
        // shift_context: oldest first; init_screen: zeros.  The same lane owns offset cc in every frame.
        if (fresh) for (int f = 0; f + 1 < ctx; ++f) q[(size_t)f * chunks_per_frame] = make_uint4(0, 0, 0, 0);
        else for (int f = 0; f + 1 < ctx; ++f) q[(size_t)f * chunks_per_frame] = q[(size_t)(f + 1) * chunks_per_frame];
    }
    // streamed once, never re-read by this kernel: one non-temporal global_store_dwordx4 per lane
    u32x4 nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(&q[(size_t)(ctx - 1) * chunks_per_frame]));
}

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
__global__ __launch_bounds__(1024) void xw_render_all_kernel(XwParams p, int tile_envs, int n_tiles, int atlas_dw) {
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
        for (int k = threadIdx.x; k < cells; k += 256) s_grid[k] = p.grid[(size_t)e * cells + k] & CELL_ICON_MASK;
        __syncthreads();
        uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * ctx * cpf;
        for (int cc = threadIdx.x; cc < cpf; cc += 256) {
            const uint4 v = xw_expand_chunk<DIM_T, CH>(p.atlas, s_grid, cc, D);
            xw_store_chunk(frame0, cc, cpf, ctx, 2, v);
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
static hipError_t render_all(const XwParams &p, hipStream_t s) {
    RenderPlan r;
    hipError_t err = plan_render<CH>(p, 16, r);
    if (err != hipSuccess) return err;
    if (p.context == 1) {
        auto kern = xw_render_all_kernel<DIM_T, CH, true>;
        static size_t configured = 0;
        if ((err = allow_big_lds(kern, r.lds, configured)) != hipSuccess) return err;
        hipLaunchKernelGGL(kern, dim3(r.n_blocks), dim3(1024), r.lds, s, p, r.tile_envs, r.n_tiles, r.atlas_dw);
    } else {
        auto kern = xw_render_all_kernel<DIM_T, CH, false>;
        static size_t configured = 0;
        if ((err = allow_big_lds(kern, r.lds, configured)) != hipSuccess) return err;
        hipLaunchKernelGGL(kern, dim3(r.n_blocks), dim3(1024), r.lds, s, p, r.tile_envs, r.n_tiles, r.atlas_dw);
    }
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

hipError_t launch_xw_compact(const XwParams &p, int mode, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    hipLaunchKernelGGL(xw_compact_kernel, grid, block, 0, s, p, mode, p.done_count);
    return hipGetLastError();
}

}  // namespace xwb
