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
#include <cstring>
#include "xw_device.h"

namespace xwb {

#ifdef XWB_STEP_PROF
// lab build (XWB_EXTRA_FLAGS=-DXWB_STEP_PROF, tools/step_prof.py): 100 MHz stamps of the LAST launch's workgroups
__device__ unsigned long long g_step_prof[2][4096][6];    // [0] xw_step_kernel, [1] xw_render_list_kernel
#define SP_T(which, k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && threadIdx.x < 64) g_step_prof[which][blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define SP_T(which, k)
#endif

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

// ... the same with a second value per entry (the done list's episode counters: done_ep)
__device__ __forceinline__ void wave_append2(bool flag, int value, uint32_t value2, int32_t *list, uint32_t *list2, int32_t *count) {
    unsigned long long m = __ballot(flag);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (flag) {
        const int k = base + __popcll(m & ((1ull << lane) - 1ull));
        list[k] = value;
        list2[k] = value2;
    }
}

// What one teach() call hands a task group's stage (Task::py_stage pushes the same into the Python env)
struct StepCtx {
    int e, D, ax, ay, steps, hit, hit_cell, ddx, ddy, vx, vy;
    bool success;
    int level;
    bool hit_is_goal;           // the item bumped into is a goal (icon type 0)
};

// One task group's stage in one Teacher::teach call: group G's task FSM (ts_in / tsteps_in -> ts_out / tsteps_out), the
// reward it adds to the teacher buffer and the event it leaves there (every py_stage overwrites it).
template <int G>
__device__ __forceinline__ void teach_group(const XwParams &p, const StepCtx &c, int ts, int tsteps_in,
                                            double &rew, int &event, int &ts_out, int &tsteps_out, bool &defer_idle) {
    defer_idle = false;
    const int e = c.e, D = c.D, ax = c.ax, ay = c.ay, steps = c.steps, hit = c.hit, hit_cell = c.hit_cell;
    const int ddx = c.ddx, ddy = c.ddy, vx = c.vx, vy = c.vy, ld_level = c.level;
    const bool success = c.success;
    const bool group2d = G ? p.group2d_2 != 0 : p.group2d != 0;
    int target = task_target(ts), kind = task_kind(ts);
    int stage = task_stage(ts);
    int tsteps = tsteps_in;
    event = EV_NONE;
    int record = -1;                                // the result this step adds to the task's window / counters
    bool timeup = false;
    rew = 0.0;
    if (group2d) {
        // rule D14b (games/xworld/tasks/xworld_task.py:184-223): the group draws a task whenever its busy
        // task is idle (teaching_task.cpp:204-222), also at step time
        if (stage == STAGE_IDLE) {
            Stream s2;
            s2.init(p.seed, p.env_gid0 + (uint32_t)e, p.episode[e], 2u);
            s2.blk = (uint32_t)steps;
            const int k2 = task_at<G>(p, sample_task<G>(p, s2, e));
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
                record = 0;                         // _record_failure
                timeup = true;
                stage = STAGE_IDLE;                 // "S -> timeup"
            } else if (ay * D + ax == target) {     // agent.loc == self.target
                tsteps = 0;
                record = 1;                         // _record_success
                event = EV_CORRECT; rew += 1.0;
                stage = STAGE_IDLE;
            }
            // `agent.loc in goal_locs` (-1.0) cannot hold: XMap::move_item never enters an occupied cell
        }
    } else if (stage == STAGE_IDLE) {
        // an idle XWorld3DNav* group picked in mid-episode (exclusive scheduling only: at reset the group's idle stage has
        // always run).  TaskGroup::run_stage draws a task and runs its idle stage, which rearranges the map: reward 0, no
        // event -- the stage itself runs in xw_idle3d_kernel, right behind this kernel, over the envs listed here.
        defer_idle = true;
    } else if (stage == STAGE_NAV) {
        rew = -0.01;                                // time_penalty
        tsteps += 1;
        const int dim = p.curriculum != 0 ? 3 + ld_level : p.dim;             // env.get_dims()
        if (tsteps >= dim * dim * p.max_steps_factor) {
            event = EV_TIMEUP;
            record = 0;
            timeup = true;
            stage = STAGE_TERMINAL;
        } else if (hit != 0 && ddx == vx && ddy == vy && c.hit_is_goal) {
            // _reach_object: id in collisions and |theta| < pi/4, i.e. the goal was bumped into along the
            // heading: MOVE_DOWN under full observation (yaw stays 1.5707963), MOVE_FORWARD in egocentric mode.
            // Target / Near / Avoid: the reached goal is in self.target (cell bit 15, set by the idle stage)
            // -> correct, else wrong.  Between: any reached goal is wrong.  Direction: (direction(g, referent,
            // agent.yaw), near) is evaluated now, the yaw being the current heading.
            bool good = kind != TASK_BETWEEN && (hit & CELL_TARGET_BIT);
            if (kind == TASK_DIRECTION && target >= 0) {        // (a replayed map may carry the bits only)
                const int rc = target & 0xff, word = (target >> 8) & 7;
                const int v2x = rc % D - hit_cell % D, v2y = rc / D - hit_cell / D;
                const int cs = vx * v2x + vy * v2y, sn = vy * v2x - vx * v2y;
                const int dirw = cs > 0 ? DIR_FRONT : (cs < 0 ? DIR_BEHIND : (sn > 0 ? DIR_RIGHT : DIR_LEFT));
                good = v2x * v2x + v2y * v2y == 1 && dirw == word;
            }
            if (good) { event = EV_CORRECT; rew += 1.0; }
            else { event = EV_WRONG; rew += -1.0; }
            record = good ? 1 : 0;                  // _successful_goal / _failed_goal
            stage = STAGE_TERMINAL;
        } else if (kind == TASK_BETWEEN && ay * D + ax == target) {
            // XWorld3DNavTargetBetween.navigation_reward: dist(agent, middle) < threshold / 2
            event = EV_CORRECT; rew += 1.0;
            record = 1;
            stage = STAGE_TERMINAL;
        }
    }
    if (record >= 0 && p.curriculum != 0) usage_push(p.cur_usage + ((size_t)e * 9 + kind) * XW_USAGE_BYTES, record);
    if (record >= 0) {
        // XWorld(3D)Task._record_success / _record_failure (xworld3d_task.py:135-142, xworld_task.py:93-99): the tallies
        // Task::obtain_performance hands to Teacher::report_task_performance; only the XWorld3D tasks keep success_steps
        unsigned long long *c = p.perf + kind * 4;
        atomicAdd(c + (record ? 0 : 1), 1ull);
        if (record && !group2d) atomicAdd(c + 2, (unsigned long long)tsteps);
        if (timeup) atomicAdd(c + 3, 1ull);
    }
    ts_out = pack_task(target, stage, event, kind);
    tsteps_out = tsteps;
}

// ------------------------------------------------------------------- step --
// What the step reads of an env before it moves (one round trip, every load independent of the others)
struct StepIn {
    int axy, steps, ts, tsteps, ts2, tsteps2, dir, level, action;
    uint32_t ep;
    uint4 gc;                                              // the env's goal-slot table (goal_cells)
};
// ... and what the move leaves for the teacher
struct Move {
    int ax, ay, hit, hit_cell, ddx, ddy, vx, vy, dir;
    bool success;
};

__device__ __forceinline__ void xw_load_step_in(const XwParams &p, int e, StepIn &in) {
    in.axy = p.agent_xy[e]; in.steps = p.num_steps[e]; in.ts = p.task_state[e]; in.tsteps = p.task_steps[e];
    in.ts2 = 0; in.tsteps2 = 0; in.dir = 1; in.level = 0; in.action = 0; in.ep = 0;
    if (p.n_tasks2 > 0) { in.ts2 = p.task_state2[e]; in.tsteps2 = p.task_steps2[e]; }
    if (p.visible_radius) in.dir = p.agent_dir[e];
    if (p.curriculum != 0) in.level = p.cur_level[e];
    if (p.actions) in.action = p.actions[e];
    if (p.swap_shadow == 2) in.ep = p.episode[e];
    in.gc = reinterpret_cast<const uint4 *>(p.goal_cells)[e];
}

// XAgent::act x act_rep on the env's grid.  `lg` = a private copy of the grid the move reads and keeps current (LDS);
// `g` = the live grid, whose two changed cells are stored as well when it is given.
__device__ __forceinline__ Move xw_move(const XwParams &p, int e, int a, int axy, int dir_in, uint16_t *lg, uint16_t *g) {
    const int D = p.max_dim;
    Move m;
    m.ax = axy & 0xffff; m.ay = axy >> 16;
    m.dir = dir_in;
    const uint16_t agent_code = lg[m.ay * D + m.ax];
    m.ddx = a == 2 ? -1 : (a == 3 ? 1 : 0);             // MOVE_LEFT / MOVE_RIGHT
    m.ddy = a == 0 ? -1 : (a == 1 ? 1 : 0);             // MOVE_UP / MOVE_DOWN
    m.vx = 0; m.vy = 1;                                  // heading: entities keep yaw 1.5707963 (+y) under full observation
    m.hit = 0; m.hit_cell = 0;
    m.success = false;
    for (int i = 0; i < p.act_rep; ++i) {
        if (p.visible_radius) {
            // XAgent::act, xitem.cpp:103-155: MOVE_FORWARD, MOVE_BACKWARD, MOVE_LEFT_FPV, MOVE_RIGHT_FPV relative to
            // the heading; TURN_LEFT / TURN_RIGHT change the yaw and "move" onto the agent's own cell, which
            // XMap::move_item refuses (xmap.cpp:76-101): a turn is an unsuccessful action without contacts
            int dir = m.dir;
            if (a == 4) dir = (dir + 3) & 3;
            else if (a == 5) dir = (dir + 1) & 3;
            m.dir = dir;
            p.agent_dir[e] = (uint8_t)dir;
            m.vx = dir == 0 ? 1 : (dir == 2 ? -1 : 0);
            m.vy = dir == 1 ? 1 : (dir == 3 ? -1 : 0);
            const int lx = m.vy, ly = -m.vx;            // MOVE_LEFT_FPV: right->up, down->right, left->down, up->left
            m.ddx = a == 0 ? m.vx : (a == 1 ? -m.vx : (a == 2 ? lx : (a == 3 ? -lx : 0)));
            m.ddy = a == 0 ? m.vy : (a == 1 ? -m.vy : (a == 2 ? ly : (a == 3 ? -ly : 0)));
        }
        const int tx = m.ax + m.ddx, ty = m.ay + m.ddy;
        m.success = false;
        if (p.visible_radius && a >= 4) continue;       // a turn
        if (tx >= 0 && ty >= 0 && tx < D && ty < D) {
            const int code = lg[ty * D + tx];
            if (code == 0) {                             // XMap::move_item: empty cell -> move
                lg[m.ay * D + m.ax] = 0;
                lg[ty * D + tx] = agent_code;
                if (g) { g[m.ay * D + m.ax] = 0; g[ty * D + tx] = agent_code; }
                m.ax = tx; m.ay = ty;
                m.success = true;
            } else {
                m.hit = code;                            // contact_list -> "collision:<id>" event
                m.hit_cell = ty * D + tx;
            }
        }
    }
    return m;
}

// Teacher::teach + XWorldSimulator::game_over for one env after its move, and the stores of everything the step leaves
// behind except the grid.
__device__ __forceinline__ void xw_teach_store(const XwParams &p, int e, const StepIn &in, const Move &m, bool &is_done, bool &idle3d) {
    const int D = p.max_dim;
    const int steps = in.steps + 1;                       // GameSimulator::take_actions: once per call
    const int hit = m.hit, hit_cell = m.hit_cell;
    // "the item bumped into is a goal": its cell is in the env's goal-slot table (0xff = no goal; cell 255 exists on a
    // 16 x 16 map only, where the icon's type is looked up instead)
    bool hit_is_goal = false;
    if (hit != 0) {
        if (D > 15) hit_is_goal = p.icon_type[(hit & CELL_ICON_MASK) - 1] == 0;
        else {
            const uint32_t rep = (uint32_t)hit_cell * 0x01010101u;
            auto has = [&](uint32_t w) { const uint32_t x = w ^ rep; return ((x - 0x01010101u) & ~x & 0x80808080u) != 0u; };
            hit_is_goal = has(in.gc.x) || has(in.gc.y) || has(in.gc.z) || has(in.gc.w);
        }
    }
    StepCtx cx{e, D, m.ax, m.ay, steps, hit, hit_cell, m.ddx, m.ddy, m.vx, m.vy, m.success, in.level, hit_is_goal};
    const int ld_ts = in.ts, ld_tsteps = in.tsteps, ld_ts2 = in.ts2, ld_tsteps2 = in.tsteps2;
    double rew = 0.0;
    int event = EV_NONE;
    if (p.exclusive && p.n_tasks2 > 0) {
        // Teacher::teach, exclusive branch (teacher.cpp:209-220): re-sort the groups, then run ONE: the last busy
        // group of the sorted list (the reference's loop has no break), else its first.  A busy 3-D group stays busy
        // until the game resets (its "terminal" stage returns "terminal"), a 2-D one until it is back in "idle".
        const int go = p.grp_order[e];
        const int first = xw_sort_groups(p, e, p.episode[e], (uint32_t)steps, go & 1), second = first ^ 1;
        const bool busy0 = task_stage(ld_ts) != STAGE_IDLE, busy1 = task_stage(ld_ts2) != STAGE_IDLE;
        const bool busy_first = first ? busy1 : busy0, busy_second = first ? busy0 : busy1;
        const int pick = busy_second ? second : (busy_first ? first : first);
        int ts_new, tsteps_new;
        double r0;
        bool defer;
        if (pick == 0) {
            teach_group<0>(p, cx, ld_ts, ld_tsteps, r0, event, ts_new, tsteps_new, defer);
            p.task_state[e] = ts_new;
            p.task_steps[e] = tsteps_new;
        } else {
            teach_group<1>(p, cx, ld_ts2, ld_tsteps2, r0, event, ts_new, tsteps_new, defer);
            p.task_state2[e] = ts_new;
            p.task_steps2[e] = tsteps_new;
        }
        rew = 0.0 + r0;
        idle3d = defer;
        p.grp_order[e] = (uint8_t)(first | (pick << 1));
    } else {
        // Teacher::teach (teacher.cpp:207-230), groups run non-exclusively in conf order: each group's Task stage adds
        // its reward to the teacher buffer and overwrites the buffer's event ("" included); only the first py_stage
        // of a teach() sees this step's collisions (XWorldSimulator::get_events_of_game clears them,
        // xworld_simulator.cpp:118-122).  One group is the usual case.
        bool defer;
        if (p.exclusive && p.minstd) {      // one group: the sort still draws once from the reference's engine
            uint32_t x = p.minstd[e];
            (void)xwb_minstd_rand_range_state(&x, (float)p.group_weight[0]);
            p.minstd[e] = x;
        }
        {
            int ts_new, tsteps_new;
            double r0;
            teach_group<0>(p, cx, ld_ts, ld_tsteps, r0, event, ts_new, tsteps_new, defer);
            rew = 0.0 + r0;                             // add_teacher_reward on a cleared buffer
            p.task_state[e] = ts_new;
            p.task_steps[e] = tsteps_new;
        }
        if (p.n_tasks2 > 0) {
            cx.hit = 0;                                 // game_events_ was consumed by the first group's py_stage
            int ts_new, tsteps_new;
            double r1;
            teach_group<1>(p, cx, ld_ts2, ld_tsteps2, r1, event, ts_new, tsteps_new, defer);
            rew += r1;
            p.task_state2[e] = ts_new;
            p.task_steps2[e] = tsteps_new;
        }
    }
    float r = 0.0f;                                 // SimulatorInterface::take_actions
    r += 0.0f;                                      // XWorldSimulator::take_action returns 0
    r = (float)((double)r + rew);                   // r += teacher_->give_reward() (double)
    const int code = done_code(p, steps, event);
    p.agent_xy[e] = m.ax | (m.ay << 16);
    p.num_steps[e] = steps;
    p.success[e] = m.success ? 1 : 0;
    p.reward[e] = r;
    p.done[e] = (uint8_t)code;
    if (p.packed) p.packed[e] = make_float2(r, (float)code);
    is_done = code != ALIVE;
}

// The step of `epb` consecutive envs by ONE wavefront (lane = env): the body of xw_step_kernel (epb = 64, one wavefront per
// workgroup) and of the step blocks of xw_step_render_kernel (the first wavefront of a 128-thread block; epb = as many grids
// as that kernel's LDS holds).
__device__ __forceinline__ void xw_step_body(const XwParams &p, const int blk, const int epb, uint4 *s_dyn) {
    // The kernel is a chain of dependent memory round trips for a few hundred wavefronts (it moves ~4 MB): everything that
    // does not depend on a previous load is fetched in the FIRST round trip -- the env's scalars, its goal-slot table (which
    // answers "is the thing I bumped into a goal" without a look-up at the end of the chain) and the GRIDS of the wavefront's
    // 64 envs, one contiguous block read cooperatively with 16-byte loads into LDS: the move then reads its cells from LDS
    // instead of paying a second round trip that depends on the agent's position.  One wavefront per workgroup: 512 workgroups
    // at C4, so every CU has one or two (256-thread groups left half the chip idle).
    SP_T(0, 0);
    uint16_t *s_grid = reinterpret_cast<uint16_t *>(s_dyn);   // [epb][max_dim^2] cell codes of this wavefront's envs
    const int lane = threadIdx.x;
    const int e0 = blk * epb, e = lane < epb ? e0 + lane : p.n;   // (a lane without an env: e = n)
    const int cells_all = p.max_dim * p.max_dim;
    int32_t *count_now = p.done_count;
    if (e == 0) {
        // (pre-generated episodes: the regeneration of the previous step's list may still be reading that list and its count --
        // the counter zeroed here is the one it read, two steps back in the rotation)
        if (p.swap_shadow) xw_wait_epoch_lane(p.sync + 8, p.regen_wait, p.sync + 4, p.poison_host);
        *p.done_count_next = 0;                // double-buffered done counter: zero the next step's
    }
    bool is_done = false, idle3d = false;
    if (e == 0 && p.idle_count_next) *p.idle_count_next = 0;
    StepIn in{};
    in.dir = 1;
    in.gc = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (e < p.n) xw_load_step_in(p, e, in);
    int axy_new = in.axy;                                  // (an env that sits this call out keeps its cell)
    {
        const int n_here = p.n - e0 < epb ? p.n - e0 : epb;
        const int total = n_here * cells_all;              // u16 elements of this block; the block starts 16-byte aligned
        const uint16_t *src = p.grid + (size_t)e0 * cells_all;
        const int full = total / 8;
        for (int c = lane; c < full; c += 64) s_dyn[c] = reinterpret_cast<const uint4 *>(src)[c];
        for (int k = full * 8 + lane; k < total; k += 64) s_grid[k] = src[k];
    }
    __syncthreads();
    SP_T(0, 1);
    if (e < p.n) {
        const int NA = p.visible_radius ? 6 : 4;           // XAgent legal_actions_, xitem.cpp:80-87
        const int a = p.actions ? in.action : policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, p.policy_step, NA);
        p.actions_out[e] = a;
        // render hand-off flag: 0 = env untouched (leave its context ring alone), 1 = stepped, 2 = fresh (reset)
        p.fresh[e] = (unsigned)a < (unsigned)NA ? 1 : 0;
        if (a == ACTION_SKIP) {
            // XWB_ACTION_SKIP: this env does not take part in the call (per-slot SimulatorInterface views)
        } else if ((unsigned)a >= (unsigned)NA) {   // CHECK_LT(action_idx, get_num_actions())
            atomicAdd(p.err_count, 1);
        } else {
            // the live grid gets the move's two cells; they are read from (and kept current in) the wavefront's LDS copy
            const Move m = xw_move(p, e, a, in.axy, in.dir, s_grid + lane * cells_all, p.grid + (size_t)e * cells_all);
            xw_teach_store(p, e, in, m, is_done, idle3d);
            axy_new = m.ax | (m.ay << 16);
        }
    }
    SP_T(0, 2);
    if (p.snap_grid_out) {
        // Look-ahead (XwParams::snap_grid_out): the built-in policy's NEXT action is known now, so the wavefront's LDS copy of the
        // grids -- this step's moves applied -- gets the next step's move as well and goes to the snapshot that step's render
        // blocks draw from while its own step blocks rewrite the live state beside them (xw_step_render_kernel).
        if (e < p.n) {
            uint16_t *lg = s_grid + lane * cells_all;
            int from;
            const int to = xw_predict_move(lg, p.max_dim, axy_new, policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, p.policy_step + 1u, 4),
                                           p.act_rep, &from);
            if (to != from) { lg[to] = lg[from]; lg[from] = 0; }
        }
        __syncthreads();                                   // (one wavefront: the lanes' moves are in the LDS copy)
        const int n_here = p.n - e0 < epb ? p.n - e0 : epb;
        const int total = n_here * cells_all, full = total / 8;
        uint16_t *dst = p.snap_grid_out + (size_t)e0 * cells_all;
        for (int c = lane; c < full; c += 64) reinterpret_cast<uint4 *>(dst)[c] = s_dyn[c];
        for (int k = full * 8 + lane; k < total; k += 64) dst[k] = s_grid[k];
    }
    if (p.swap_shadow == 2) {
        // a plain step whose reset_done will install pre-generated episodes (list_swap): nothing to install here, but the list
        // appended below may still be read by the previous step's regeneration
        if (__ballot(is_done)) {
            if ((threadIdx.x & 63) == 0) xw_wait_epoch_lane(p.sync + 8, p.regen_wait, p.sync + 4, p.poison_host);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (p.swap_shadow == 1) {
        // xwb_step_autoreset with pre-generated episodes: a finished env starts its next episode here -- its shadow state (the
        // reset kernel's output for episode + 1, made beside an earlier render) is copied over the live state; reward and code
        // keep the terminal transition's values, the render that follows draws the new episode's first frame.  The list
        // (appended below) is what the side queue regenerates next; it and the shadows of envs that finish again are only
        // touched once the previous regeneration is through.
        const unsigned long long m = __ballot(is_done);
        if (m) {
            const int lane = threadIdx.x & 63, cells = p.max_dim * p.max_dim;
            if (lane == __ffsll((long long)m) - 1) xw_wait_epoch_lane(p.sync + 8, p.regen_wait, p.sync + 4, p.poison_host);
            __builtin_amdgcn_wave_barrier();
            const uint32_t ep_old = is_done ? p.episode[e] : 0u;
            const int slot = (int)((ep_old + 1u) & 1u);                                  // the shadow slot that holds episode + 1
            const size_t es = (size_t)slot * (size_t)p.n + (size_t)e;
            if (is_done) {
                p.agent_xy[e] = p.sh_agent_xy[es];
                p.task_state[e] = p.sh_task_state[es];
                p.task_steps[e] = 0;
                if (p.n_tasks2 > 0) { p.task_state2[e] = p.sh_task_state2[es]; p.task_steps2[e] = 0; }
                p.sent_names[e] = p.sh_sent_names[es];
                p.cand2d[e] = p.sh_cand2d[es];
                reinterpret_cast<uint4 *>(p.goal_cells)[e] = reinterpret_cast<const uint4 *>(p.sh_goal_cells)[es];
                p.num_steps[e] = 0;
                p.episode[e] = ep_old + 1u;
                p.fresh[e] = 2;                         // init_screen: the older context frames start black
                atomicAdd(p.perf + 36, 1ull);           // games reset
            }
            __threadfence();                            // the agent's move was stored by one lane; the copy below overwrites it
            unsigned long long mm = m;
            while (mm) {
                const int j = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const int ej = __shfl(e, j);
                const uint16_t *src = p.sh_grid + ((size_t)__shfl(slot, j) * (size_t)p.n + (size_t)ej) * cells;
                uint16_t *g = p.grid + (size_t)ej * cells;
                for (int c = lane; c < cells; c += 64) g[c] = src[c];
            }
        }
    }
    // (lazy path: the list carries each env's episode counter, so the installing list render finds the shadow slot without a
    // round trip of its own)
    if (p.swap_shadow == 2) wave_append2(is_done, e, in.ep, p.done_list, p.done_ep, count_now);
    else wave_append(is_done, e, p.done_list, count_now);
    if (p.idle_list) wave_append(idle3d, e, p.idle_list, p.idle_count);
    SP_T(0, 3);
    // "this step finished the env": stays put until the next step, whatever a reset_done does to the done codes meanwhile
    if (e < p.n) p.term_flag[e] = is_done ? 1 : 0;
    SP_T(0, 4);
    if (!p.visible_radius && !p.swap_shadow) {
        // terminal snapshot: the frame of a finished env is rendered from this copy, which lets xwb_reset_done rebuild
        // the live grid on the side stream while the big render is still running.  The wavefront copies the grids of
        // its finished envs together (consecutive lanes = consecutive cells).
        unsigned long long m = __ballot(is_done);
        __syncthreads();                                   // (one wavefront: the lanes' LDS copies are complete)
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int ej = __shfl(e, j);
            uint16_t *t = p.term_grid + (size_t)ej * cells_all;
            for (int c = lane; c < cells_all; c += 64) t[c] = s_grid[j * cells_all + c];
        }
    }
}

__global__ __launch_bounds__(64) void xw_step_kernel(XwParams p) {
    extern __shared__ uint4 s_dyn[];                       // [64][max_dim^2] cell codes of this wavefront's envs
    xw_step_body(p, (int)blockIdx.x, 64, s_dyn);
}

hipError_t launch_xw_step(const XwParams &p, hipStream_t s) {
    dim3 grid((p.n + 63) / 64), block(64);
    const size_t lds = ((size_t)64 * p.max_dim * p.max_dim * 2 + 15) & ~(size_t)15;
    hipLaunchKernelGGL(xw_step_kernel, grid, block, lds, s, p);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void xw_wait_kernel(const uint32_t *epoch_slot, uint32_t want, uint32_t *poison, uint32_t *poison_host,
                                                      unsigned long long budget) {
    xw_wait_epoch(epoch_slot, want, poison, poison_host, budget);
}

hipError_t launch_xw_wait(const uint32_t *epoch_slot, uint32_t want, uint32_t *poison, uint32_t *poison_host, hipStream_t s,
                          unsigned long long budget_ticks) {
    hipLaunchKernelGGL(xw_wait_kernel, dim3(1), dim3(64), 0, s, epoch_slot, want, poison, poison_host,
                       budget_ticks ? budget_ticks : XW_WATCHDOG_TICKS);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void xw_signal_kernel(uint32_t *epoch_slot, uint32_t value) {
    if (threadIdx.x == 0) xw_publish_epoch(epoch_slot, value);
}

hipError_t launch_xw_signal(uint32_t *epoch_slot, uint32_t value, hipStream_t s) {
    hipLaunchKernelGGL(xw_signal_kernel, dim3(1), dim3(64), 0, s, epoch_slot, value);
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
template <int DIM_T, int CH, int ES>
__device__ __forceinline__ uint4 xw_expand_chunk(const uint32_t *atlas, const uint16_t *g, int cc, int dim_rt) {
    const int D = DIM_T ? DIM_T : dim_rt;
    const int RD = XW_TILE_DW * D;      // u8 dwords (4-pixel groups) per frame row
    const int RH = XW_TILE * D;         // rows per channel
    if (ES == 4) {
        // float32 frames: a 16-byte chunk is ONE 4-pixel group; the table holds it as one aligned uint4
        int ch = cc / (RH * RD);
        const int rem = cc - ch * (RH * RD);
        const int y = rem / RD, dx = rem - y * RD;
        const int cy = y / XW_TILE, py = y - cy * XW_TILE;
        const int cx = dx / XW_TILE_DW, kk = dx - cx * XW_TILE_DW;
        const uint32_t code = g[cy * D + cx];
        return reinterpret_cast<const uint4 *>(atlas)[code * (CH * 36) + ch * 36 + py * 3 + kk];
    }
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

// SKIP_DONE (xwb_step_autoreset): frames of finished envs are left to the list render that follows their reset on
// the side stream, so that render runs beside this kernel instead of after it.
// ES = bytes per pixel: 1 = uint8 frames; 4 = float32 frames (pixel * 1/255, py_simulator.cpp:262-272) expanded from
// a float copy of the tile table (628 KB, still L2-resident): the same kernel with 48-byte tile rows.
// LDS of one span: s_out4 [SPAN + PAD / 4 + TD / 4 + 1] uint4, s_code [SPAN * 16 / (144 * CH * ES) + 2 * XW_MAX_DIM^2] u16, s_done [NE]
template <int CH, int BS, int PER, int ES>
struct RenderLds {
    static constexpr int SPAN = BS * PER, TD = 3 * ES, PAD = (TD + 3) / 4 * 4;
    static constexpr int OUT4 = SPAN + PAD / 4 + TD / 4 + 1;
    static constexpr int CODES = SPAN * 16 / (144 * CH * ES) + 2 * XW_MAX_DIM * XW_MAX_DIM;
    static constexpr int NE = SPAN * 16 / (144 * CH * ES) + 2;               // envs a span can touch
};

// One span (workgroup `blk`) of the whole-batch render.  SNAP (xw_step_render_kernel): the cell codes come from the look-ahead
// snapshot the previous step left (p.snap_grid_in: the grids with THIS step's moves already applied).
template <int DIM_T, int CH, bool CTX1, int BS, int PER, int RMODE, int ES, bool SNAP>
__device__ __forceinline__ void xw_render_span(const XwParams &p, const unsigned blk, uint4 *s_out4, uint16_t *s_code, uint8_t *s_done) {
    constexpr bool SKIP_DONE = RMODE == 2, TERM = RMODE == 3;
    constexpr int SPAN = BS * PER;
    constexpr int TB = 12 * ES, TD = 3 * ES;                               // bytes / dwords per tile row
    constexpr int PAD = (TD + 3) / 4 * 4;                                   // dword index of the span's first chunk
    constexpr int IT = ((SPAN * 16 + TB - 1) / TB + 1 + BS - 1) / BS;       // tile rows per lane
    uint32_t *s_out = reinterpret_cast<uint32_t *>(s_out4);
    const int D = DIM_T ? DIM_T : p.max_dim;
    const int cells = D * D;
    const unsigned PB = 144u * ES * cells, FB = CH * PB, RB = 12u * ES * D;   // bytes per plane, frame, frame row
    const int cpf = (int)(FB / 16);
    const int tid = threadIdx.x;
    const unsigned long long n_chunks = (unsigned long long)p.n * cpf;
    const unsigned long long c_lo = (unsigned long long)blk * SPAN;
    const unsigned long long c_hi = c_lo + SPAN < n_chunks ? c_lo + SPAN : n_chunks;
    const unsigned long long b_lo = c_lo * 16, b_hi = c_hi * 16;
    const int e0 = (int)(b_lo / FB), e1 = (int)((b_hi - 1) / FB);
    const int ncode = (e1 - e0 + 1) * cells;
    for (int i = tid; i < ncode; i += BS) {
        const size_t gi = (size_t)e0 * cells + i;
        if (SNAP) { s_code[i] = p.snap_grid_in[gi] & CELL_ICON_MASK; continue; }
        // (TERM: the flag, the live cell and the snapshot's cell are fetched together and selected -- flag-then-cell was two
        // dependent round trips at the head of every workgroup; worth ~1 us of the 101 on C4)
        // (uint8 frames; float32 frames -- four times the bytes per workgroup -- measured better with the dependent form)
        uint16_t code;
        if (TERM && ES == 1) { const uint16_t live = p.grid[gi], snap = p.term_grid[gi]; code = p.term_flag[e0 + i / cells] ? snap : live; }
        else code = (TERM && p.term_flag[e0 + i / cells] ? p.term_grid : p.grid)[gi];
        s_code[i] = code & CELL_ICON_MASK;
    }
    if (SKIP_DONE) for (int i = tid; i <= e1 - e0; i += BS) s_done[i] = p.done[e0 + i];
    __syncthreads();
    // TB-byte units [u0, u1) cover the span; env and plane boundaries are multiples of TB, so flooring b_lo to a unit
    // never leaves env e0
    const unsigned long long u0 = b_lo / TB, u1 = (b_hi + TB - 1) / TB;
    const int nu = (int)(u1 - u0);
    const unsigned r0 = (unsigned)(u0 * TB - (unsigned long long)e0 * FB);   // byte offset of unit u0 inside env e0
    const int shift = PAD - (int)(b_lo - u0 * TB) / 4;                        // dword index of unit u0: chunk 0 = dword PAD
    uint32_t v[IT][TD];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = it * BS + tid;
        const unsigned rr = r0 + (unsigned)TB * (unsigned)(i < nu ? i : 0);
        const unsigned le = rr / FB, r = rr - le * FB;
        const unsigned ch = r / PB, r2 = r - ch * PB, y = r2 / RB, cx = (r2 - y * RB) / (unsigned)TB, cy = y / 12u, py = y - cy * 12u;
        const uint32_t code = s_code[le * cells + cy * D + cx];
        const uint32_t *src = p.atlas + (code * (CH * 36) + ch * 36 + py * 3) * ES;   // tile 0 = empty cell (white)
        if (ES == 4) {
#pragma unroll
            for (int q = 0; q < TD / 4; ++q) {
                const uint4 t = reinterpret_cast<const uint4 *>(src)[q];
                v[it][4 * q] = t.x; v[it][4 * q + 1] = t.y; v[it][4 * q + 2] = t.z; v[it][4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int d = 0; d < TD; ++d) v[it][d] = src[d];
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = it * BS + tid;
        if (i < nu) {
            const int o = shift + TD * i;
            if (ES == 4) {
#pragma unroll
                for (int q = 0; q < TD / 4; ++q)
                    s_out4[o / 4 + q] = make_uint4(v[it][4 * q], v[it][4 * q + 1], v[it][4 * q + 2], v[it][4 * q + 3]);
            } else {
#pragma unroll
                for (int d = 0; d < TD; ++d) s_out[o + d] = v[it][d];
            }
        }
    }
    __syncthreads();
    const int nc = (int)(c_hi - c_lo);
    uint4 *obs4 = reinterpret_cast<uint4 *>(p.obs);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = k * BS + tid;
        if (c >= nc) break;
        const uint4 val = s_out4[PAD / 4 + c];
        if (SKIP_DONE && s_done[((unsigned)(b_lo - (unsigned long long)e0 * FB) + 16u * (unsigned)c) / FB]) continue;
        if (CTX1) {                                       // frames are back to back: the chunk index IS the address
            u32x4 nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4 *>(obs4 + c_lo + c));
        } else {
            const unsigned long long gc = c_lo + c;
            const int e = (int)(gc / cpf), cc = (int)(gc - (unsigned long long)e * cpf);
            // (a finished env was stepped by this call: ring shift, whatever the reset beside us already wrote to fresh[])
            xw_store_chunk(obs4 + (size_t)e * p.context * cpf, cc, cpf, p.context, TERM && p.term_flag[e] ? 1 : p.fresh[e], val);
        }
    }
}

template <int DIM_T, int CH, bool CTX1, int BS, int PER, int RMODE, int ES>
__global__ __launch_bounds__(BS) void xw_render_all_kernel(XwParams p) {
    typedef RenderLds<CH, BS, PER, ES> L;
    __shared__ uint4 s_out4[L::OUT4];
    __shared__ uint16_t s_code[L::CODES];
    __shared__ uint8_t s_done[L::NE];
    // this kernel running = the step kernel queued before it is complete: tell the reset kernel's queue (xw_device.h)
    if (p.sig_epoch && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 1, p.sig_epoch);
    if (p.no_draw) return;                                 // (xwb_xw_set_draw(sim, 0): launched with one workgroup, for the epoch)
    xw_render_span<DIM_T, CH, CTX1, BS, PER, RMODE, ES, false>(p, blockIdx.x, s_out4, s_code, s_done);
}

// xwb_step of the default loop in ONE launch (XWB_PATH_LAZY_FUSED): blocks [0, step_blocks) step `epb` envs each (first
// wavefront; xw_step_body), every other block draws one span of the batch's frames from the look-ahead snapshot the previous
// step left (xw_render_span<SNAP>).  Nothing a render block reads is written by a step block of the same launch: the step
// blocks write the live state and the OTHER snapshot set.  The step's dependent round trips (8 us as a launch
// of its own, in front of a render that is bound by the chip's write stream) run beside the render's first workgroups.
constexpr int XW_FUSED_LDS = 8192;                         // bytes: a step block's grids (epb x cells x 2) or a span's staging
__host__ __device__ inline int xw_fused_epb(int max_dim) {
    const int per = max_dim * max_dim * 2;
    return XW_FUSED_LDS / per >= 64 ? 64 : (XW_FUSED_LDS / per >= 32 ? 32 : 16);
}

template <int DIM_T, int CH>
__global__ __launch_bounds__(128, 8) void xw_step_render_kernel(XwParams p, int step_blocks) {
    typedef RenderLds<CH, 128, 2, 1> L;
    static_assert((L::OUT4 * 16 + L::CODES * 2 + L::NE + 15) / 16 * 16 <= XW_FUSED_LDS, "a span's staging must fit");
    __shared__ uint4 s_mem[XW_FUSED_LDS / 16];
    if ((int)blockIdx.x < step_blocks) {
        if (threadIdx.x >= 64) return;
        xw_step_body(p, (int)blockIdx.x, xw_fused_epb(DIM_T ? DIM_T : p.max_dim), s_mem);
        return;
    }
    uint16_t *s_code = reinterpret_cast<uint16_t *>(s_mem + L::OUT4);
    xw_render_span<DIM_T, CH, true, 128, 2, 0, 1, true>(p, blockIdx.x - (unsigned)step_blocks, s_mem, s_code,
                                                          reinterpret_cast<uint8_t *>(s_code + L::CODES));
}

// the compacted list of freshly reset envs, tile table through L1/L2.  A frame is cut into `parts` pieces of at most 512
// chunks, one workgroup pass each: a lane owns at most two chunks and has the eight gathers of both in flight before its first
// store (one env per workgroup looped five times over load -> store: 10.8 us for the ~115 envs a C4 step finishes).  The chain
// of a workgroup: {count, list entry, its episode counter, epoch} -> grid -> gathers -> stores.
template <int DIM_T, int CH, int ES>
__global__ __launch_bounds__(256) void xw_render_list_kernel(XwParams p, const int32_t *count_now) {
    __shared__ uint16_t s_grid[XW_MAX_DIM * XW_MAX_DIM];
    const int D = DIM_T ? DIM_T : p.max_dim;
    const int cells = D * D;
    const int ctx = p.context;
    const int cpf = CH * 9 * cells * ES;
    const int parts = (cpf + 511) / 512, per = (cpf + parts - 1) / parts;
    // first round trip: the count, this workgroup's first list entry (the list is the step kernel's, complete long ago) and
    // the epoch, together
    SP_T(1, 0);
    // (after a fused step + render launch this kernel is the first one behind the step in the caller's queue: it tells the
    // internal queue, whose regeneration pass reads the done list, that the step is complete)
    if (p.sig_epoch && blockIdx.x == 0 && threadIdx.x == 0) xw_publish_epoch(p.sync + 1, p.sig_epoch);
    const int cnt = *count_now;
    const int i_first = (int)blockIdx.x / parts;
    const int e_first = p.done_list[i_first < p.n ? i_first : 0];
    const uint32_t ep_first = p.list_swap ? p.done_ep[i_first < p.n ? i_first : 0] : 0u;
    const long long items = (long long)cnt * parts;
    if ((long long)blockIdx.x >= items) { SP_T(1, 5); return; }   // nothing to draw (and nothing to wait for)
    SP_T(1, 1);
    // The listed envs were regenerated by a reset kernel on the other queue; its epoch stands in for a barrier packet.
    // Normally that kernel finished long ago.  When it has not, the spinning workgroups must not be able to fill the machine:
    // the kernels that publish the epoch need wave slots of their own.  render_list() therefore launches at most half the
    // machine's wave slots when a wait is attached (a batch whose envs all finish on one step otherwise parks one spinning
    // workgroup in every slot: seen as a 4 s stall on the 8x8 workload, where many envs time out on the same step).
    if (p.wait_epoch) xw_wait_epoch(p.sync + p.wait_slot, p.wait_epoch, p.sync + 4, p.poison_host);
    for (long long b = blockIdx.x; b < items; b += gridDim.x) {
        const int i = (int)(b / parts), part = (int)(b - (long long)i * parts);
        const bool first = b == (long long)blockIdx.x;
        const int e = first ? e_first : p.done_list[i];
        int la_axy = 0, la_act = 0;
        __syncthreads();
        if (p.list_swap) {
            // xwb_reset_done with pre-generated episodes: install the env's next episode (what the reset kernel made for it
            // beside an earlier render) -- the grid through the workgroup of the frame's first piece, the scalars through its
            // first thread -- and draw it.  Every piece reads the SHADOW grid (nothing rewrites it beside this kernel).
            const uint32_t ep_old = first ? ep_first : p.done_ep[i];
            const size_t es = (size_t)((ep_old + 1u) & 1u) * (size_t)p.n + (size_t)e;
            for (int k = threadIdx.x; k < cells; k += 256) {
                const uint16_t code = p.sh_grid[es * cells + k];
                if (part == 0) {
                    p.grid[(size_t)e * cells + k] = code;
                    if (p.snap_grid_out) p.snap_grid_out[(size_t)e * cells + k] = code;   // (the snapshot the next fused step draws from)
                }
                s_grid[k] = code & CELL_ICON_MASK;
            }
            if (part == 0 && threadIdx.x == 64) {         // (a lane of the second wavefront: beside the first one's gathers)
                la_axy = p.sh_agent_xy[es];
                // (look-ahead, below: the built-in policy's action of step p.policy_step, the step the caller runs next -- drawn
                // here, while the staging loads are in flight)
                if (p.snap_grid_out) la_act = policy_action(p.policy_seed, p.env_gid0 + (uint32_t)e, p.policy_step, 4);
                p.agent_xy[e] = la_axy;
                p.task_state[e] = p.sh_task_state[es];
                p.task_steps[e] = 0;
                if (p.n_tasks2 > 0) { p.task_state2[e] = p.sh_task_state2[es]; p.task_steps2[e] = 0; }
                p.sent_names[e] = p.sh_sent_names[es];
                p.cand2d[e] = p.sh_cand2d[es];
                reinterpret_cast<uint4 *>(p.goal_cells)[e] = reinterpret_cast<const uint4 *>(p.sh_goal_cells)[es];
                p.num_steps[e] = 0;
                p.episode[e] = ep_old + 1u;
                atomicAdd(p.perf + 36, 1ull);             // games reset
            }
        } else {
            for (int k = threadIdx.x; k < cells; k += 256) s_grid[k] = p.grid[(size_t)e * cells + k] & CELL_ICON_MASK;
        }
        __syncthreads();
        if (p.list_swap && p.snap_grid_out && part == 0 && threadIdx.x == 64) {
            // look-ahead: the snapshot row written above gets the next step's move of the new episode, beside the gathers
            int from;
            const int to = xw_predict_move(s_grid, D, la_axy, la_act, p.snap_act_rep, &from);
            if (to != from) { p.snap_grid_out[(size_t)e * cells + to] = s_grid[from]; p.snap_grid_out[(size_t)e * cells + from] = 0; }
        }
        uint4 *frame0 = reinterpret_cast<uint4 *>(p.obs) + (size_t)e * ctx * cpf;
        SP_T(1, 2);
        const int lo = part * per, hi = p.no_draw ? 0 : (lo + per < cpf ? lo + per : cpf);   // (drawing off: the install and the flags only)
        const int c0 = lo + threadIdx.x, c1 = c0 + 256;
        // (branch-free: a lane without a chunk gathers the piece's first one again and stores nothing)
        const uint4 v0 = xw_expand_chunk<DIM_T, CH, ES>(p.atlas, s_grid, c0 < hi ? c0 : lo, D);
        const uint4 v1 = xw_expand_chunk<DIM_T, CH, ES>(p.atlas, s_grid, c1 < hi ? c1 : lo, D);
        SP_T(1, 3);
        if (c0 < hi) xw_store_chunk(frame0, c0, cpf, ctx, p.list_flag, v0);
        if (c1 < hi) xw_store_chunk(frame0, c1, cpf, ctx, p.list_flag, v1);
        if (part == 0 && threadIdx.x == 0 && p.list_flag == 2) { p.fresh[e] = 0; if (p.auto_reset == 2) p.done[e] = 0; }
        SP_T(1, 4);
    }
}

// render_all launch shape: xwb_config.debug_render_shape overrides the default (A/B switch)
template <int DIM_T, int CH, int BS, int PER, int SKIP, int ES>
static hipError_t render_all_shape(const XwParams &p, hipStream_t s) {
    const unsigned long long n_chunks = (unsigned long long)p.n * (CH * 9 * ES * p.max_dim * p.max_dim);
    const unsigned blocks = p.no_draw ? 1u : (unsigned)((n_chunks + BS * PER - 1) / (BS * PER));
    if (p.context == 1) hipLaunchKernelGGL((xw_render_all_kernel<DIM_T, CH, true, BS, PER, SKIP, ES>), dim3(blocks), dim3(BS), 0, s, p);
    else hipLaunchKernelGGL((xw_render_all_kernel<DIM_T, CH, false, BS, PER, SKIP, ES>), dim3(blocks), dim3(BS), 0, s, p);
    return hipGetLastError();
}

template <int DIM_T, int CH, int SKIP, int ES>
static hipError_t render_all(const XwParams &p, hipStream_t s) {
    // measured on C4 / 8x8 / 11x11 (profiles/r1/render_shapes.txt): 128 x 2 is best everywhere (8 KiB spans, up to
    // 16 two-wave groups per CU); one chunk per lane leaves too few bytes per barrier, four too few groups in flight
    switch (p.dbg_render_shape) {                            // (xwb_config.debug_render_shape: A/B switch)
        case 1: return render_all_shape<DIM_T, CH, 64, 2, SKIP, ES>(p, s);
        case 2: return render_all_shape<DIM_T, CH, 256, 2, SKIP, ES>(p, s);
        default: return render_all_shape<DIM_T, CH, 128, 2, SKIP, ES>(p, s);
    }
}

template <int DIM_T, int CH>
static hipError_t step_render(const XwParams &p, hipStream_t s) {
    const unsigned long long n_chunks = (unsigned long long)p.n * (CH * 9 * p.max_dim * p.max_dim);
    const int epb = xw_fused_epb(p.max_dim);
    const int step_blocks = (p.n + epb - 1) / epb;
    const unsigned blocks = (unsigned)step_blocks + (unsigned)((n_chunks + 255) / 256);
    hipLaunchKernelGGL((xw_step_render_kernel<DIM_T, CH>), dim3(blocks), dim3(128), 0, s, p, step_blocks);
    return hipGetLastError();
}

hipError_t launch_xw_step_render(const XwParams &p, hipStream_t s) {
    if (p.visible_radius || p.obs_f32 || p.context != 1 || p.no_draw || p.actions || !p.snap_grid_in || !p.snap_grid_out) return hipErrorInvalidValue;
#define XW_CASE(DIMV) case DIMV: return p.channels == 3 ? step_render<DIMV, 3>(p, s) : step_render<DIMV, 1>(p, s);
    switch (p.max_dim) {
        XW_CASE(7) XW_CASE(8) XW_CASE(11)
        default: return p.channels == 3 ? step_render<0, 3>(p, s) : step_render<0, 1>(p, s);
    }
#undef XW_CASE
}

template <int DIM_T, int CH, int ES>
static hipError_t render_list(const XwParams &p, hipStream_t s) {
    // looping workgroups; a long list (a whole batch finishing together) keeps 8 per CU busy -- 4 per CU (half the wave
    // slots of the 256 CUs) when the kernel may have to spin on the other queue's epoch, see the kernel
    dim3 grid(p.wait_epoch ? 1024 : 2048), block(256);
    hipLaunchKernelGGL((xw_render_list_kernel<DIM_T, CH, ES>), grid, block, 0, s, p, (const int32_t *)p.done_count);
    return hipGetLastError();
}

template <int CH, int ES>
static hipError_t render_dispatch(const XwParams &p, int indexed, hipStream_t s) {
#define XW_CASE(DIMV) case DIMV: return indexed == 1 ? render_list<DIMV, CH, ES>(p, s) : (indexed == 2 ? render_all<DIMV, CH, 2, ES>(p, s) : (indexed == 3 ? render_all<DIMV, CH, 3, ES>(p, s) : render_all<DIMV, CH, 0, ES>(p, s)));
    switch (p.max_dim) {
        XW_CASE(7) XW_CASE(8) XW_CASE(11)
        default: return indexed == 1 ? render_list<0, CH, ES>(p, s) : (indexed == 2 ? render_all<0, CH, 2, ES>(p, s) : (indexed == 3 ? render_all<0, CH, 3, ES>(p, s) : render_all<0, CH, 0, ES>(p, s)));
    }
#undef XW_CASE
}

hipError_t launch_xw_render(const XwParams &p, int indexed, hipStream_t s, hipEvent_t ev_front, hipEvent_t ev_list, hipEvent_t ev_cells) {
    if (p.visible_radius) return launch_xw_render_ego(p, indexed == 3 ? 0 : indexed, s, ev_front, ev_list, ev_cells);
    if (p.obs_f32) return p.channels == 3 ? render_dispatch<3, 4>(p, indexed, s) : render_dispatch<1, 4>(p, indexed, s);
    return p.channels == 3 ? render_dispatch<3, 1>(p, indexed, s) : render_dispatch<1, 1>(p, indexed, s);
}

// The draw state of every env -- what a renderer elsewhere needs to reproduce the frames this batch shows (xwb_xw_pack_grids,
// include/xwb.h "gather the state, not the pixels"): the cell codes each env's CURRENT frame was drawn from (icon + 1, target
// bit stripped) and the context-ring operation of its last draw (xw_store_chunk's flag: 0 untouched, 1 ring shift, 2 fresh).
// src = what the last frame-drawing verb read: 0 the live grid with fresh[]; 1 xwb_step's terminal snapshots (render mode 3);
// 2 a list render (reset_done / reset_masked: the envs it drew are the ones at step 0).
__global__ __launch_bounds__(256) void xw_pack_grids_kernel(XwParams p, int src, uint16_t *out_grid, uint8_t *out_flag) {
    const int cells = p.max_dim * p.max_dim;
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= (size_t)p.n * cells) return;
    const int e = (int)(gi / cells);
    const bool term = src == 1 && p.term_flag[e];
    out_grid[gi] = (term ? p.term_grid[gi] : p.grid[gi]) & CELL_ICON_MASK;
    if (out_flag && gi == (size_t)e * cells) {
        const bool at_start = p.num_steps[e] == 0;
        out_flag[e] = (uint8_t)(src == 2 ? (at_start ? 2 : 0) : (term ? 1 : (at_start ? 2 : p.fresh[e])));
    }
}

hipError_t launch_xw_pack_grids(const XwParams &p, int src, uint16_t *out_grid, uint8_t *out_flag, hipStream_t s) {
    const size_t total = (size_t)p.n * p.max_dim * p.max_dim;
    hipLaunchKernelGGL(xw_pack_grids_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, src, out_grid, out_flag);
    return hipGetLastError();
}

hipError_t launch_xw_compact(const XwParams &p, int mode, hipStream_t s) {
    dim3 grid((p.n + 255) / 256), block(256);
    hipLaunchKernelGGL(xw_compact_kernel, grid, block, 0, s, p, mode, p.done_count);
    return hipGetLastError();
}

}  // namespace xwb

#ifdef XWB_STEP_PROF
extern "C" __attribute__((visibility("default"))) int xwb_debug_step_prof(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(xwb::g_step_prof), sizeof(unsigned long long) * 2 * 4096 * 6) == hipSuccess ? 0 : -1;
}
#endif
