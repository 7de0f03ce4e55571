// xw_device.h -- small device helpers shared by the XWorld2D kernels.
#pragma once
#include "xwb_common.h"
#include "../../include/xwb_minstd.h"

namespace xwb {

enum : int { STAGE_IDLE = 0, STAGE_NAV = 1, STAGE_TERMINAL = 2 };
enum : int { EV_NONE = 0, EV_CORRECT = 1, EV_WRONG = 2, EV_TIMEUP = 3 };

// tasks of the XWorld3DNav group (confs/navigation2d.json order) and the direction words of NavTargetDirection
enum : int { TASK_TARGET = 0, TASK_NEAR = 1, TASK_BETWEEN = 2, TASK_DIRECTION = 3, TASK_AVOID = 4,
              // the 2-D-native group "XWorldNav" of confs/walls.json (games/xworld/tasks/XWorldNav*.py, rule D14b)
              TASK2D_TARGET = 5, TASK2D_NEAR = 6, TASK2D_COLOR = 7, TASK2D_BETWEEN = 8 };
enum : int { DIR_FRONT = 1, DIR_BEHIND = 2, DIR_LEFT = 3, DIR_RIGHT = 4 };

// task_state word: target (name id for NavTarget, middle cell for NavTargetBetween, target cell for the
// 2-D-native tasks, else -1) | stage | event | task
__device__ __forceinline__ int pack_task(int target, int stage, int event, int kind) {
    return (target & 0xffff) | (stage << 16) | (event << 20) | (kind << 24);
}
__device__ __forceinline__ int task_target(int ts) { return (int)(int16_t)(ts & 0xffff); }
__device__ __forceinline__ int task_stage(int ts) { return (ts >> 16) & 0xf; }
__device__ __forceinline__ int task_event(int ts) { return (ts >> 20) & 0xf; }
__device__ __forceinline__ int task_kind(int ts) { return (ts >> 24) & 0xf; }

// cell code: bits 0..14 = palette icon + 1 (0 = empty), bit 15 = the goal belongs to the teacher's target set
constexpr uint32_t CELL_ICON_MASK = 0x7fffu, CELL_TARGET_BIT = 0x8000u;

// The idle stage of a 2-D-native task (XWorldNavTarget.py:22-33, XWorldNavColorTarget.py:8-20; Near / Between never
// find a target in this snapshot): Task::reset, then a uniformly chosen reachable [coloured] goal.
// `draw(n)` supplies the decisions (reset stream at reset time, stream 2 / block = num_steps at step time).
template <typename Draw>
__device__ __forceinline__ void idle_2d(int kind, uint32_t cand, const uint8_t *goal_cells, Draw draw,
                                        int &target, int &stage, int &tsteps) {
    tsteps = 0;
    target = -1;
    stage = STAGE_IDLE;
    if (kind != TASK2D_TARGET && kind != TASK2D_COLOR) return;
    const uint32_t m = kind == TASK2D_TARGET ? (cand & 0xffffu) : (cand >> 16);
    const int nc = __popc(m);
    if (nc == 0) return;
    int k = (int)draw((uint32_t)nc);                       // random.choice(targets)
    uint32_t mm = m;
    while (k-- > 0) mm &= mm - 1;                          // drop the k lowest set bits
    target = goal_cells[__ffs(mm) - 1];
    stage = STAGE_NAV;
}

// p.tasks[i] for a per-lane i: a select chain over the eight scalars instead of a vector load from the kernel-argument
// buffer (which is not cached like device memory: a full memory round trip in the middle of a latency-bound kernel)
template <int G = 0>
__device__ __forceinline__ int task_at(const XwParams &p, int i) {
    const int *tasks = G ? p.tasks2 : p.tasks;
    int r = tasks[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = i == k ? tasks[k] : r;
    return r;
}

// Epochs in device memory order the two queues of the step loop without event / barrier packets.  The publisher is always
// the FIRST thread of the kernel that FOLLOWS the producing kernel in its in-order queue: when that kernel starts, the
// producer has completed and the queue's kernel-boundary release / acquire has made its writes visible device-wide, so
// the store needs no fence of its own (a per-workgroup __threadfence() in the producer was measured: +9 us on a 6 us
// kernel).  Pollers use relaxed device-scope loads (no cache invalidate per iteration) and one acquire fence at the end.
__device__ __forceinline__ void xw_publish_epoch(uint32_t *epoch_slot, uint32_t value) {
    __hip_atomic_store(epoch_slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait (one lane spins, the workgroup follows through the barrier) until *epoch_slot has reached `want` (wrap-safe).
// Host-side invariant (xwb_verbs.hip): the kernel that publishes an epoch is ALWAYS enqueued before the kernel that waits for
// it, so two streams that share one in-order hardware queue (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues), or a tool
// that serialises kernels in submission order, run publisher-then-waiter and the loop never spins; only when the queues run
// concurrently does the waiter poll, and then its publisher is already in flight.  Each (batch, caller stream) pair is also
// probed once for real concurrency before epochs are used on it (epoch_selftest), events being the fallback.
// Watchdog: a spin that still has not been released after `budget` ticks of the 100 MHz wall clock (default 4 s) POISONS the
// batch -- *poison (device) and *poison_host (pinned host memory the library reads without a sync) are raised, every later
// verb of the batch fails with XWB_ERR_STATE, and every later wait returns at once so the queues drain instead of hanging.
constexpr unsigned long long XW_WATCHDOG_TICKS = 400000000ull;
__device__ __forceinline__ void xw_wait_epoch(const uint32_t *epoch_slot, uint32_t want, uint32_t *poison, uint32_t *poison_host,
                                              unsigned long long budget = XW_WATCHDOG_TICKS) {
    if (threadIdx.x == 0) {
        if ((int32_t)(__hip_atomic_load(epoch_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 &&
            __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned long long t0 = wall_clock64();                  // 100 MHz
            while ((int32_t)(__hip_atomic_load(epoch_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > budget) {
                    atomicExch(poison, 1u);
                    if (poison_host) __hip_atomic_store(poison_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// the same wait for ONE lane inside a running kernel (no barrier): the other lanes of its wavefront wait with it
__device__ __forceinline__ void xw_wait_epoch_lane(const uint32_t *epoch_slot, uint32_t want, uint32_t *poison, uint32_t *poison_host) {
    if ((int32_t)(__hip_atomic_load(epoch_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 &&
        __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(epoch_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > XW_WATCHDOG_TICKS) {
                atomicExch(poison, 1u);
                if (poison_host) __hip_atomic_store(poison_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
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


// TaskGroup::run_stage's sample_task (teaching_task.cpp:204-213): util::get_rand_ind, or for the "weighted" schedule
// util::simple_importance_sampling (simulator_util.cpp:57-86): a float uniform in [0, float(total)), first task whose
// accumulated weight is >= it.  One draw either way.
// XWB_RNG_MINSTD: the decision comes from env e's own minstd_rand0 (the reference's thread-local engine); the stream's draw
// is still consumed so that everything else the stream decides (the map) does not depend on the RNG mode.
template <int G = 0, typename S>
__device__ inline int sample_task(const XwParams &p, S &s, int e) {
    const int n_conf = G ? p.n_tasks2 : p.n_tasks;
    const int n = n_conf > 0 ? n_conf : 1;
    const double *task_acc = G ? p.task_acc2 : p.task_acc;
    if (!(G ? p.task_weighted2 : p.task_weighted)) {
        int t = (int)s.below((uint32_t)n);
        if (p.minstd) { uint32_t x = p.minstd[e]; t = xwb_minstd_rand_ind_state(&x, n); p.minstd[e] = x; }
        return t;
    }
    float val = s.unit() * (float)task_acc[n - 1];
    if (p.minstd) { uint32_t x = p.minstd[e]; val = xwb_minstd_rand_range_state(&x, (float)task_acc[n - 1]); p.minstd[e] = x; }
    const double w = (double)val;
    for (int i = 0; i < n; ++i) if (w <= task_acc[i]) return i;
    return n - 1;
}

// Teacher::nondeterministic_sort_task_groups (teacher.cpp:143-163) for two groups: position 0 takes one of the two with
// probability proportional to its weight -- util::simple_importance_sampling: a float uniform in [0, float(w_a + w_b)), the
// first accumulated weight >= it --, position 1 draws over the one weight that is left (index 0 whatever the value: only
// the reference's engine notices).  `first` = conf index of the group heading the list before the call; returns the one
// heading it afterwards.  Decisions ("xwb-taskgen-v1"): stream 4, block = num_steps of the teach() call (0 at reset),
// word 0; XWB_RNG_MINSTD: two draws of the env's own engine, as in the reference.
__device__ inline int xw_sort_groups(const XwParams &p, int e, uint32_t episode, uint32_t steps, int first) {
    const double wa = first ? p.group_weight[1] : p.group_weight[0], wb = first ? p.group_weight[0] : p.group_weight[1];
    const double total = wa + wb;
    const uint4 o = philox4x32_10(steps, episode, 4u, 0u, p.seed, p.env_gid0 + (uint32_t)e);
    float val = (float)(o.x >> 8) * (1.0f / 16777216.0f) * (float)total;
    int idx;
    if (p.minstd) {
        uint32_t x = p.minstd[e];
        val = xwb_minstd_rand_range_state(&x, (float)total);
        idx = (double)val <= wa ? 0 : 1;
        (void)xwb_minstd_rand_range_state(&x, (float)(idx ? wa : wb));
        p.minstd[e] = x;
    } else {
        idx = (double)val <= wa ? 0 : 1;
    }
    return idx ? first ^ 1 : first;
}

// ---- curriculum (FLAGS_curriculum != 0) ----
// XWorld(3D)Task.__record_result (xworld3d_task.py:129-133, xworld_task.py:87-91): success_seq.append(res), the oldest of
// more than performance_window_size = 200 dropped.  One window: len, sum, head, pad, 200 bits.
__device__ inline void usage_push(uint8_t *u, int res) {
    int len = u[0], sum = u[1], head = u[2];
    uint8_t *bits = u + 4;
    auto get = [&](int i) { return (bits[i >> 3] >> (i & 7)) & 1; };
    auto put = [&](int i, int v) { bits[i >> 3] = (uint8_t)((bits[i >> 3] & ~(1 << (i & 7))) | (v << (i & 7))); };
    if (len < 200) {
        put((head + len) % 200, res);
        len++; sum += res;
    } else {
        sum += res - get(head);
        put(head, res);
        head = (head + 1) % 200;
    }
    u[0] = (uint8_t)len; u[1] = (uint8_t)sum; u[2] = (uint8_t)head;
}

// XWorldNav._configure with curriculum != 0 (XWorldNav.py:27-55) + XWorldEnv.get_current_usage (xworld_env.py:103-110):
// every 100th reset that finds a task with results compares the worst task's success rate over its window with the
// flag and moves to the next of the six levels.  Returns the level of the episode that is being set up.
__device__ inline int curriculum_configure(const XwParams &p, int e) {
    int counter = p.cur_counter[e] + 1;
    const uint8_t *u = p.cur_usage + (size_t)e * 9 * XW_USAGE_BYTES;
    double usage = 0;
    bool any = false;
    for (int k = 0; k < 9; ++k) any = any || u[k * XW_USAGE_BYTES] > 0;
    if (counter >= 100 && any) {
        usage = 2;
        for (int k = 0; k < 9; ++k) {
            const int len = u[k * XW_USAGE_BYTES], sum = u[k * XW_USAGE_BYTES + 1];
            if (len > 0) { const double r = (double)sum / (double)len; if (r < usage) usage = r; }
        }
        counter = 0;
    }
    p.cur_counter[e] = counter;
    int level = p.cur_level[e];
    if (usage >= p.curriculum && level < 5) level++;
    p.cur_level[e] = (uint8_t)level;
    return level;
}

// The move XAgent::act x act_rep (xitem.cpp:89-101) + XMap::move_item (xmap.cpp:76-101) makes on the cell codes `lg` of one env
// under full observation: the agent's code goes up to act_rep cells along action a (MOVE_UP, MOVE_DOWN, MOVE_LEFT, MOVE_RIGHT)
// while the next cell is inside the map and empty.  Returns the agent's new cell; *from = its old one (equal: no move).
__device__ __forceinline__ int xw_predict_move(const uint16_t *lg, int D, int axy, int a, int act_rep, int *from) {
    int ax = axy & 0xffff, ay = axy >> 16;
    *from = ay * D + ax;
    const int ddx = a == 2 ? -1 : (a == 3 ? 1 : 0), ddy = a == 0 ? -1 : (a == 1 ? 1 : 0);
    for (int i = 0; i < act_rep; ++i) {
        const int tx = ax + ddx, ty = ay + ddy;
        if (tx < 0 || ty < 0 || tx >= D || ty >= D || lg[ty * D + tx] != 0) break;      // (blocked once = blocked for good)
        ax = tx; ay = ty;
    }
    return ay * D + ax;
}

// ---- shared by the render kernels ----
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

// all envs: ONE-SHOT workgroups in dispatch order -- the store structure that reaches the write ceiling on this
// chip (tools/render_lab.hip: one-shot 6.7 TB/s, every persistent / looping structure <= 5.7 TB/s; the persistent
// LDS-table kernel this replaces ran at 4.4 TB/s).  Each workgroup owns SPAN = BS * PER consecutive 16-byte chunks
// of the batch's frame bytes, cut at 1 KiB multiples of the global chunk index so every wavefront store is a whole
// number of cache lines although env frames (7x7x3: 21 168 B) are not 128-byte aligned.  A frame row is a run of
// 12-byte tile rows, so the span is assembled in LDS in OUTPUT order from 12-byte rows gathered from the tile table
// through L2 (157 KB, resident in every XCD's L2; one 12-byte load per tile row -- a per-dword gather is TA-bound at
// 3.2 TB/s) and leaves as one 16-byte non-temporal store per lane.  All of a lane's gathers are issued before its
// first LDS write (loads-first: 119 -> 111 us on C4 in the lab; 104 us = 6.7 TB/s inside the step loop).

}  // namespace xwb
