// xwb_common.h -- shared declarations of libxwb.so (MI355X / gfx950 only).
//
// Kernel parameter blocks, the xwb-rng-v1 Philox stream (device side) and the
// launch entry points implemented by the per-game .hip files.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xwb {

// ---------------------------------------------------------------- RNG ------
// xwb-rng-v1: Philox4x32-10; key = (seed, global env id);
// counter = (block index, episode, stream id, 0); draws are successive words.
// stream 0 = reset decisions of that episode, stream 1 = built-in random policy
// (block index = rollout step).  DESIGN.md "xwb-rng-v1".
#ifndef XWB_PHILOX_ATTR
#define XWB_PHILOX_ATTR __forceinline__
#endif
__device__ XWB_PHILOX_ATTR uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return make_uint4(c0, c1, c2, c3);
}

struct Stream {
    uint32_t k0, k1, blk, episode, sid;
    uint32_t b0, b1, b2, b3;     // the words of the current block that are still to be handed out, next one in b0 (plain
                                 // scalars shifted down per draw: an indexed uint4 ends up in scratch memory)
    int have;
    // optional: the stream's first `npre` blocks, already computed (by the other lanes of a wavefront that resets ONE env:
    // Philox is a large part of that serial path's instructions and its blocks are independent of each other)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint4 __attribute__((address_space(3))) *lds_block_ptr;   // always LDS: ds_read instead of a flat load
#else
    typedef const uint4 *lds_block_ptr;                                     // (the host pass only parses this)
#endif
    lds_block_ptr pre;
    uint32_t npre;
    __device__ __forceinline__ void init(uint32_t seed, uint32_t gid, uint32_t ep, uint32_t stream_id) {
        k0 = seed; k1 = gid; blk = 0; episode = ep; sid = stream_id; have = 0; pre = nullptr; npre = 0;
        b0 = b1 = b2 = b3 = 0;
    }
    __device__ __forceinline__ uint32_t u32() {
        if (have == 0) {
            uint4 buf;
            if (blk < npre) buf = pre[blk];
            else buf = philox4x32_10(blk, episode, sid, 0u, k0, k1);
            b0 = buf.x; b1 = buf.y; b2 = buf.z; b3 = buf.w;
            blk += 1; have = 4;
        }
        const uint32_t v = b0;
        b0 = b1; b1 = b2; b2 = b3;
        have -= 1;
        return v;
    }
    // uniform in [0, n): multiply-shift.  Always consumes exactly one draw (also for n <= 1), so the
    // number of draws consumed up to any program point is the same for every lane of a wavefront and
    // the Philox refill branch stays wave-uniform.
    __device__ __forceinline__ uint32_t below(uint32_t n) {
        const uint32_t v = u32();
        return n <= 1 ? 0u : __umulhi(v, n);
    }
    // uniform float in [0, 1): 24 bits
    __device__ __forceinline__ float unit() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); }
};

__device__ __forceinline__ int policy_action(uint32_t policy_seed, uint32_t gid, uint32_t step, int num_actions) {
    const uint4 o = philox4x32_10(step, 0u, 1u, 0u, policy_seed, gid);
    return (int)__umulhi(o.x, (uint32_t)num_actions);
}

// GameOverCode bits (simulator.h:42-48)
enum : int { ALIVE = 0, MAX_STEP = 1, DEAD = 2, SUCCESS = 4, LOST_LIFE = 8 };

// XWB_ACTION_SKIP (include/xwb.h): the env does not take part in this step call
enum : int { ACTION_SKIP = -1 };

// which envs a state-changing kernel applies to
enum : int { MODE_STEP = 0, MODE_RESET_ALL = 1, MODE_RESET_DONE = 2, MODE_RESET_MASK = 3 };

// ------------------------------------------------------------ SimpleGame ---
struct SgParams {
    int n, array_size, context, max_steps, act_rep, mode, auto_reset, n_steps;
    uint32_t policy_seed, env_gid0, policy_step;
    const int32_t *actions;     // nullable
    const uint8_t *mask;        // MODE_RESET_MASK
    int32_t *actions_out;
    int32_t *pos;               // _cur_pos
    uint8_t *flags;             // bit0: rewards[0] consumed, bit1: rewards[N-1] consumed
    int32_t *num_steps;
    uint32_t *episode;
    float   *reward;
    float2 *packed;             // nullable: (reward, game_over code as float) of every stepped env, for a one-buffer exchange
    uint8_t *done;
    uint8_t *obs;               // [n][context][array_size]
    int32_t *err_count;
    int32_t *reset_partial;     // nullable (launches that reset nothing): [workgroups] envs this launch reset, per workgroup --
                                // plain stores, summed by xwb_done_count (one atomic per wavefront on a shared counter made
                                // the reset_done pass 13 us: L2 serialises same-address atomics, and under a random policy
                                // nearly every wavefront holds an env that just finished)
};
hipError_t launch_simple_game(const SgParams &p, hipStream_t s);

// ------------------------------------------------------------ SimpleRace ---
struct RaceParams {
    int n, context, max_steps, act_rep, mode, auto_reset, n_steps;
    uint32_t policy_seed, env_gid0, policy_step, seed;
    int track_type, random, difficulty_hard, n_legal;
    int legal[9];
    // track constants, computed on the host with the reference's float/double conversions
    float width, length;
    float mid_x, mid_y, start_x, start_y, end_x, end_y;          // StraightTrack
    float center_x, center_y, inner_radius, outer_radius;         // CircleTrack
    float delta_fwd, delta_ang;
    double reward_scale;
    const int32_t *actions;
    const uint8_t *mask;
    int32_t *actions_out;
    float *x, *y, *angle;
    int32_t *num_steps;
    uint32_t *episode;
    float *reward;
    float2 *packed;             // nullable: (reward, game_over code as float) of every stepped env, for a one-buffer exchange
    uint8_t *done;
    float *obs;                 // [n][context][4]
    int32_t *err_count;
    int32_t *reset_partial;     // see SgParams
    uint32_t *minstd;           // nullable: XWB_RNG_MINSTD, one libstdc++ minstd_rand0 state per env (include/xwb_minstd.h)
};
hipError_t launch_simple_race(const RaceParams &p, hipStream_t s);

// -------------------------------------------------------------- XWorld2D ---
constexpr int XW_MAX_DIM = 16;
constexpr int XW_MAX_GOALS = 16;
constexpr int XW_USAGE_BYTES = 32;  // one task class's success window
constexpr int XW_TILE = 12;           // block_size, xworld_simulator.cpp:57
constexpr int XW_TILE_DW = 3;         // dwords per tile row (12 bytes)

struct XwParams {
    int n, context, max_steps, act_rep, auto_reset;
    int map_kind, max_dim, dim, num_goals, num_blocks, max_steps_factor, task_mode, channels;
    int visible_radius;          // FLAGS_visible_radius: 0 = full observation; odd r > 0 = egocentric r x r view, 6 actions
    int out_dim;                 // egocentric frame edge: r * (84 / r)
    int obs_f32;                 // frames are float32 (pixel * 1/255), the tile table too
    int n_icons;
    int n_tasks, tasks[8];       // tasks of the teacher's group, sampled whenever the group is idle: uniformly, or
    int task_weighted;           // schedule "weighted": util::simple_importance_sampling over the accumulated weights
    double task_acc[8];
    // a second task group (conf order: after the first), run non-exclusively: Teacher::teach's else branch
    // (teacher.cpp:221-225).  n_tasks2 == 0: none.  One group holds the XWorld3DNav* tasks, the other the 2-D-native ones.
    int n_tasks2, tasks2[8], task_weighted2, group2d_2;
    double task_acc2[8];
    int32_t *task_state2, *task_steps2;   // [n] the second group's Task FSM (same encoding as task_state / task_steps)
    // Teacher::teach's exclusive branch (teacher.cpp:209-220; FLAGS_task_groups_exclusive outside lang_acquisition): every
    // teach() re-sorts the groups by weighted sampling without replacement (nondeterministic_sort_task_groups, :143-163),
    // then ONE group runs: the last busy one of that order, else its first.
    int exclusive;
    double group_weight[2];      // the conf's per-group "weight" keys, conf order
    uint8_t *grp_order;          // [n] bit 0: conf index of the group that heads Teacher::task_groups_ (the sort is in place
                                 // and the list lives as long as the teacher: across resets); bit 1: the group the last teach() ran
    int32_t *idle_list;          // two groups, exclusive: envs whose IDLE XWorld3DNav* group this step picked -- its
    int32_t *idle_count;         // map-rearranging idle stage runs right behind the step kernel (xw_idle3d_kernel);
    int32_t *idle_count_next;    // counters double-buffered like done_count
    int list_flag;               // list render: 2 = first frame of a new episode (init_screen: older context frames zeroed,
                                 // fresh / done flags cleared); 1 = the terminal frame of a finished env (ring shift only)
    int group2d;                 // the group holds the 2-D-native tasks (rule D14b): idle stages also run at step time
    uint32_t policy_seed, env_gid0, policy_step, seed;
    // icon tables (device)
    const uint8_t *icon_type;    // [n_icons]
    const int16_t *icon_name;    // [n_icons]
    const int16_t *name_first;   // [3][max_names+1] flattened: offsets into name_variants, per type
    const int16_t *name_variants;
    int n_names[3];
    int name_first_off[3];       // start of each type's offset table inside name_first
    int name_first_len, name_variants_len;
    const uint32_t *atlas;       // [n_icons + 1][channels][12][3] dwords (tile table; entry 0 = empty cell), or
                                 // [n_icons + 1][channels][12][12] floats when obs_f32
    const int32_t *actions;
    const uint8_t *mask;
    int32_t *actions_out;
    // state
    uint16_t *grid;              // [n][max_dim*max_dim] cell code = icon + 1 (0 empty) | bit 15: target goal
    int32_t *agent_xy;           // x | y << 16
    int32_t *task_steps;         // steps_in_cur_task
    uint8_t *goal_cells;         // [n][XW_MAX_GOALS] cell of goal slot i (entity order), 0xff = none
    uint16_t *term_grid;         // [n][max_dim*max_dim] the grid of an env at the step that finished it (see term_flag)
    uint8_t *term_flag;          // [n] 1: this step finished the env; its terminal frame is rendered from term_grid, so a
                                 //     reset_done running beside the render may already regenerate the live grid
    double curriculum;           // FLAGS_curriculum (0 = off; XWorldNav only)
    uint8_t *cur_level;          // [n] curriculum: XWorldEnv.current_level
    int32_t *cur_counter;        // [n] curriculum: XWorldEnv.curriculum_check_counter
    uint8_t *cur_usage;          // [n][9][XW_USAGE_BYTES] curriculum: per task class the window of its last 200 results
                                 //     (len, sum, head, -, 25 bytes of bits; xw_device.h usage_push)
    uint32_t *sent_names;        // [n] goal-name ids the idle stage binds into the teacher's sentence: a | b << 16 (0xffff none)
    uint8_t *agent_dir;          // [n] egocentric heading: 0 right, 1 down, 2 left, 3 up (XItem::get_item_facing_dir)
    double *goal_warp;           // [n][XW_MAX_GOALS][6] egocentric: inverse affine map of the goal's icon warp
    const uint8_t *atlas64;      // egocentric: 4 bytes per pixel (B, G, R, 0): [n_icons][64][64] item images (XItem::item_size_
                                 // = 64), one white and one black pixel, then the turned copies of the agent icons
    const uint32_t *ego_agent_rot;   // egocentric: [n_icons] pixel offset in atlas64 of an agent icon's three turned copies
                                 // (heading right, left, up; heading down is the icon itself)
    uint32_t *goal_img;          // egocentric: [n][num_goals][64 * 64] warped goal images (B | G << 8 | R << 16)
    const void *ego_taps;        // egocentric: cv::resize taps of the two resizes, then the four headings' layout tables
    int no_wall_shadow;          // FLAGS_wall_shadow = false: nothing is blacked out behind walls
    int ego_list_beside;         // egocentric list render: launched beside the big render -> small workgroups that fit into
                                 // the slots it frees (a 1024-thread group needs a whole idle CU and would wait for the end)
    int ego_fast;                // egocentric: interior pixels can be copied from ego_tab (kernels_xworld_ego.hip)
    const uint8_t *ego_tab;      // egocentric: [(n_icons + 2) * 4] frames "every cell shows icon i", per heading (interior pixels)
    // egocentric: rendered goal cells, filled lazily.  The interior pixels of a view cell that shows a goal depend only on
    // the goal's warped image (fixed for the episode), the cell's place in the view and the heading: entry
    // [env][goal slot][view cell][heading] = those pixels ([channel][row][col], ego_cache_entry bytes), valid when its bit
    // in ego_cache_valid ([env][ego_cache_words]) is set; the warp kernel clears an env's bits when its goals get new poses.
    uint8_t *ego_cache;          // nullable (not enough free memory: every goal cell is evaluated every frame)
    uint32_t *ego_cache_valid;
    uint32_t ego_cache_entry, ego_cache_words;
    // egocentric, span path (kernels_xworld_ego.hip: cells -> misses -> gather), nullable / 0 when the geometry rules it out
    int ego_span;                // the view cells' pixel rectangles tile the frame in equal squares (no straddling rows / columns)
    uint32_t *ego_cellinfo;      // [n][r * r] what each square of the frame shows (xw_ego_cells_kernel has the bit layout)
    const uint8_t *ego_cls;      // [n_icons + 2] dense index of the images that are the same in every env (blocks, agents, an
                                 //     empty cell, a black one), 0xff: a goal;  ego_cls_icon [ego_ncls]: class -> table slot
    const uint16_t *ego_cls_icon;
    int ego_ncls;
    const uint8_t *ego_tab3;     // [heading][class c][class a][class l][channel][square] squares (kernels_xworld_ego.hip, EgoSq)
    const uint8_t *ego_flat;     // [heading][c][a][l][square]: that entry of ego_tab3 is one flat colour: 1 = 255 (empty cells), 2 = 0 (outside
                                 //     the map, shadow); found by scanning the table once.  The gather reads such squares from ONE shared
                                 //     128-byte line (ego_constline: 128 x 0xff, 128 x 0x00), which stays in every CU's L1 -- two thirds of a
                                 //     frame's squares, whose table lines otherwise each miss to L2 (the gather was bound by L1 miss handling)
    const uint8_t *ego_constline;
    uint2 *ego_cellsrc;          // [n][r * r] per square: where the gather finds its pixels, and what it patches in from where
                                 //     (xw_ego_cells_kernel has the bit layout of both words)
    uint2 *ego_miss;             // goal cells the cache does not hold yet: (env, view cell | slot << 8 | heading << 16)
    int32_t *ego_miss_count;
    uint2 *ego_cellsrc_list;     // the same three for the done-list render, which runs beside the whole-batch gather
    uint2 *ego_miss_list;
    int32_t *ego_miss_count_list;
    const uint32_t *ego_xtab;    // [heading][c][a][l][d][square]: the pixel where a border row crosses a border column (B | G << 8 | R << 16, or the
                                 //     gray value), when the four cells around it show classes c (the square's own), a (above), l (left), d (above left)
    const uint2 *ego_clsimg;     // [heading][class]: (pixel offset of the class's image in atlas64, index mask: -1 an image, 0 one constant pixel)
    uint32_t *cand2d;            // [n] goal slots the agent can reach, blocks as the only obstacles: bits 0..15
                                 //     any goal (XWorldNavTarget), bits 16..31 coloured goals (XWorldNavColorTarget)
    const uint8_t *icon_colored; // [n_icons] properties.txt colour != "na"
    int32_t *task_state;         // target (low 16) | stage << 16 | event << 20 | task << 24  (xw_device.h)
    int32_t *num_steps;
    uint32_t *episode;
    uint8_t *success;            // last_action_success
    uint8_t *fresh;              // set by reset, consumed by render (context ring init)
    float *reward;
    float2 *packed;             // nullable: (reward, game_over code as float) of every stepped env, for a one-buffer exchange
    uint8_t *done;
    uint8_t *obs;                // [n][context][channels][12*max_dim][12*max_dim]
    int32_t *done_list;          // compacted env ids
    uint32_t *done_ep;           // lazy path (swap_shadow == 2): episode counter of each listed env at the step that listed it
    int32_t *done_count;         // counter the current step / compaction appends to
    int32_t *done_count_next;    // the other one of the pair; zeroed by the step kernel
    int32_t *err_count;
    // what Teacher::report_task_performance adds up (teacher.cpp:175-200; teaching_task.h:22-36 BenchmarkRes), for the whole
    // batch since it was created: perf[task class][0..3] = successes, failures, success_steps, time-ups (a subset of the
    // failures); perf[36] = games reset.  Bumped with atomics by the few lanes that record a result (~0.3 % of a step's envs).
    unsigned long long *perf;
    // Pre-generated next episodes (full observation, no curriculum / minstd / exclusive scheduling: the next episode of an env is
    // then a pure function of (seed, global env id, episode + 1)).  The reset kernel run with `shadow` set writes the episode
    // after the newest one env e already holds (sh_ep[e] counts them) into slot (episode & 1) of the sh_* arrays ([2][n]; the
    // host swaps them in for grid, agent_xy, ...; live counters and flags are left alone).  Who installs a shadow:
    //   swap_shadow = 1 (xwb_step_autoreset)  the step kernel, for the envs it finishes: the reset and its first-frame render
    //                  leave the critical path, the step's one render draws every env;
    //   list_swap (xwb_reset_done after a step run with swap_shadow = 2)  the list render, right before it draws the first
    //                  frame: the step then keeps no terminal snapshot (its render reads the live grid, which nothing
    //                  rewrites beside it) -- 3 us off the step kernel, 2 off the render on C4.
    // Regeneration runs on the side queue beside the render, two slots per env so that it never writes what an installer may
    // still read; whoever touches the done list next (the step kernel's wavefronts that hold a finished env, the thread that
    // zeroes the rotating counter, the installing list render) first waits, device-side, for it: sync[8] >= regen_wait.
    int shadow, swap_shadow, list_swap;
    uint32_t regen_wait;
    uint32_t *sh_ep;
    uint16_t *sh_grid;
    int32_t *sh_agent_xy, *sh_task_state, *sh_task_state2;
    uint32_t *sh_sent_names, *sh_cand2d;
    uint8_t *sh_goal_cells;
    // Look-ahead snapshots for the fused step + render launch (xw_step_render_kernel, the default loop's xwb_step under the built-in
    // policy): a step changes at most two cells of an env's grid (XMap::move_item, xmap.cpp:76-101), whether it does is a function
    // of (grid, agent cell, action) alone, and the built-in policy's action of step t + 1 is a function of (seed, env, t + 1).  A
    // lazy step t therefore also writes, into one of two snapshot sets (snap_grid_out), every env's grid AS STEP t + 1 WILL LEAVE
    // IT; the installing list render does the same for the episodes it starts.  xwb_step t + 1 is then ONE launch: its render
    // blocks draw from that snapshot (snap_grid_in) -- the plain whole-batch render, reading another array -- while its step
    // blocks update the live state and fill the other set beside them.  Null on every other path.  (Deriving the moved cells
    // inside the render blocks from a descriptor + the call's actions was built first: every extra load, vector or scalar, at the
    // head of the 85 000 workgroups cost more than the step kernel it saved -- profiles/NOTES.md round 6.)
    const uint16_t *snap_grid_in;
    uint16_t *snap_grid_out;
    int snap_act_rep;            // act_rep of the predicted move (the list render's; the step kernel predicts with its own)
    // device-side hand-off between the two queues of the step loop, instead of event / barrier packets (each costs the
    // loop ~3-6 us of idle GPU): sync[1] = epoch of the last completed step kernel, sync[3] = of the last completed reset
    // kernel, sync[4] != 0: a wait gave up (xw_device.h: xw_publish_epoch / xw_wait_epoch).  render_all with sig_epoch != 0 publishes it to sync[1] when it
    // starts (= the step kernel before it in the queue is complete); the list render with wait_epoch != 0 waits for sync[3].
    uint32_t *sync;
    uint32_t *poison_host;       // pinned host word raised together with sync[4] when a wait's watchdog expires (the host reads
                                 // it at the top of every verb without a sync: the batch is poisoned from then on)
    uint32_t sig_epoch, wait_epoch;
    int wait_slot;               // the list render's wait: sync[wait_slot] >= wait_epoch (3: the reset kernel's epoch, 8: the regeneration's)
    uint32_t *minstd;            // nullable: XWB_RNG_MINSTD, one libstdc++ minstd_rand0 state per env: the teacher's task draw
    int dbg_ego_per, dbg_ego_pad, dbg_render_shape;   // xwb_config.debug_* (launch-shape A/B switches; 0 = defaults)
    int dbg_ego_miss_blocks;     // XWB_DEBUG ego_miss_blocks=N: goal-cell workgroups of the whole-batch evaluation launch (0 = the default)
                                 // (lab: frames wrong next to goals), bits 4.. its launch shape
    int no_draw;                 // xwb_xw_set_draw(sim, 0): the renders keep their bookkeeping (epochs, installs, fresh / done flags) and store no pixels
};
hipError_t launch_xw_step(const XwParams &p, hipStream_t s);
// step + render(all) in ONE launch: the render blocks draw from the snapshot p.snap_*_in + this call's actions (uint8 frames,
// context 1, full observation); hipErrorInvalidValue when the configuration has no such kernel
hipError_t launch_xw_step_render(const XwParams &p, hipStream_t s);
// exclusive scheduling of two groups: the idle stages of the XWorld3DNav* group that the step kernel deferred (idle_list)
hipError_t launch_xw_idle3d(const XwParams &p, hipStream_t s);
// one wavefront that ends once *epoch_slot has reached `want`: orders the work queued behind it after the publisher
// (budget_ticks: watchdog in 100 MHz ticks, 0 = the default 4 s; poison / poison_host: xw_device.h xw_wait_epoch)
hipError_t launch_xw_wait(const uint32_t *epoch_slot, uint32_t want, uint32_t *poison, uint32_t *poison_host, hipStream_t s,
                          unsigned long long budget_ticks = 0);
// one thread that publishes `value`: queued behind a kernel, it tells the other queue that kernel is complete
hipError_t launch_xw_signal(uint32_t *epoch_slot, uint32_t value, hipStream_t s);
// reset envs: mode RESET_ALL -> every env; otherwise the compacted done_list / done_count
// before_warp (egocentric): the redraw of the goal images waits for it (kernels still reading the old images);
// defer_warp: the goal images are left to the caller (launch_xw_render(p, 7, ...): redrawn beside the list's cell tables)
hipError_t launch_xw_reset(const XwParams &p, int mode, hipStream_t s, hipEvent_t before_warp = nullptr, const uint32_t *warp_epoch_slot = nullptr,
                           uint32_t warp_epoch = 0, int defer_warp = 0);
// compaction of done[] (mode RESET_DONE) or mask (RESET_MASK) into done_list / done_count
hipError_t launch_xw_compact(const XwParams &p, int mode, hipStream_t s);
// the batch's draw state (cell codes its current frames show + context-ring flags) for a renderer elsewhere; src: see the kernel
hipError_t launch_xw_pack_grids(const XwParams &p, int src, uint16_t *out_grid, uint8_t *out_flag, hipStream_t s);
// render: indexed == 0 -> all envs (LDS-resident atlas, persistent workgroups);
//         indexed == 1 -> envs in done_list (atlas through L2)
// indexed: 0 = every env, 1 = the compacted done list, 2 = every env whose done code is 0 (the rest follows as a list),
// 3 = every env, those the step just finished from their terminal snapshot (term_grid)
hipError_t launch_xw_render(const XwParams &p, int indexed, hipStream_t s, hipEvent_t ev_front = nullptr, hipEvent_t ev_list = nullptr, hipEvent_t ev_cells = nullptr);
// egocentric: indexed 4 = a step's frames on the span path (kernels_xworld_ego.hip), see ego_span_render; the list render of the
// span path in parts (ego_span_render_list): 5 = its two front kernels, 6 = its gather, 7 = as 5 with the listed envs' goal images
// redrawn in the first launch (launch_xw_reset's defer_warp), 8 = everything, with that first launch
hipError_t launch_xw_render_ego(const XwParams &p, int indexed, hipStream_t s, hipEvent_t ev_front = nullptr, hipEvent_t ev_list = nullptr, hipEvent_t ev_cells = nullptr);
bool xw_ego_span(const XwParams &p);
hipError_t launch_xw_warp_goals(const XwParams &p, bool list, hipStream_t s);
struct EgoTap;
hipError_t xw_ego_tables(int r, int max_dim, int out_dim, EgoTap **dev_out, int *fast_out, int *cell_edge_out, int *span_out);
size_t xw_ego_cache_entry_bytes(const XwParams &p, int cell_edge);
size_t xw_ego_tab_bytes(const XwParams &p);
hipError_t launch_xw_ego_build_tab(const XwParams &p, hipStream_t s);
size_t xw_ego_square_tab_bytes(const XwParams &p);
size_t xw_ego_square_entry_bytes(const XwParams &p);
size_t xw_ego_xtab_bytes(const XwParams &p);
hipError_t launch_xw_ego_build_squares(const XwParams &p, hipStream_t s);

// host: builds the 12x12 tile table (OpenCV 3.2 fixed-point bilinear + BGR2GRAY) from 64x64 icons
void build_tile_table(const uint8_t *icons64, int n_icons, int channels, uint8_t *out /* n*c*12*12 */);

}  // namespace xwb
