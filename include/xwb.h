/*
 * include/xwb.h -- C ABI of libxwb.so, the MI355X-native batched XWorld simulator.
 *
 * The reference (PaddlePaddle/XWorld) has no C ABI / FFI: its plugin boundary is
 * the C++ class simulator::GameSimulator (simulator.h:52-231) behind the facade
 * simulator::SimulatorInterface (simulator_interface.h:40-89), exported to Python
 * by the Boost.Python module py_simulator (python/py_simulator.cpp:310-329).
 * Each entry point below states which reference interface it replaces; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - one xwb_sim = one *batch* of num_envs independent environments resident on
 *     one GPU (the reference: one object = one environment);
 *   - all per-env state lives in HBM as structure-of-arrays, env index fastest;
 *   - pointers named *_dev are device pointers; `stream` is a hipStream_t passed
 *     as void* (NULL = the default stream).  Calls are asynchronous on `stream`
 *     unless stated otherwise;
 *   - every function returns XWB_OK (0) or a negative error code; the message of
 *     the last error on the calling thread is xwb_last_error().  The reference
 *     aborts the process (CHECK / LOG(FATAL)) where this ABI returns an error;
 *   - there is no CPU fallback: xwb_create fails with XWB_ERR_HIP when no gfx950
 *     device is usable.
 */
#ifndef XWB_H
#define XWB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libxwb.so is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end of
 * the file are exported (tests/test_abi_and_host.py compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define XWB_ABI_VERSION 5

enum {
    XWB_OK = 0,
    XWB_ERR_ARG = -1,       /* bad argument / unsupported configuration */
    XWB_ERR_HIP = -2,       /* HIP runtime failure (message has hipGetErrorString) */
    XWB_ERR_STATE = -3,     /* call not valid in the current state */
    XWB_ERR_ACTION = -4     /* action id out of range (reference: CHECK_LT -> abort) */
};

/* games selected by name in SimulatorInterface::SimulatorInterface, simulator_interface.cpp:41-56 */
enum { XWB_SIMPLE_GAME = 0, XWB_SIMPLE_RACE = 1, XWB_XWORLD2D = 2 };

/* action id that leaves an env out of a xwb_step call (its state, reward, game_over and observation
 * are not touched): lets a per-env SimulatorInterface view step one slot of a batch */
#define XWB_ACTION_SKIP (-1)

/* GameOverCode, simulator.h:42-48 */
enum { XWB_ALIVE = 0, XWB_MAX_STEP = 1, XWB_DEAD = 2, XWB_SUCCESS = 4, XWB_LOST_LIFE = 8 };

enum { XWB_MAP_NAV = 0, XWB_MAP_WALLS = 1 };             /* games/xworld/maps/XWorldNav.py, XWorldWalls.py */
/* tasks of games/xworld3d/tasks/XWorld3DNav*.py (the full-observation 2-D game runs the same Python tasks) */
enum { XWB_TASK_TARGET = 0, XWB_TASK_NEAR = 1, XWB_TASK_BETWEEN = 2, XWB_TASK_DIRECTION = 3, XWB_TASK_AVOID = 4,
       /* the 2-D-native group "XWorldNav" of confs/walls.json: games/xworld/tasks/XWorldNav{Target,Near,ColorTarget,
        * Between}.py with XWorldTask.simple_navigation_reward (xworld_task.py:184-223).  A group holds tasks of one
        * family only. */
       XWB_TASK2D_TARGET = 5, XWB_TASK2D_NEAR = 6, XWB_TASK2D_COLOR = 7, XWB_TASK2D_BETWEEN = 8 };
#define XWB_CELL_ICON_MASK 0x7fff   /* cell code & mask = palette icon + 1 (0 = empty) */
#define XWB_CELL_TARGET    0x8000   /* cell code bit: this goal belongs to the task's target set */
enum { XWB_TASKMODE_LANG_ACQ = 0, XWB_TASKMODE_ONE_CHANNEL = 1 };   /* FLAGS_task_mode, xworld_simulator.cpp:33-37 */
enum { XWB_EV_NONE = 0, XWB_EV_CORRECT_GOAL = 1, XWB_EV_WRONG_GOAL = 2, XWB_EV_TIME_UP = 3 };
enum { XWB_OBS_U8 = 0, XWB_OBS_F32 = 1 };
enum { XWB_SCHEDULE_RANDOM = 0, XWB_SCHEDULE_WEIGHTED = 1 };
enum { XWB_ICON_GOAL = 0, XWB_ICON_BLOCK = 1, XWB_ICON_AGENT = 2 };  /* xworld_env.py:66 grid_types */
/* where the decisions the reference takes with util::get_rand_ind / get_rand_range_val come from (simulator_util.cpp:38-73):
 * the batch's own counter-based streams (default), or one libstdc++ minstd_rand0 per env, seeded and consumed exactly as the
 * reference's thread-local engine (include/xwb_minstd.h) -- replays a reference run made with --simulator_seed != 0 */
enum { XWB_RNG_PHILOX = 0, XWB_RNG_MINSTD = 1 };

/*
 * Batch configuration.  Field names follow the reference's gflags / py_simulator
 * option names (python/py_simulator.cpp:97-136, simulator.cpp:21-27).  The
 * reference keeps these as process-global gflags; here they are per batch.
 */
typedef struct xwb_config {
    int32_t  abi_version;        /* XWB_ABI_VERSION */
    int32_t  game;               /* XWB_SIMPLE_GAME | XWB_SIMPLE_RACE | XWB_XWORLD2D */
    int32_t  num_envs;           /* environments in this batch (this GPU's shard) */
    int32_t  device;             /* HIP device ordinal */
    uint32_t env_gid0;           /* global id of local env 0: RNG streams are keyed by global id,
                                    so results do not depend on how a batch is sharded over GPUs */
    uint32_t seed;               /* xwb-rng-v1 seed for reset streams */
    uint32_t policy_seed;        /* xwb-rng-v1 seed for the built-in uniform random policy */
    int32_t  context;            /* FLAGS_context (simulator.cpp:21) */
    int32_t  max_steps;          /* FLAGS_max_steps (simulator.cpp:22), 0 = off */

    /* simple_game (simple_game_simulator.cpp:19) */
    int32_t  array_size;

    /* simple_race (simple_race_simulator.cpp:17-26) */
    int32_t  track_type;         /* 0 "straight", 1 "circle" */
    int32_t  race_full_manouver;
    int32_t  random;
    int32_t  difficulty_hard;    /* 0 "easy", 1 "hard" */
    double   track_width, track_length, track_radius, reward_scale;

    /* xworld (xworld_simulator.cpp:22-37, teacher.cpp:22-25, simulator.cpp:23-25) */
    int32_t  map_kind;           /* XWB_MAP_* : the "map" key of the conf JSON (xworld.cpp:73-76) */
    int32_t  max_dim;            /* max_height == max_width of the map class */
    int32_t  dim;                /* actual dim (== max_dim when FLAGS_curriculum == 0) */
    int32_t  num_goals, num_blocks;
    int32_t  max_steps_factor;   /* FLAGS_max_steps_factor (10) */
    int32_t  task_mode;          /* XWB_TASKMODE_* */
    int32_t  n_tasks;            /* tasks of the teacher's group (conf JSON "teacher"."task_groups", teacher.py:
                                  * TaskGroup samples one per episode); 0 = { XWB_TASK_TARGET } */
    int32_t  tasks[8];           /* XWB_TASK_* */
    int32_t  color;              /* FLAGS_color: 3-channel planar BGR when set, else 1-channel gray */
    int32_t  visible_radius;     /* FLAGS_visible_radius: 0 = full observation (4 actions, 12 px per cell); odd r > 0 =
                                  * egocentric r x r view, 6 first-person actions, frame edge r * (84 / r)
                                  * (xworld_simulator.cpp:62-68, xitem.cpp:80-87); XWorldNav maps only */
    int32_t  obs_format;         /* XWB_OBS_U8 (the reference's screen bytes) | XWB_OBS_F32: float32 pixel * (1/255.0f),
                                  * the scaling py_simulator.cpp:262-272 applies in get_state(), done on the device */
    int32_t  n_icons;            /* icons this map class can place (its "palette") */
    const uint8_t *icons64;      /* host, n_icons x 64 x 64 x 3, BGR as cv::imread(path, 1) (xitem.cpp:38) */
    const int32_t *icon_type;    /* host, n_icons, XWB_ICON_* */
    const int32_t *icon_name;    /* host, n_icons, index into the sorted names of that type */
    const int32_t *icon_colored; /* host, n_icons or NULL: properties.txt colour != "na" (xworld_env.py:201-205);
                                  * only XWB_TASK2D_COLOR reads it */
    double   curriculum;         /* FLAGS_curriculum (py_simulator.cpp:127): != 0 -> every XWorldNav env grows through the six
                                  * levels of XWorldNav.py:27-55 (dims 3..8, goals 2/2/2/4/4/4, blocks 0/3/6/9/12/16) as its own
                                  * success rate passes the value -- checked every 100th reset over the last 200 results of
                                  * each task class (xworld_env.py:103-110, xworld3d_task.py:129-146); dim / num_goals /
                                  * num_blocks are then ignored and max_dim must be 8.  XWorldWalls never reads it. */
    int32_t  start_level;        /* XWorldNav(item_path, start_level): the level a --curriculum_stamp file holds (xworld.cpp:93-100) */
    int32_t  task_schedule;      /* the group's "schedule" (teaching_task.cpp:204-213): XWB_SCHEDULE_RANDOM = uniform over its
                                  * tasks, XWB_SCHEDULE_WEIGHTED = util::simple_importance_sampling over task_weights */
    double   task_weights[8];    /* the per-task numbers of the conf JSON (TaskGroup::add_task: > 0); read when weighted */
    int32_t  no_wall_shadow;     /* != 0: FLAGS_wall_shadow = false (xmap.cpp:19,170): the egocentric view keeps the cells
                                  * behind walls visible (a gflag of the C++ binaries; not settable from py_simulator) */
    int32_t  rng_mode;           /* XWB_RNG_* */
    int32_t  simulator_seed;     /* FLAGS_simulator_seed (simulator_util.cpp:26): must be != 0 with XWB_RNG_MINSTD */
    int32_t  thread_base;        /* XWB_RNG_MINSTD: simulator threads the reference process had created before this batch's
                                  * first env; global env g uses the engine of thread number thread_base + g + 1 */
    /* A second task group of the conf's "task_groups" (listed after the first; the teacher keeps conf order,
     * teacher.cpp:56-98): n_tasks2 == 0 -> none.  One of the two groups holds XWorld3DNav* tasks, the other the 2-D-native
     * XWorldNav* ones.  Run NON-exclusively -- Teacher::teach's else branch (teacher.cpp:221-225) --, every teach() runs
     * each group's stage in conf order, rewards add up, the last group's event ("" included) is what game_over() sees, and
     * only the first group's task sees the step's collision events (xworld_simulator.cpp:118-122) -- unless
     * task_groups_exclusive is in force, see below.
     * LIMIT: at most TWO built groups per batch.  The reference's Teacher takes any number (teacher.cpp:110-163); none of
     * its shipped confs lists more than two whose tasks exist in this snapshot (confs/walls.json: XWorldNav + the Rec / Lan
     * groups this tier leaves out; confs/navigation2d.json: one).  A conf with a third built group is refused by the loader
     * (assets.conf_groups), and xwb_xw_load_map_task replays a map for ONE group's task. */
    int32_t  n_tasks2;
    int32_t  tasks2[8];
    int32_t  task_schedule2;
    double   task_weights2[8];
    int32_t  task_groups_exclusive;  /* FLAGS_task_groups_exclusive (teacher.cpp:22-24).  task_mode lang_acquisition forces it
                                      * off exactly as the reference does (simulator_interface.cpp:46-48).  Set: Teacher::teach's
                                      * exclusive branch (teacher.cpp:209-220) -- every teach() first re-sorts the groups by
                                      * weighted sampling without replacement (nondeterministic_sort_task_groups, :143-163; the
                                      * order persists from call to call), then runs ONE group's stage: the last busy group of
                                      * that order, else its first.  An idle XWorld3DNav* group picked in mid-episode runs its
                                      * map-rearranging idle stage at step time. */
    double   task_group_weight;      /* the groups' "weight" keys of the conf JSON (teacher.cpp:83-91; 0 when absent), read by */
    double   task_group_weight2;     /* the exclusive branch's shuffle only */
    int32_t  queue_sync;             /* XWB_QUEUE_SYNC_*: how the batch's two internal queues hand over inside a step (see
                                      * xwb_queue_sync_mode) */
    /* Debug configuration: A/B switches of the library's own kernel paths -- every path gives the same results, byte for
     * byte; the tests hold one against the other.  All zero in production.  Per batch (the reference keeps such switches as
     * process-global gflags).  The ONE process-wide override: the environment variable XWB_DEBUG, a comma-separated list of
     * the lower-case flag names (no_pregen, no_lazy, ego_no_cache, ego_no_span, ego_no_flat) and of ego_per=N, ego_pad=N,
     * render_shape=64x2|256x2, read once at the first xwb_create and OR-ed into every batch created afterwards (for tools
     * that cannot reach the configuration).  Besides it only XWB_QUEUE_SYNC is read from the environment. */
    int32_t  debug_flags;            /* XWB_DEBUG_* */
    int32_t  debug_ego_per;          /* egocentric gather: 16-byte chunks per lane, 2 | 4 | 8; 0 = the default (4) */
    int32_t  debug_ego_pad;          /* egocentric gather: extra LDS bytes a workgroup asks for, + 1 (0 = the default occupancy cap) */
    int32_t  debug_render_shape;     /* full-observation render_all: 0 = 128 threads x 2 chunks (default), 1 = 64 x 2, 2 = 256 x 2 */
} xwb_config;
enum { XWB_DEBUG_NO_PREGEN = 1,      /* no pre-generated episodes: every verb on the classic path */
       XWB_DEBUG_NO_LAZY = 2,        /* xwb_step + xwb_reset_done on the classic path (xwb_step_autoreset keeps its pre-generation) */
       XWB_DEBUG_EGO_NO_CACHE = 4,   /* egocentric: no cache of rendered goal cells (and hence no span path) */
       XWB_DEBUG_EGO_NO_SPAN = 8,    /* egocentric: one workgroup per env instead of the span path */
       XWB_DEBUG_EGO_NO_FLAT = 16,   /* egocentric span path: no shared constant line for one-colour squares */
       XWB_DEBUG_NO_FUSED = 32 };    /* xwb_step of the default loop as two launches (step, render) instead of one */

/* xwb_config.queue_sync.  AUTO: device-side epochs (no event / barrier packets: 12 us per step on the C4 loop) on every caller
 * stream that passed a one-time concurrency probe against the batch's internal stream -- the default stream is probed by
 * xwb_create, any other stream by the caller's xwb_queue_sync_mode(sim, stream, ...) call, which synchronises that stream once;
 * the step verbs never probe (they stay asynchronous) and use events on a stream nobody probed --, events otherwise (the probe fails when
 * the two streams share one hardware queue -- HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues -- or when a tool
 * serialises kernel execution; a probe that finds no concurrency first replaces the batch's internal stream by one that does
 * run beside the probed stream and beside every stream that passed before -- a shared hardware queue would also put the map
 * generator behind the render instead of beside it --, and fails only when no such stream is found); the environment variable XWB_QUEUE_SYNC=events|epochs and the presence of a serialising tool
 * (rocprofv3 counter collection, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING) override AUTO.  EVENTS / EPOCHS force one mode. */
enum { XWB_QUEUE_SYNC_AUTO = 0, XWB_QUEUE_SYNC_EVENTS = 1, XWB_QUEUE_SYNC_EPOCHS = 2 };
/* why xwb_queue_sync_mode reports the mode it reports */
enum { XWB_SYNC_REASON_PROBE_OK = 0, XWB_SYNC_REASON_CONFIG = 1, XWB_SYNC_REASON_ENV = 2, XWB_SYNC_REASON_TOOL = 3,
       XWB_SYNC_REASON_PROBE_FAILED = 4, XWB_SYNC_REASON_PROBE_ERROR = 5, XWB_SYNC_REASON_NOT_USED = 6,
       XWB_SYNC_REASON_NOT_PROBED = 7 /* nobody probed this stream (or it is under graph capture): events */ };

typedef struct xwb_sim xwb_sim;

/* fills *cfg with the reference's defaults for `game` (gflags DEFINE_* defaults) */
int xwb_default_config(int32_t game, xwb_config *cfg);

/* SimulatorInterface::SimulatorInterface(name, false) for num_envs envs, simulator_interface.cpp:37-85.
 * Allocates the SoA state in HBM, uploads the icon atlas.  Synchronous. */
int xwb_create(const xwb_config *cfg, xwb_sim **out);
int xwb_destroy(xwb_sim *sim);

/* SimulatorInterface::reset_game for every env, simulator_interface.cpp:95-105
 * (game reset -> teacher reset + idle stage -> init_screen). */
int xwb_reset(xwb_sim *sim, void *stream);

/* The reference example loop's `if game_over() != alive: reset_game()` (examples/test_xworld.cpp:41-45,
 * python/examples/test_simple_game.py:19-21) applied to every env whose game_over code is non-zero:
 * wavefront-ballot compaction of the done mask, reset of the compacted list, re-render of those envs.
 * Stream ordering: reward, game_over codes and the observation frames of the preceding step (the terminal frames of the
 * finished envs included) stay readable by work queued on `stream` before this call: the codes are cleared and the first
 * frames of the new episodes are stored on `stream` (tests/test_gpu_stream_order.py).  For XWorld2D the map generation
 * itself runs on an internal stream beside that step's render, so the STATE arrays of the finished envs (grid, agent cell,
 * num_steps: the xwb_*_dev views) may already hold the new episode: read them after the step through xwb_get_env_state
 * (synchronises) or before calling this function with `stream` synchronised. */
int xwb_reset_done(xwb_sim *sim, void *stream);

/* same, for an explicit device mask (mask_dev[e] != 0 -> reset env e) */
int xwb_reset_masked(xwb_sim *sim, const uint8_t *mask_dev, void *stream);

/* reset_game of ONE env (per-slot SimulatorInterface views) */
int xwb_reset_env(xwb_sim *sim, int32_t env, void *stream);

/* SimulatorInterface::take_actions(actions, act_rep, false) for every env, simulator_interface.cpp:126-137:
 * GameSimulator::take_actions (num_steps_++ once, act_rep x take_action) -> teacher -> make_context_screens.
 * actions_dev: int32[num_envs] ("action" id of each env's StatePacket), or NULL to draw each env's action
 * from the built-in uniform random policy (xwb-rng-v1 stream 1; the actions used are kept in xwb_actions_dev).
 * Out-of-range action ids set the env's error flag (see xwb_check_errors) and leave that env untouched. */
int xwb_step(xwb_sim *sim, const int32_t *actions_dev, int32_t act_rep, void *stream);

/* xwb_step with the action ids in HOST memory (int32[num_envs]): 4 bytes per env over PCIe, the only host->device traffic of a
 * step.  Page-locked memory (hipHostMalloc / hipHostRegister, torch pin_memory) is read by the step kernel in place -- keep it
 * unchanged until `stream` has passed the call --; pageable memory is copied to the device on `stream` first. */
int xwb_step_host(xwb_sim *sim, const int32_t *actions_host, int32_t act_rep, void *stream);

/* xwb_step followed by xwb_reset_done in one call, with a single render of the final state
 * (the observation of a finished env is the first frame of its next episode; reward / game_over
 * keep the values of the terminal transition).  XWorld2D under full observation (no curriculum, no XWB_RNG_MINSTD, no
 * exclusive scheduling of two groups): every env's NEXT episode is kept pre-generated, the step kernel itself starts it for
 * the envs it finishes and one render draws the whole batch -- same results, byte for byte, as xwb_step + xwb_reset_done;
 * the pre-generated episodes are rebuilt (once, whole batch) when another verb started episodes in between. */
int xwb_step_autoreset(xwb_sim *sim, const int32_t *actions_dev, int32_t act_rep, void *stream);
/* n_steps consecutive xwb_step_autoreset calls under the built-in random policy (actions drawn on the device).  For
 * SimpleGame / SimpleRace they run inside ONE launch -- a step there moves a few MB and is launch-bound -- with every
 * step's reward, code and observation written exactly as separate launches would; XWorld2D loops on the host. */
int xwb_step_n(xwb_sim *sim, int32_t n_steps, int32_t act_rep, void *stream);

/* The reference example loop's body -- `if (game_over) reset_game(); take_actions(random action)`, examples/test_simple_race.cpp:26-53,
 * python/examples/test_simple_game.py:15-30 -- `iterations` times over under the built-in policy, issued from C: flags 0 =
 * xwb_step(NULL, act_rep) then xwb_reset_done per iteration (terminal frames, then first frames: both calls' effects, in order),
 * XWB_RUN_AUTORESET = xwb_step_autoreset per iteration.  Exactly the launches of that many separate calls, one foreign-function
 * call: a binding's per-call cost (ctypes: ~1.5 us) leaves the loop, which matters where a step is launch-bound (the simple games). */
enum { XWB_RUN_AUTORESET = 1 };
int xwb_run(xwb_sim *sim, int32_t iterations, int32_t act_rep, int32_t flags, void *stream);

/* returns the number of envs that flagged an out-of-range action since the last call (synchronises stream).
 * Fails with XWB_ERR_STATE when the batch is poisoned (see xwb_queue_sync_mode). */
int xwb_check_errors(xwb_sim *sim, void *stream, int32_t *n_bad);

/* How the batch hands work between the caller's `stream` and its internal stream: *mode = XWB_QUEUE_SYNC_EVENTS or
 * XWB_QUEUE_SYNC_EPOCHS for calls made on `stream`.  This call is what PROBES a stream (once; it synchronises `stream` and
 * the host; skipped -- events, reason NOT_PROBED -- while the stream is under graph capture): a trainer that issues the verbs
 * on a stream of its own calls it once after creating that stream, *reason = XWB_SYNC_REASON_* (NOT_USED: the game has no internal stream).  Safety of the epoch hand-off: (1) the kernel
 * that publishes an epoch is always enqueued before the kernel that waits for it, so streams that turn out to share a hardware
 * queue, or kernels serialised in submission order, cannot deadlock; (2) the probe; (3) a device-side watchdog: a wait that is
 * not released within 4 s POISONS the batch -- the queues drain, and every later verb of the batch (step, reset, getters,
 * xwb_check_errors) fails with XWB_ERR_STATE; results since the last successful xwb_check_errors are void and the batch can
 * only be destroyed.  Returns XWB_ERR_STATE itself when the batch is poisoned. */
int xwb_queue_sync_mode(xwb_sim *sim, void *stream, int32_t *mode, int32_t *reason);
/* Which kernel sequence the LAST step call (xwb_step / xwb_step_autoreset / xwb_step_n / xwb_run) ran -- the verbs choose it from the
 * configuration and from what the caller did before, and a number measured on one path says nothing
 * about another: *path = XWB_PATH_*; *sync_mode (nullable) = XWB_QUEUE_SYNC_EVENTS | _EPOCHS of that call (AUTO: the game has
 * no internal queue); *shadow_breaks (nullable) = how often another verb made the pre-generated episodes stale (after three
 * the default loop stays on the classic path). */
enum { XWB_PATH_NONE = 0,        /* SimpleGame / SimpleRace (one kernel), or no step yet */
       XWB_PATH_CLASSIC = 1,     /* step -> render(all; finished envs from terminal snapshots); reset on the internal queue */
       XWB_PATH_LAZY = 2,        /* pre-generated episodes installed by xwb_reset_done's list render (the default loop) */
       XWB_PATH_PREGEN = 3,      /* xwb_step_autoreset: the step kernel itself starts the pre-generated episode */
       XWB_PATH_EGO_SPAN = 4, XWB_PATH_EGO_PER_ENV = 5, /* egocentric renders, see xwb_ego_render_path */
       XWB_PATH_LAZY_FUSED = 6 };/* XWB_PATH_LAZY as ONE launch: under the built-in policy (actions_dev == NULL) the previous lazy step also
                                  * left the grids as THIS step will leave them (the policy's next action is a function of seed, env and
                                  * step number; a step moves at most two cells), so the render's workgroups draw from that look-ahead
                                  * snapshot while the step's workgroups run beside them in the same kernel.  Context 1, same act_rep as
                                  * the previous call, no verb in between that rewrote the maps; anything else: XWB_PATH_LAZY */
int xwb_step_path(xwb_sim *sim, int32_t *path, int32_t *sync_mode, int32_t *shadow_breaks);
/* drops what the batch remembers about `stream` (call before destroying a probed stream: a later stream may reuse the handle) */
int xwb_queue_sync_forget(xwb_sim *sim, void *stream);

/* ---- observation / result buffers (device pointers, valid until xwb_destroy) ---- */
/* "screen" of get_state(): [num_envs][context][c][h][w]; uint8 for simple_game / xworld (planar B,G,R),
 * float32 for simple_race (simple_race_simulator.cpp:412-430).  Newest frame last (simulator.cpp:51-60). */
int xwb_obs_dev(xwb_sim *sim, void **ptr, size_t *bytes_per_env);
/* Optional second output of every step: packed_dev[e] = (reward, game_over code as float) for every env the call steps,
 * float[num_envs][2] in caller-owned device memory (NULL = off) -- one buffer to ship per step (sharding.ResultGather). */
int xwb_bind_results(xwb_sim *sim, float *packed_dev);
/* The same with a ring of `slots` such buffers, float[slots][num_envs][2]: the k-th step call after the bind writes slot
 * k % slots (one xwb_step_n call = one slot for every game: each of its steps writes it, the last one stays).  A per-step record of a rollout without a
 * host call or a copy kernel per step -- bench.py's parity gate and the pipelined result exchange read it (SURVEY 8(d):
 * "parity gates reported with every perf number"). */
int xwb_bind_results_ring(xwb_sim *sim, float *packed_dev, int64_t slots);

/* redirect the observation output to caller-owned device memory (e.g. a shard of a gathered tensor) */
int xwb_bind_obs(xwb_sim *sim, void *obs_dev);
int xwb_reward_dev(xwb_sim *sim, float **ptr);          /* float[num_envs]: return value of take_actions */
int xwb_game_over_dev(xwb_sim *sim, uint8_t **ptr);     /* uint8[num_envs]: SimulatorInterface::game_over() code */
int xwb_actions_dev(xwb_sim *sim, int32_t **ptr);       /* int32[num_envs]: actions applied by the last step */
int xwb_num_steps_dev(xwb_sim *sim, int32_t **ptr);     /* int32[num_envs]: get_num_steps() */
int xwb_success_dev(xwb_sim *sim, uint8_t **ptr);       /* uint8[num_envs]: last_action_success() */
int xwb_episode_dev(xwb_sim *sim, uint32_t **ptr);      /* uint32[num_envs]: resets so far (RNG episode index) */
int xwb_xw_grid_dev(xwb_sim *sim, uint16_t **ptr);      /* xworld: uint16[num_envs][max_dim*max_dim] cell codes -- READ-ONLY (since round 5): the
                                                         * step kernel finds goals through a per-env goal-slot table, the default loop draws from
                                                         * look-ahead snapshots of the grids, the egocentric render caches pixels that depend on
                                                         * the cells around a goal for the length of an episode -- all of which a write through
                                                         * this pointer would leave stale; a map is changed through xwb_xw_load_map_task */
int xwb_minstd_state_dev(xwb_sim *sim, uint32_t **ptr); /* XWB_RNG_MINSTD: uint32[num_envs] engine states (else NULL) */
int xwb_done_count(xwb_sim *sim, void *stream, int32_t *n_done);   /* envs reset by the last reset_done (sync) */
/* xworld, egocentric: which kernels draw the whole-batch frames: 1 = the span path (cells -> evaluated pixels -> gather,
 * kernels_xworld_ego.hip), 0 = one workgroup per env.  Both are bit-exact; callers that report kernel times need to know
 * which ran.  Radii: XMap::image_masking admits ODD radii only (CHECK, xmap.cpp:277; xwb_create refuses even ones with the same
 * words) and XWorldSimulator::init clips the radius to the map's edge (xworld_simulator.cpp:62-68), so r is one of 1, 3, ..., 15.
 * The span path draws r = 3, 5, 7 -- squares of 28, 16 and 12 pixels -- on every map size.  The per-env kernel
 * (XWB_PATH_EGO_PER_ENV) is what draws, and the only thing that can draw:
 *   - visible_radius = 1, 9, 11, 13, 15: r = 1 is one 84-pixel square (the agent's own cell: nothing to tile); from r = 9 on the
 *     square's edge is not a multiple of four pixels (r = 9: 9 px, 81 x 81 frames; 11: 7 px, 77 x 77; 13 and 15: 6 and 5 px) --
 *     frame rows are not whole dwords, frames not whole 16-byte chunks -- and the view has more than 64 cells, which the
 *     cells kernel's shadow masks (one 64-bit word per env) do not hold.  Not generalised: DESIGN.md section 9 has the
 *     measurement (0.035 of the roofline at r = 9) and what the generalisation needs;
 *   - palettes with more than 16 images that every env shares (blocks, agents, empty, black): the span path's table of
 *     squares is keyed by three such classes and is capped at 128 MB;
 *   - XWB_DEBUG_EGO_NO_SPAN / _NO_CACHE, or not enough free device memory for the span path's tables at xwb_create.
 * It is an order of magnitude slower per frame (bench.py --workload xworld11_ego9 prints a line for it).  (No conf of the
 * reference sets visible_radius at all -- python/examples/test_xworld.py:39 passes 0 --; 3, 5 and 7 are the radii with whole
 * squares that fit the 7x7 / 8x8 / 11x11 maps of its confs.) */
int xwb_ego_render_path(xwb_sim *sim, int32_t *path);

/* The whole batch's outputs copied into caller-owned memory, host or device (hipMemcpyDefault), ordered on `stream`;
 * host destinations are complete when the call returns.  obs: num_envs * bytes_per_env (see xwb_obs_dev);
 * reward: float[num_envs]; done: uint8[num_envs] game_over codes. */
int xwb_get_obs(xwb_sim *sim, void *dst, size_t bytes, void *stream);
int xwb_get_reward(xwb_sim *sim, float *dst, void *stream);
int xwb_get_done(xwb_sim *sim, uint8_t *dst, void *stream);

/* ---- static queries (SimulatorInterface getters) ---- */
int xwb_get_num_actions(const xwb_sim *sim, int32_t *n);                         /* get_num_actions() */
int xwb_get_screen_out_dimensions(const xwb_sim *sim, size_t *h, size_t *w, size_t *c);
int xwb_get_world_dimensions(const xwb_sim *sim, double *X, double *Y, double *Z);
int xwb_num_envs(const xwb_sim *sim, int32_t *n);

/* ---- per-env host access: the scalar SimulatorInterface surface (all synchronise `stream`) ---- */
typedef struct xwb_env_state {
    float    reward;             /* last take_actions return value */
    int32_t  game_over;          /* SimulatorInterface::game_over() */
    int32_t  lives;              /* get_lives() */
    int64_t  num_steps;          /* get_num_steps() */
    int32_t  last_action;        /* last_action() as an id, -1 before the first step */
    int32_t  last_action_success;
    /* game specific */
    int32_t  sg_pos;             /* simple_game: _cur_pos */
    float    race_x, race_y, race_angle;
    int32_t  xw_agent_x, xw_agent_y, xw_event, xw_stage, xw_target_name, xw_steps_in_task;
    uint32_t episode;
    int32_t  xw_task;            /* XWB_TASK_* of this episode */
    int32_t  xw_target;          /* TARGET: goal name id; BETWEEN and the 2-D-native tasks: cell y * max_dim + x;
                                  * DIRECTION: referent cell | direction word << 8 (1 front 2 behind 3 left 4 right); else -1 */
    int32_t  xw_agent_dir;       /* egocentric heading: 0 right, 1 down, 2 left, 3 up; 1 under full observation */
    int32_t  xw_level;           /* curriculum: XWorldEnv.current_level (0 when FLAGS_curriculum == 0) */
    int32_t  xw_check_counter;   /* curriculum: XWorldEnv.curriculum_check_counter */
    uint32_t xw_sentence_names;  /* goal-name ids the idle stage binds into the teacher's sentence: a | b << 16 (0xffff none):
                                  * TARGET G = the picked goal; NEAR G = g1; BETWEEN G1, G2; DIRECTION, AVOID G = the referent */
    /* the second task group's Task FSM (n_tasks2 > 0; else zeros); xw_event / xw_event2 = what each group's task recorded */
    int32_t  xw_task2, xw_stage2, xw_event2, xw_target2, xw_steps_in_task2;
    /* exclusive scheduling of two groups (else -1): conf index (0 / 1) of the group that heads Teacher::task_groups_ after
     * the last sort, and of the one group the last teach() ran */
    int32_t  xw_group_first, xw_group_ran;
} xwb_env_state;
int xwb_get_env_state(xwb_sim *sim, int32_t env, void *stream, xwb_env_state *out);
/* copies env's "screen" (context frames) to host memory; bytes must equal bytes_per_env */
int xwb_get_env_obs(xwb_sim *sim, int32_t env, void *stream, void *out_host, size_t bytes);
/* xworld: cell codes (palette icon + 1, 0 = empty; | XWB_CELL_TARGET on the task's target goals),
 * max_dim*max_dim uint16, row-major [y][x] */
int xwb_get_env_grid(xwb_sim *sim, int32_t env, void *stream, uint16_t *out_host);

/* replay an externally generated map into env (golden-map parity): grid cell codes, agent cell,
 * teacher target name id and actual dim; runs init_screen.  Synchronous. */
int xwb_xw_load_map(xwb_sim *sim, int32_t env, const uint16_t *grid_host, int32_t agent_x, int32_t agent_y,
                    int32_t target_name, int32_t dim);
/* the same for any task: grid codes carry XWB_CELL_TARGET on the target goals; `target` as xw_target above */
int xwb_xw_load_map_task(xwb_sim *sim, int32_t env, const uint16_t *grid_host, int32_t agent_x, int32_t agent_y,
                         int32_t dim, int32_t task, int32_t target);
/* egocentric replays: the agent's heading, and the pose (xworld_env.py:207-223: yaw, scale, offset) of the goal at a
 * cell; the warp matrix is derived on the host exactly as XItem::get_item_image does.  Synchronous; call
 * xwb_xw_refresh_obs afterwards to re-render the env. */
int xwb_xw_set_agent_dir(xwb_sim *sim, int32_t env, int32_t dir);
int xwb_xw_set_goal_pose(xwb_sim *sim, int32_t env, int32_t cell_x, int32_t cell_y, double yaw, double scale, double offset);
int xwb_xw_refresh_obs(xwb_sim *sim, int32_t env);
/* simple_race: overwrite the car state of env (test hook).  Synchronous. */
int xwb_race_set_car(xwb_sim *sim, int32_t env, float x, float y, float angle);

/* xworld: the strings behind the palette's name ids -- goal_names[id] for xwb_config.icon_name of goal icons, and per icon its
 * name and its colour ("na": none; properties.txt).  With them the library builds the teacher's sentences itself
 * (xworld_amd/csrc/xwb_language.h: the reference's per-task context-free grammars, python/context_free_grammar.py and the
 * tasks' _define_grammar, expanded with xwb-rng-v1 stream 3); the strings are copied. */
int xwb_set_names(xwb_sim *sim, const char *const *goal_names, int32_t n_goal_names, const char *const *icon_names,
                  const char *const *icon_colors, int32_t n_icons);
/* The teacher's sentence of one env after the last call, NUL-terminated ("" where the reference's get_state() shows "-").
 * Returns the bytes needed in *need; writes when cap suffices.  Needs xwb_set_names.  Synchronises `stream`. */
int xwb_sentence(xwb_sim *sim, int32_t env, void *stream, char *out, size_t cap, size_t *need);

/* The sentence functions behind xwb_sentence, callable without a batch or a GPU (host only; tests pin them to
 * xworld_amd/language.py, which is pinned to the reference's CFG): a 3-D task's sentence from its state (stage, event as in
 * xwb_env_state; name ids into goal_names, 0xffff none; direction 1 front, 2 behind, 3 left, 4 right), and a 2-D-native
 * task's instruction (task 5 / 7; timeup != 0: its "Time up ." message).  NUL-terminated; *need = bytes needed. */
int xwb_language_sentence(int32_t task, int32_t stage, int32_t event, const char *const *goal_names, int32_t n_goal_names,
                          uint32_t name_a, uint32_t name_b, int32_t direction, uint32_t seed, uint32_t gid, uint32_t episode,
                          char *out, size_t cap, size_t *need);
int xwb_language_sentence_2d(int32_t task, int32_t timeup, const char *goal_name, const char *color, uint32_t seed, uint32_t gid,
                             uint32_t episode, uint32_t num_steps, char *out, size_t cap, size_t *need);

/* SimulatorInterface::get_state(reward) of one env, serialised in the reference's StatePacket wire
 * layout (data_packet.h:313-319, data_packet.cpp:143-174, memory_util.h:307-333): keys "reward",
 * "screen" [, "sentence" for xworld].  Returns bytes needed in *need; writes when cap suffices.
 * "sentence" is the teacher's sentence (xworld_simulator.cpp:486-493), "-" when the teacher is silent -- and always "-" until
 * xwb_set_names has handed over the strings behind the name ids (the config only carries ids). */
int xwb_get_state_packet(xwb_sim *sim, int32_t env, float reward, void *stream,
                         uint8_t *out_host, size_t cap, size_t *need);

/* ---- checkpoint / resume of a batch (replaces the reference's --curriculum_stamp style restarts) ----
 * The whole dynamic state of the batch -- SoA arrays, episode counters (= the RNG stream positions: every draw is a
 * pure function of (seed, global env id, episode / step counters)), the built-in policy's step counter and, when
 * `include_obs` is set, the observation buffer (needed for bit-exact context rings; without it the frames are
 * re-rendered from the state on load and the older context frames start black) -- as one host blob.  A blob loads into
 * any batch created with the same configuration.  Both calls synchronise. */
int xwb_state_bytes(xwb_sim *sim, int32_t include_obs, size_t *bytes);
int xwb_save_state(xwb_sim *sim, int32_t include_obs, uint8_t *out_host, size_t cap);
int xwb_load_state(xwb_sim *sim, const uint8_t *in_host, size_t bytes);

/* SimulatorInterface::get_extra_info of one env.  XWorld2D (xworld_simulator.cpp:495-504):
 * "<pid>|task:<task class>,event:<event>,height:<actual h>,width:<actual w>"; the other games: "". */
int xwb_get_extra_info(xwb_sim *sim, int32_t env, void *stream, char *out, size_t cap);

/* SimulatorInterface::teacher_report_task_performance (simulator_interface.cpp:149-153 -> Teacher::report_task_performance,
 * teacher.cpp:175-200): what every task object's obtain_performance() returns -- num_successes, num_failures, success_steps
 * (xworld3d_task.py:135-142, 383-384) -- summed over the batch's envs since xwb_create, per task class (index = XWB_TASK_*).
 * success_steps: the XWorld3DNav* tasks add steps_in_cur_task on every success; the 2-D-native tasks keep none (their
 * obtain_performance returns a 2-tuple: the reference's report would abort on them).  time_ups = the failures that were
 * time-ups; *resets (nullable) = games reset.  Counted on the device by the step / reset kernels; the call synchronises. */
typedef struct xwb_task_performance { int64_t successes, failures, success_steps, time_ups; } xwb_task_performance;
int xwb_get_task_performance(xwb_sim *sim, void *stream, xwb_task_performance out[9], int64_t *resets);
/* the same as the text the reference logs: per task class that occurred "=== <task class> ===" and
 * "=== <S>(S)/<F>(F) -> <success rate>@<steps per success>" (-1 when no success).  *need = bytes needed incl. NUL. */
int xwb_task_performance_report(xwb_sim *sim, void *stream, char *out, size_t cap, size_t *need);

/* GameSimulator::decode_game_over_code, simulator.cpp:125-144 ("alive" | "max_step|dead|...") */
int xwb_decode_game_over_code(int32_t code, char *out, size_t cap);

/* the xworld 12x12 tile table built from icons64 by the OpenCV-3.2 fixed-point bilinear rule
 * (what XWorldSimulator::down_sample_image produces for one cell): host copy,
 * n_icons x c x 12 x 12 bytes. */
int xwb_xw_get_tile_table(const xwb_sim *sim, uint8_t *out_host, size_t cap, size_t *need);

/* ---- the draw state: "gather the state, not the pixels" ----
 * Under full observation a frame is a pure function of the env's cell codes (2 * max_dim^2 bytes against 144 * c * max_dim^2
 * bytes of pixels: C5 242 B against 52 272 B), so a holder elsewhere -- the root GPU of a sharded batch -- can draw the frames
 * itself.  xwb_xw_pack_grids writes, for every env, the cell codes its CURRENT observation was drawn from (uint16
 * [num_envs][max_dim^2], icon + 1, 0 = empty; the teacher's target bit is stripped) and the context-ring operation of its
 * last draw (uint8[num_envs]: 0 untouched, 1 ring shift, 2 first frame of an episode -- the older context frames start
 * black); flags_dev may be NULL when context == 1.  Ordered on `stream` behind the verbs queued before it.  With
 * context > 1 it must follow EVERY verb that draws frames (step, reset_done, ...: a ring is replayed one draw at a time),
 * else XWB_ERR_STATE -- also after the map-replay hooks (xwb_xw_load_map*, xwb_xw_refresh_obs), which redraw one env out of
 * turn.  Egocentric batches: XWB_ERR_STATE (their frames also depend on heading, poses and shadows).
 * xwb_xw_render_grids draws n_envs frames from such codes with THIS batch's tile table and frame format into
 * obs_dev [n_envs][bytes_per_env] -- n_envs is the caller's, not num_envs: the root draws the whole sharded batch with the
 * kernel that draws its own shard (xw_render_all_kernel), byte for byte what the shards drew. */
int xwb_xw_pack_grids(xwb_sim *sim, uint16_t *grids_dev, uint8_t *flags_dev, void *stream);
/* A batch whose frames are drawn elsewhere need not draw them itself: xwb_xw_set_draw(sim, 0) turns the pixel stores of
 * every verb off (the renders still run, as one workgroup or over the done list, for their bookkeeping: queue epochs,
 * installing pre-generated episodes, the fresh / done flags) -- the observation buffer is then stale, xwb_xw_pack_grids is
 * how the frames leave; on = 1 turns them back on (the next verb that draws every env -- xwb_reset, or a render from
 * xwb_xw_pack_grids + xwb_xw_render_grids into the batch's own buffer -- makes the buffer current again).  Full
 * observation only.  A step of the C4 batch is then the step kernel and the list pass: ~25 us instead of ~113. */
int xwb_xw_set_draw(xwb_sim *sim, int32_t on);
int xwb_xw_render_grids(xwb_sim *sim, const uint16_t *grids_dev, const uint8_t *flags_dev, int32_t n_envs, void *obs_dev,
                        void *stream);

/* (test and measurement hooks -- xwb_debug_stall_handoff, xwb_profile_begin / _end / _stop -- are not part of this boundary:
 * include/xwb_testing.h, version node XWB_TESTING of csrc/libxwb.map) */

/* ---- multi-GPU: one xwb_sim per GPU holds a shard of the batch (contiguous global env ids, xwb_config.env_gid0); the per-step
 * exchange is RCCL over xGMI, issued by the library itself so that C / C++ holders of a xwb_sim shard like the Python layer
 * does.  Replaces the reference's scale-out point: one OS process per environment behind SimulatorServer's TCP socket
 * (examples/demo_interface.cpp:67-95, simulator_interface.cpp:170-313).  RCCL is resolved at run time (the librccl.so.1
 * already loaded in the process, else the system's): libxwb.so has no link-time RCCL dependency. ----
 * xwb_comm = an RCCL communicator plus a stream of its own, on which the screens travel beside the caller's kernels. */
typedef struct xwb_comm xwb_comm;
#define XWB_COMM_ID_BYTES 128
int xwb_comm_version(int32_t *version);                                  /* ncclGetVersion of the RCCL in use */
int xwb_comm_unique_id(uint8_t out[XWB_COMM_ID_BYTES]);                  /* ncclGetUniqueId: rank 0 makes it, everyone gets a copy */
/* ncclCommInitRank on `device` (collective: every rank of the world calls it) */
int xwb_comm_init_rank(const uint8_t id[XWB_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, xwb_comm **out);
/* wrap a communicator (ncclComm_t) the caller created with the RCCL of this process; it is not destroyed with the object */
int xwb_comm_adopt(void *nccl_comm, int32_t device, xwb_comm **out);
int xwb_comm_destroy(xwb_comm *comm);
int xwb_comm_info(const xwb_comm *comm, int32_t *world, int32_t *rank);
/* ncclGroupStart / ncclGroupEnd: lets several shards that live on ONE rank (two batches on one device, a loopback
 * communicator) post their halves of an exchange as one group.  Shards that share the ROOT's rank must gather their
 * screens / grids inside such a group (a send to the own rank only matches a receive of the same group; without one the
 * calls return XWB_ERR_STATE).  Shards that share a rank call in ascending shard order (RCCL matches the sends and
 * receives between two ranks in issue order) and pass ONE all_dev buffer to xwb_gather_results between them: a shard never
 * receives the rows of a shard on its own rank, it finds them where that shard's call put them. */
int xwb_comm_group_start(xwb_comm *comm);
int xwb_comm_group_end(xwb_comm *comm);
/* A gather's layout: n_shards shards in global-env-id order, shard i = counts[i] envs held by communicator rank
 * peers[i] (peers == NULL: rank i, the usual one shard per rank); `shard` = the caller's own.
 * xwb_gather_results: every shard's packed (reward, game_over code) rows -- what xwb_bind_results receives, float[count][2] --
 * into float[sum(counts)][2] on every holder: one ncclAllGather (equal shards) or grouped ncclSend / ncclRecv, on `stream`.
 * Empty shards (counts[i] == 0) are legal: they send nothing and still receive everybody's rows. */
int xwb_gather_results(xwb_comm *comm, const float *packed_dev, float *all_dev, const int32_t *counts, const int32_t *peers,
                       int32_t n_shards, int32_t shard, void *stream);
/* The same exchange BESIDE the step loop (round 5): the rows the LAST xwb_step / xwb_step_autoreset call of `sim` wrote into
 * its results ring (xwb_bind_results / xwb_bind_results_ring: XWB_ERR_STATE without one, or before the first step) are
 * gathered on the COMMUNICATOR's stream, ordered behind that call's step kernel.  When the call handed over through epochs
 * (full-observation xworld batches on a probed stream: xwb_step_path) nothing at all is enqueued on `stream` -- a
 * one-wavefront kernel on the communicator's stream waits for the step's epoch, *by_epoch (nullable) = 1 --; otherwise one
 * event is recorded on `stream` (*by_epoch = 0).  A marker packet on the step's stream costs the loop ~6 us per step of idle
 * GPU (profiles/r5: forced one-rank exchange).  The caller keeps all_dev and the ring's row untouched until the exchange is
 * complete: xwb_comm_mark / xwb_comm_wait or xwb_gather_screens_end order a reader behind it; exchanges of one communicator
 * run in issue order.  Not inside an open group. */
int xwb_gather_results_beside(xwb_sim *sim, xwb_comm *comm, float *all_dev, const int32_t *counts, const int32_t *peers,
                              int32_t n_shards, int32_t shard, void *stream, int32_t *by_epoch);
/* The screens of every shard as ONE contiguous tensor on the root shard's GPU: dst_dev (root only; else NULL) =
 * [sum(counts)][bytes_per_env].  _begin orders the transfer behind the work already queued on `stream` (the step's render)
 * and issues it on the communicator's own stream -- the root posts one ncclRecv per remote shard into that shard's slice
 * (its own slab is copied unless xwb_bind_obs already points the batch at its slice), the others one ncclSend --, so it
 * runs beside whatever `stream` does next; _end makes `stream` wait for it.  With two observation buffers per batch
 * (xwb_bind_obs) the transfer of step t overlaps the kernels of step t + 1.  Each remote GPU reaches the root over ONE
 * xGMI link (~153.6 GB/s): a C4 shard (693.6 MB) needs >= 4.5 ms against 0.115 ms of compute (DESIGN.md "multi-GPU"). */
int xwb_gather_screens_begin(xwb_sim *sim, xwb_comm *comm, void *dst_dev, const int32_t *counts, const int32_t *peers,
                             int32_t n_shards, int32_t shard, int32_t root_shard, void *stream);
int xwb_gather_screens_end(xwb_comm *comm, void *stream);
/* The same tensor on the root from 1 / 216 of the bytes (full observation only): every shard ships its draw state
 * (xwb_xw_pack_grids: 2 * max_dim^2 + 1 bytes per env, packed on `stream` into a staging slab of the communicator) and the
 * root draws ALL frames into dst_dev with its own batch's render kernel (xwb_xw_render_grids) on the communicator's stream,
 * behind the receives.  C5: 242 B instead of 52 272 B per env cross a link -- the exchange stops being link-bound (a shard:
 * 7.9 MB, ~52 us) and the root's HBM write stream is the bound, as it is for any holder of one contiguous tensor
 * (DESIGN.md "multi-GPU").  Every shard still draws its own frames locally.  Same arguments and completion protocol as
 * xwb_gather_screens_begin (xwb_gather_screens_end, or xwb_comm_mark / xwb_comm_wait); two staging slabs alternate, a slab
 * is reused only after the transfer (root: the render) that read it.  context > 1: call after every frame-drawing verb. */
int xwb_gather_grids_begin(xwb_sim *sim, xwb_comm *comm, void *dst_dev, const int32_t *counts, const int32_t *peers,
                           int32_t n_shards, int32_t shard, int32_t root_shard, void *stream);
/* Frees the staging slabs `comm` keeps for `sim` (xwb_gather_grids_begin allocates them on first use, keyed by the batch):
 * call before xwb_destroy(sim) when the communicator outlives the batch -- otherwise they are only freed by xwb_comm_destroy.
 * Waits for the communicator's stream; XWB_ERR_STATE inside an open group. */
int xwb_comm_release_sim(xwb_comm *comm, const xwb_sim *sim);
/* Completion marks for pipelined gathers (two destination tensors alternating): xwb_comm_mark records mark `slot`
 * (0 .. XWB_COMM_MARKS - 1) on the communicator's stream behind everything begun so far -- inside an open group: when the
 * outermost group ends --; xwb_comm_wait orders `stream` behind that mark (no-op if the mark was never recorded).
 * xwb_gather_screens_end = mark + wait on a mark of its own. */
#define XWB_COMM_MARKS 4
int xwb_comm_mark(xwb_comm *comm, int32_t slot);
int xwb_comm_wait(xwb_comm *comm, int32_t slot, void *stream);

/* ---- the reference's thread-local RNG on the host (include/xwb_minstd.h): what XWB_RNG_MINSTD runs per env on the device ----
 * xwb_minstd_seed_thread: the engine state of the nth simulator thread under FLAGS_simulator_seed (simulator_util.cpp:44-52);
 * xwb_minstd_rand_ind / rand_range: util::get_rand_ind / get_rand_range_val on a caller-held state. */
uint32_t xwb_minstd_seed_thread(int32_t simulator_seed, int32_t nth_thread);
int32_t  xwb_minstd_rand_ind(uint32_t *state, int32_t size);
float    xwb_minstd_rand_range(uint32_t *state, float upper);

const char *xwb_last_error(void);
const char *xwb_version(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* XWB_H */
