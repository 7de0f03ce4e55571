/*
 * oracle/oracle.h -- CPU restatement of the PaddlePaddle/XWorld hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load liboracle.so; the product (xworld_amd/, libxwb.so) never links, imports
 * or calls it and fails loudly when its HIP library is missing.
 *
 * What this is: a plain-C, one-environment-at-a-time restatement of the
 * reference's GameSimulator::take_action / get_screen / reset_game path for
 * SimpleGame, SimpleRace and XWorld2D, written to follow the reference
 * function by function (every function cites the reference file:line it
 * restates; paths are relative to the reference root).  It deliberately keeps
 * the reference's own data shapes (per-env one-hot vector + reward vector,
 * entity lists, item stacks per cell, a 64 px canvas that is then resized) so
 * that it is an independent check of the SoA / bit-packed / tile-table HIP
 * product rather than a copy of it.
 *
 * Pinning status (see DESIGN.md "Oracle pinning"):
 *   - SimpleGame           pinned by tests/test_simple_game_simulator.cpp:21-47
 *   - minstd RNG           pinned by tests/test_simulator_seed.cpp:22-50
 *   - StatePacket wire     pinned by tests/test_statepacket.cpp:77-104 (round trip)
 *   - maze / BFS / maps / NavTarget teacher: pinned by golden vectors produced
 *     by importing the reference's Python modules in the build container
 *     (tests/golden/make_golden.py)
 *   - SimpleRace           no reference test exists and the reference C++
 *     cannot be built here without stand-in headers: pinned only by the
 *     known-answer values recorded in SURVEY.md 8(a) -> "parity partially pinned"
 *   - XWorld2D pixels      OpenCV 3.2.0 is absent: the resize / BGR2GRAY
 *     arithmetic is restated from the library's published algorithm ->
 *     "parity unpinned" for pixel values (bit-exact vs this restatement only)
 */
#ifndef XW_ORACLE_H
#define XW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* GameOverCode, simulator.h:42-48 */
enum { ORC_ALIVE = 0, ORC_MAX_STEP = 1, ORC_DEAD = 2, ORC_SUCCESS = 4, ORC_LOST_LIFE = 8 };

/* ---------------------------------------------------------------- RNG ---- */
/* libstdc++ minstd_rand0 + distributions as used by simulator_util.cpp:38-73 */
/* ---- trig.c: cos / sin of the SimpleRace and goal-warp call sites: the host's libm (default) or include/xwb_trig.h ---- */
void   orc_set_trig_libm(int on);
int    orc_get_trig_libm(void);
double orc_trig_cos(double x);
double orc_trig_sin(double x);
void   orc_xwb_sincos(double x, double *s, double *c);

typedef struct { uint32_t x; } orc_minstd;
void     orc_minstd_seed(orc_minstd *g, uint64_t s);            /* engine.seed(s) */
uint32_t orc_minstd_next(orc_minstd *g);                        /* engine()      */
int      orc_minstd_rand_ind(orc_minstd *g, int size);          /* get_rand_ind  */
float    orc_minstd_rand_range(orc_minstd *g, float upper);     /* get_rand_range_val */
uint64_t orc_std_hash_string(const char *s, size_t len);        /* std::hash<std::string> (libstdc++ murmur2-64, seed 0xc70f6907) */
/* seed of the n-th ThreadCounter (n = value of ++__num_threads) for FLAGS_simulator_seed */
void     orc_minstd_seed_thread(orc_minstd *g, int simulator_seed, int nth_thread);

/* Philox4x32-10 (Salmon et al. 2011) -- the *build's* counter-based stream
 * ("xwb-rng-v1"): no reference counterpart, restated here so the
 * GPU reset / random-policy streams can be replayed on the CPU. */
void     orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
typedef struct { uint32_t key[2]; uint32_t ctr[4]; uint32_t buf[4]; int have; } orc_stream;
void     orc_stream_init(orc_stream *s, uint32_t seed, uint32_t env_gid, uint32_t episode, uint32_t stream_id);
uint32_t orc_stream_u32(orc_stream *s);
uint32_t orc_stream_below(orc_stream *s, uint32_t n);  /* (u32 * n) >> 32 ; always consumes one draw */
float    orc_stream_unit(orc_stream *s);               /* (u32 >> 8) * 2^-24 in [0,1) */
int32_t  orc_policy_action(uint32_t policy_seed, uint32_t env_gid, uint32_t step, int num_actions);

/* ---------------------------------------------------------- SimpleGame ---- */
typedef struct orc_simple_game orc_simple_game;
orc_simple_game *orc_sg_create(int array_size, int max_steps, int context);
void    orc_sg_destroy(orc_simple_game *g);
void    orc_sg_reset_game(orc_simple_game *g);                 /* SimulatorInterface::reset_game ordering */
float   orc_sg_take_actions(orc_simple_game *g, int action, int act_rep);
int     orc_sg_game_over(const orc_simple_game *g);
int     orc_sg_get_lives(const orc_simple_game *g);
int64_t orc_sg_num_steps(const orc_simple_game *g);
int     orc_sg_pos(const orc_simple_game *g);
void    orc_sg_get_screen(const orc_simple_game *g, uint8_t *out /* array_size */);
void    orc_sg_get_state_screen(const orc_simple_game *g, uint8_t *out /* context*array_size */);

/* ---------------------------------------------------------- SimpleRace ---- */
typedef struct {
    int   track_type;         /* 0 straight, 1 circle */
    double track_width, track_length, track_radius;   /* gflags are doubles */
    int   race_full_manouver;
    int   random;
    int   difficulty_hard;    /* 0 easy, 1 hard */
    double reward_scale;
    int   max_steps;
    int   context;
    int   simulator_seed;     /* only used when random: minstd thread seed */
    int   nth_thread;
} orc_race_cfg;
typedef struct orc_simple_race orc_simple_race;
void    orc_race_default_cfg(orc_race_cfg *c);
orc_simple_race *orc_race_create(const orc_race_cfg *c);
void    orc_race_destroy(orc_simple_race *g);
void    orc_race_reset_game(orc_simple_race *g);
/* reset with externally supplied uniforms in [0,1) (the batched product draws them from Philox) */
void    orc_race_reset_game_with(orc_simple_race *g, float u_track, float u_dy, float u_dx, float u_angle);
float   orc_race_take_actions(orc_simple_race *g, int action, int act_rep);
int     orc_race_game_over(const orc_simple_race *g);
int     orc_race_get_lives(const orc_simple_race *g);
int     orc_race_num_actions(const orc_simple_race *g);
int64_t orc_race_num_steps(const orc_simple_race *g);
void    orc_race_get_car(const orc_simple_race *g, float *x, float *y, float *angle);
void    orc_race_set_car(orc_simple_race *g, float x, float y, float angle);
void    orc_race_get_screen(const orc_simple_race *g, float *out4);
void    orc_race_get_state_screen(const orc_simple_race *g, float *out /* context*4 */);

/* ------------------------------------------------------------ XWorld2D ---- */
/* icon table: one record per 64x64 icon of games/xworld/images */
typedef struct {
    int type;      /* 0 goal, 1 block, 2 agent  (xworld_env.py grid_types) */
    int name_id;   /* index into the sorted list of names of that type     */
    int colored;   /* properties.txt colour of this image != "na" (xworld_env.py:201-205) */
} orc_icon_info;

enum { ORC_MAP_NAV = 0, ORC_MAP_WALLS = 1 };
enum { ORC_EV_NONE = 0, ORC_EV_CORRECT = 1, ORC_EV_WRONG = 2, ORC_EV_TIMEUP = 3 };
enum { ORC_STAGE_IDLE = 0, ORC_STAGE_NAV = 1, ORC_STAGE_TERMINAL = 2 };
enum { ORC_TASKMODE_LANG_ACQ = 0, ORC_TASKMODE_ONE_CHANNEL = 1 };
/* tasks of the XWorld3DNav group in confs/navigation2d.json order */
enum { ORC_TASK_TARGET = 0, ORC_TASK_NEAR = 1, ORC_TASK_BETWEEN = 2, ORC_TASK_DIRECTION = 3, ORC_TASK_AVOID = 4,
       /* the 2-D-native group "XWorldNav" of confs/walls.json (games/xworld/tasks/XWorldNav*.py, rule D14b) */
       ORC_TASK2D_TARGET = 5, ORC_TASK2D_NEAR = 6, ORC_TASK2D_COLOR = 7, ORC_TASK2D_BETWEEN = 8 };

typedef struct {
    int map_kind;            /* ORC_MAP_NAV | ORC_MAP_WALLS */
    int max_dim;             /* max_height == max_width (XWorldNav 8, XWorldWalls 7) */
    int dim;                 /* actual dim this episode (== max_dim when curriculum == 0) */
    int num_goals, num_blocks;
    int max_steps;           /* FLAGS_max_steps */
    int max_steps_factor;    /* FLAGS_max_steps_factor (10) */
    int task_mode;           /* ORC_TASKMODE_* */
    int color;               /* FLAGS_color */
    int context;
    uint32_t seed;           /* xwb-rng-v1 seed */
    int visible_radius;      /* FLAGS_visible_radius: 0 = full observation; odd r > 0 = egocentric r x r view, 6 actions */
    int n_tasks;             /* tasks of the group, sampled uniformly per episode (teaching_task.cpp:204-213); 0 = {TARGET} */
    int tasks[8];            /* ORC_TASK_* in conf order */
    double curriculum;       /* FLAGS_curriculum: != 0 -> XWorldNav grows with the agent's success rate (XWorldNav.py:36-53) */
    int start_level;         /* XWorldNav(item_path, start_level): the level a --curriculum_stamp file holds (xworld.cpp:93-100) */
    int task_schedule;       /* 0 "random": util::get_rand_ind; 1 "weighted": util::simple_importance_sampling (teaching_task.cpp:204-213) */
    double task_weights[8];  /* TaskGroup::add_task weights, conf order */
    int no_wall_shadow;      /* FLAGS_wall_shadow = false (xmap.cpp:19,170) */
    int simulator_seed;      /* FLAGS_simulator_seed != 0: the decisions the reference takes with util::get_rand_ind /
                              * get_rand_range_val (the teacher's task draw, teaching_task.cpp:204-213) come from the env's
                              * own minstd_rand0, seeded like the reference's thread number thread_base + env id + 1
                              * (simulator_util.cpp:38-55); the xwb-rng-v1 draw they replace is still consumed */
    int thread_base;
    /* a second task group of the teacher, listed after the first in the conf (teacher.cpp:56-98 keeps conf order); 0 = none.
     * The groups run non-exclusively -- Teacher::teach's else branch (teacher.cpp:221-225), which lang_acquisition forces
     * (simulator_interface.cpp:46-48): every teach() runs each group's stage in conf order; rewards add up in the teacher
     * buffer, every task overwrites the buffer's event (also with ""), and only the first py_stage of a teach() sees this
     * step's collision events (XWorldSimulator::get_events_of_game clears them, xworld_simulator.cpp:118-122). */
    int n_tasks2;
    int tasks2[8];
    int task_schedule2;
    double task_weights2[8];
    /* FLAGS_task_groups_exclusive (teacher.cpp:22-24; ignored under lang_acquisition, simulator_interface.cpp:46-48):
     * Teacher::teach's exclusive branch (teacher.cpp:209-220) -- per teach() nondeterministic_sort_task_groups (:143-163)
     * re-sorts the group list in place by weighted sampling without replacement over group_weight (the conf's "weight" keys,
     * 0 when absent), then ONE group runs its stage: the last busy one of that order, else the first */
    int task_groups_exclusive;
    double group_weight[2];
} orc_xw_cfg;

typedef struct {
    int type;            /* 0 goal 1 block 2 agent */
    int x, y;
    int icon;            /* global icon index */
    int name_id;
    int serial;          /* running id -> "<name>_<serial>" */
} orc_entity;

typedef struct orc_xworld orc_xworld;
/* Task FSM of task group g (0 or 1) after the last call; ev = the event ITS task recorded in that call */
void   orc_xw_group_state(const orc_xworld *w, int g, int *kind, int *stage, int *steps_in_task, int *event,
                          int *target2d_x, int *target2d_y);
/* exclusive scheduling: the conf index of the group that heads Teacher::task_groups_ after the last sort */
int    orc_xw_group_first(const orc_xworld *w);
/* icons64: n_icons*64*64*3 BGR bytes (may be NULL when no rendering is asked for) */
orc_xworld *orc_xw_create(const orc_xw_cfg *cfg, int n_icons, const orc_icon_info *info,
                          const uint8_t *icons64);
void    orc_xw_destroy(orc_xworld *w);
/* SimulatorInterface::reset_game: map generation from stream (seed, env_gid, episode), teacher idle */
void    orc_xw_reset_game(orc_xworld *w, uint32_t env_gid, uint32_t episode);
/* replay an externally produced map (e.g. a reference-generated golden map), then teacher idle
 * with an explicit target pick (index into the reachable-goal list, or -1 to draw from the stream) */
void    orc_xw_load_map(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                        int target_pick, uint32_t env_gid, uint32_t episode);
/* golden replay of any task: `decisions` are consumed, in order, wherever the idle stage would draw below(n) */
void    orc_xw_load_map_ex(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                           const int *decisions, int n_decisions, uint32_t env_gid, uint32_t episode);
/* the same, but `decisions` (kept by the caller) stay installed for the idle stages that run at step time
 * (2-D-native tasks return to "idle"); orc_xw_forced_left tells how many are still unconsumed */
void    orc_xw_load_map_forced(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                               const int *decisions, int n_decisions, uint32_t env_gid, uint32_t episode);
int     orc_xw_forced_left(const orc_xworld *w);
/* 2-D-native tasks: the recorded target cell (C++ coordinates), (-1,-1) when none */
void    orc_xw_target2d(const orc_xworld *w, int *x, int *y);
/* egocentric mode: per-entity pose (xworld_env.py:207-223): yaw, scale, offset; entity index = order of orc_xw_get_entities */
void    orc_xw_set_pose(orc_xworld *w, int ent, double yaw, double scale, double offset);
void    orc_xw_get_pose(const orc_xworld *w, int ent, double *yaw, double *scale, double *offset);
double  orc_xw_agent_yaw(const orc_xworld *w);
/* poses (yaw, scale, offset per entity, in entity order) the NEXT orc_xw_load_map* applies before the teacher's idle stage */
void    orc_xw_stage_poses(orc_xworld *w, const double *poses, int n_entities);
/* XMap::image_masking for the agent: ROI origin (padded coordinates) and the r*r shadow flags */
void    orc_xw_agent_masking(const orc_xworld *w, int *x_st, int *y_st, uint8_t *shadow);
/* re-render after poses were set by hand (load_map draws no poses): init_screen */
void    orc_xw_refresh_screen(orc_xworld *w);
/* XMap::to_image(agent, false, visible_radius), xmap.cpp:125-206: the (r*64)^2 view before XWorldSimulator's two resizes,
 * interleaved B,G,R; and XItem::get_item_image (xitem.cpp:33-63) of one entity, 64 x 64 x 3 */
void    orc_xw_agent_view(const orc_xworld *w, uint8_t *view);
void    orc_xw_entity_image(const orc_xworld *w, int ent, uint8_t *out);
float   orc_xw_take_actions(orc_xworld *w, int action, int act_rep);
int     orc_xw_game_over(const orc_xworld *w);
int     orc_xw_get_lives(const orc_xworld *w);
int     orc_xw_num_actions(const orc_xworld *w);
int64_t orc_xw_num_steps(const orc_xworld *w);
int     orc_xw_last_action_success(const orc_xworld *w);
int     orc_xw_event(const orc_xworld *w);
int     orc_xw_stage(const orc_xworld *w);
int     orc_xw_target_name(const orc_xworld *w);
int     orc_xw_task_kind(const orc_xworld *w);
void    orc_xw_between_cell(const orc_xworld *w, int *x, int *y);
/* NavTargetDirection: self.target = (referent, direction): the referent's cell and the word (1 front 2 behind 3 left 4 right; 0 none) */
/* goal-name ids bound into the teacher's sentence by the idle stage (self._bind("G -> ...")); -1 = none */
void    orc_xw_sentence_names(const orc_xworld *w, int *a, int *b);
void    orc_xw_direction_target(const orc_xworld *w, int *x, int *y, int *word);
void    orc_xw_get_target_cells(const orc_xworld *w, uint8_t *out);
int     orc_xw_steps_in_task(const orc_xworld *w);
int     orc_xw_n_entities(const orc_xworld *w);
void    orc_xw_get_entities(const orc_xworld *w, orc_entity *out);
void    orc_xw_agent_xy(const orc_xworld *w, int *x, int *y);
/* grid of (icon+1) per cell, 0 = empty, top item of the stack; max_dim*max_dim ints, row-major [y][x] */
void    orc_xw_get_grid(const orc_xworld *w, int32_t *out);
void    orc_xw_screen_dims(const orc_xworld *w, int *h, int *wd, int *c);
/* XWorldSimulator::get_screen: canvas -> get_screen_rgb -> down_sample_image */
void    orc_xw_get_screen(const orc_xworld *w, uint8_t *out);
void    orc_xw_get_state_screen(const orc_xworld *w, uint8_t *out /* context * c*h*w */);

/* curriculum (FLAGS_curriculum != 0, XWorldNav only): level and check counter of the env; the pieces, for golden replays:
 * orc_xw_curriculum_configure = the level logic XWorldNav._configure runs at every reset (returns the level; dim, goals,
 * blocks of that level); orc_xw_record_result = XWorld(3D)Task.__record_result of the task class `kind` */
void    orc_xw_curriculum_state(const orc_xworld *w, int *level, int *counter);
int     orc_xw_curriculum_configure(orc_xworld *w, int *dim, int *num_goals, int *num_blocks);
void    orc_xw_record_result(orc_xworld *w, int kind, int result);

/* reference helper restatements exposed for golden-vector tests */
/* maze2d.spanning_tree_maze_generator with the shuffle decisions drawn from `s`;
 * maze: X*X chars, ' ' or '#', row-major [y][x] */
void    orc_maze_generate(orc_stream *s, int X, char *maze);
/* maze2d.bfs reachability (obstacles given as a mask X*Y, row-major [y][x]) */
int     orc_bfs_reachable(int sx, int sy, int ex, int ey, int X, int Y, const uint8_t *obstacle);

/* OpenCV 3.2 restatements (third-party, version pinned by cmake/opencv.cmake:5-6) */
void    orc_cv_get_rotation_matrix_2d(double cx, double cy, double angle_deg, double scale, double M[6]);
void    orc_cv_warp_affine_8uc3(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh, int dw, const double M[6],
                                const uint8_t border[3]);
void    orc_cv_resize_linear_8u(const uint8_t *src, int sh, int sw, int cn, uint8_t *dst, int dh, int dw);
void    orc_cv_bgr2gray_8u(const uint8_t *src, int n_pixels, uint8_t *dst);

/* -------------------------------------------------- StatePacket wire ---- */
/* data_packet.h:313-319, data_packet.cpp:143-174, memory_util.h:307-333 */
typedef struct {
    const char *key;
    int has_reals;  const float   *reals;  size_t n_reals;
    int has_pixels; const uint8_t *pixels; size_t n_pixels;
    int has_id;     const int32_t *id;     size_t n_id;
    int has_str;    const char    *str;
} orc_packet_field;
/* returns bytes needed; writes when out != NULL and cap is large enough */
size_t  orc_packet_encode(const orc_packet_field *fields, int n_fields, uint8_t *out, size_t cap);
/* decodes into caller arrays of at most max_fields; pointers alias `buf`; returns n_fields or -1 */
int     orc_packet_decode(const uint8_t *buf, size_t len, orc_packet_field *fields, int max_fields);

/* GameSimulator::decode_game_over_code, simulator.cpp:125-144 ; returns length written */
int     orc_decode_game_over_code(int code, char *out, int cap);

/* --------------------------------------------------- batch drivers ---- */
/* bench.py cpu_baseline + large parity tests: run n envs for `steps` steps of the
 * reference example loop (game_over? -> reset; get_state; random action; take_actions)
 * with the xwb-rng-v1 policy stream; returns number of env-steps executed.  */
typedef struct {
    double   reward_sum;
    uint64_t resets;
    /* xworld: what Teacher::report_task_performance sums up (teacher.cpp:175-200, teaching_task.h:22-36), over every env of the
     * rollout: per task class (ORC_TASK_*) successes, failures, success_steps (XWorld3DTask._record_success adds
     * steps_in_cur_task; the 2-D-native tasks keep none), and the failures that were time-ups */
    int64_t  task_perf[9][4];
} orc_rollout_stats;
/* the same tallies of one orc_xworld since it was created */
void orc_xw_get_performance(const orc_xworld *w, int64_t out[9][4]);
/* optional per-step outputs, layout [steps][n_envs] (pass NULL to skip):
 *   rewards  float   return value of take_actions
 *   codes    uint8   game_over() after the step
 *   obs_ck   uint64  position-weighted checksum of the observation get_state() returned *before*
 *                    the step:  sum_i obs_byte[i] * ((i+1) * 0x9E3779B97F4A7C15)  mod 2^64        */
typedef struct { float *rewards; uint8_t *codes; uint64_t *obs_ck; } orc_rollout_out;
uint64_t orc_obs_checksum(const void *obs, size_t n_bytes);
uint64_t orc_sg_rollout(int n_envs, int array_size, int context, int steps, uint32_t policy_seed,
                        uint32_t env_gid0, orc_rollout_stats *st, const orc_rollout_out *out);
uint64_t orc_race_rollout(int n_envs, const orc_race_cfg *cfg, uint32_t seed, int steps, uint32_t policy_seed,
                          uint32_t env_gid0, orc_rollout_stats *st, const orc_rollout_out *out);
uint64_t orc_xw_rollout(int n_envs, const orc_xw_cfg *cfg, int n_icons, const orc_icon_info *info,
                        const uint8_t *icons64, int steps, uint32_t policy_seed,
                        uint32_t env_gid0, int render, orc_rollout_stats *st, const orc_rollout_out *out);

#ifdef __cplusplus
}
#endif
#endif /* XW_ORACLE_H */
