/* oracle/xworld_internal.h -- TEST INFRASTRUCTURE ONLY: shared between xworld2d.c and xworld_tasks.c */
#ifndef XW_ORACLE_INTERNAL_H
#define XW_ORACLE_INTERNAL_H
#include "oracle.h"

#define MAXDIM   16
#define MAXCELLS (MAXDIM * MAXDIM)
#define MAXENT   (MAXCELLS + 8)
#define MAXSTACK 4
#define ITEM_SIZE 64          /* XItem::item_size_, xitem.h:151 */

/* the Task FSM fields of one task group (teaching_task.h:63-69 + the Python task's own fields); the group that is running
 * has them in orc_xworld's working fields, teacher_teach() swaps */
typedef struct {
    int stage, steps_in_cur_task, target_name, task_kind;
    uint8_t target_ent[MAXENT];
    int between_x, between_y, sent_a, sent_b, dir_ref_ent, dir_word, target2d_x, target2d_y;
    int last_event;
} orc_group_state;

struct orc_xworld {
    orc_xw_cfg cfg;
    orc_group_state grp[2];          /* saved FSMs; the working fields below always end a call holding group 0's */
    int n_groups;
    int grp_order[2];                /* Teacher::task_groups_ as the last sort left it (exclusive scheduling): conf indices.
                                      * Lives as long as the teacher: across game resets; a new env (episode 0) starts in conf order */
    /* the running group's task list (orc_task_idle) */
    int act_n_tasks, act_schedule;
    const int *act_tasks;
    const double *act_weights;
    orc_minstd reng;                 /* cfg.simulator_seed != 0: this env's thread-local engine */
    int n_icons;
    orc_icon_info *info;
    const uint8_t *icons64;          /* borrowed */
    /* per type: names and their icon variants (xworld_env.py:247-255 set_goal_subtrees) */
    int n_names[3];
    int *name_variants[3];           /* flattened icon ids grouped by name */
    int *name_first[3];              /* offsets, n_names+1 */
    /* XWorld (xworld.h): item list, map */
    orc_entity ents[MAXENT];
    double e_yaw[MAXENT], e_scale[MAXENT], e_offset[MAXENT];   /* Entity.yaw / scale / offset (xworld_env.py:42) */
    int n_ents;
    int agent_idx;
    int height, width;               /* max dims: what C++ sees (get_max_dims) */
    int actual_h, actual_w, offset_h, offset_w;
    int cube[MAXDIM][MAXDIM][MAXSTACK];   /* XMap::item_ptr_cube_ (entity indices) */
    int cube_n[MAXDIM][MAXDIM];
    int running_id;
    /* XWorldSimulator */
    int hits[MAXENT]; int n_hits;    /* ids in game_events_ ("collision:a|b\n" lines) */
    int last_action_success;
    /* TeachingEnvBuffer (simulator.h:265-292) */
    double teacher_reward;
    int event;
    /* Task FSM (teaching_task.h:63-69) + XWorld3DTask fields */
    int stage;
    int steps_in_cur_task;
    int target_name;                 /* NavTarget: name id of the picked goal (introspection) */
    int task_kind;                   /* ORC_TASK_* of the busy task */
    uint8_t target_ent[MAXENT];      /* self.target: entity is a target goal */
    int between_x, between_y;        /* NavTargetBetween: the middle cell (C++ coordinates) */
    int sent_a, sent_b;              /* goal-name ids the idle stage binds into the task's grammar (G / G1, G2); -1 none */
    int dir_ref_ent, dir_word;       /* NavTargetDirection: self.target = (referent, direction) */
    int target2d_x, target2d_y;      /* 2-D-native tasks: XWorldTask.target (C++ coordinates), -1 = none */
    uint32_t env_gid, episode;       /* of the running episode: step-time idle stages draw from stream 2 */
    int forced_sticky;
    const double *staged_poses; int n_staged_poses;
    const int *forced; int n_forced, forced_at;   /* golden replay: decisions instead of stream draws */
    /* curriculum: XWorldEnv.current_level / curriculum_check_counter (xworld_env.py:73-77); per task class the
     * success_seq window of the last 200 results (xworld3d_task.py:129-146) -- current_usage holds a class once it recorded */
    int64_t perf[9][4];              /* XWorld(3D)Task.num_successes / num_failures / success_steps (+ time-ups), per task class */
    int cur_level, cur_counter;
    int use_len[9], use_sum[9], use_head[9];
    uint8_t use_bits[9][200];
    /* GameSimulator */
    int64_t num_steps;
    uint8_t *screens;
    int img_h_out, img_w_out, channels;
    orc_stream rs;
};


/* xworld2d.c */
/* xworld_ego.c */
int  orc_facing_dir(double yaw);                        /* XItem::get_item_facing_dir: 0 right 1 down 2 left 3 up */
void orc_xw_image_masking(const orc_xworld *w, int ax, int ay, double yaw, int r, int *x_st, int *y_st, uint8_t *shadow);
void orc_xw_ego_view(const orc_xworld *w, int r, uint8_t *view);
void orc_xw_item_image(const orc_xworld *w, int ent, uint8_t *out);   /* XItem::get_item_image */
void orc_xw_rebuild_map(orc_xworld *w);                 /* XWorld::reset(false): rebuild the cube from the entity list */
int  orc_xw_draw_below(orc_xworld *w, int n);           /* next decision: forced (golden replay) or stream draw */
/* xworld_tasks.c */
void orc_xw_record_result(orc_xworld *w, int kind, int result);
void orc_task_idle(orc_xworld *w);                      /* TaskGroup::run_stage: sample a task, run its idle stage */
int  orc_task_is_target(const orc_xworld *w, int ent);
void orc_task2d_navigation_reward(orc_xworld *w);       /* XWorldTask.simple_navigation_reward */
int  orc_task_reachable_ex(const orc_xworld *w, int goal_ent, int goals_are_obstacles);
#endif
