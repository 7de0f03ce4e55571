/*
 * oracle/xworld_tasks.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The idle stages of the five tasks of the XWorld3DNav group (confs/navigation2d.json:9-17), restated from
 *   games/xworld3d/tasks/XWorld3DNavTarget.py:28-43        target = goals named like a reachable goal
 *   games/xworld3d/tasks/XWorld3DNavTargetNear.py:28-62     two goals moved onto a "pair tile", target = goals around g1
 *   games/xworld3d/tasks/XWorld3DNavTargetBetween.py:28-62  two goals moved to the ends of a T tile, target = the middle
 *   games/xworld3d/tasks/XWorld3DNavTargetDirection.py:28-130 two goals on an L tile, target = goal left/right/front of the referent
 *   games/xworld3d/tasks/XWorld3DNavTargetAvoid.py:28-43    target = every goal not named like the referent
 * with their helpers in xworld3d_task.py (_get_p_tiles :226-250, _get_t_tiles :252-273, _get_l_tiles :300-322,
 * _middle_loc :324-326, _propagate_agent :343-353, _get_surrounding_goals/_empty_grids :190-224,
 * _get_direction_and_distance :98-124), maze2d.flood_fill (python/maze2d.py:21-40), the env edits
 * delete_entity / set_entity_inst (xworld_env.py:227-242) and the task sampling of TaskGroup::run_stage
 * (teaching_task.cpp:204-222, schedule "random": get_rand_ind(#tasks)).
 *
 * Everything works in the Python env's coordinates (cell = loc - padding offset).  Random decisions come from
 * orc_xw_draw_below(): the xwb-rng-v1 stream, or a forced list when a golden trace is replayed.  Decision order
 * ("xwb-taskgen-v1"):
 *   task   = below(#tasks)
 *   TARGET : below(#reachable goals)
 *   AVOID  : below(#reachable goals), below(#goals with another name)
 *   NEAR / BETWEEN / DIRECTION : HEAD2(#goals) -> (g1, g2); HEAD2(#tiles) -> tile; [DIRECTION: below(#empty
 *            neighbours, row-major)]; below(#agent cells in flood-fill order)
 *   HEAD2(n) = the first two positions of a Fisher-Yates shuffle: below(n), then below(n-1) if n >= 2
 *            (random.shuffle(lst) followed by lst[:2] or lst[0]).
 * Where the reference asserts ("map too crowded?") the episode keeps its map and has no target.
 */
#include "xworld_internal.h"
#include <math.h>
#include <string.h>

typedef struct { int x, y; } pcell;
typedef struct { pcell a, b; } ptile;

typedef struct {
    orc_xworld *w;
    int X, Y;
    uint8_t avail[MAXCELLS];       /* env.available_grids as a set */
    int goals[MAXENT], ng;         /* env.get_goals(): entity indices in entity order */
} penv;

static int px(const penv *p, int e) { return p->w->ents[e].x - p->w->offset_w; }
static int py_(const penv *p, int e) { return p->w->ents[e].y - p->w->offset_h; }
static int in_board(const penv *p, int x, int y) { return x >= 0 && y >= 0 && x < p->X && y < p->Y; }
static int is_avail(const penv *p, int x, int y) { return in_board(p, x, y) && p->avail[y * p->X + x]; }

/* update_entities_from_cpp (xworld_env.py:386-404): entities inside the actual dims, available = the rest */
static void penv_init(penv *p, orc_xworld *w) {
    p->w = w; p->X = w->actual_w; p->Y = w->actual_h; p->ng = 0;
    memset(p->avail, 0, sizeof p->avail);
    for (int c = 0; c < p->X * p->Y; ++c) p->avail[c] = 1;
    for (int i = 0; i < w->n_ents; ++i) {
        int x = w->ents[i].x - w->offset_w, y = w->ents[i].y - w->offset_h;
        if (!in_board(p, x, y)) continue;                      /* padding block */
        p->avail[y * p->X + x] = 0;
        if (w->ents[i].type == 0) p->goals[p->ng++] = i;
    }
}

static void set_loc(penv *p, int e, int x, int y) {
    p->w->ents[e].x = x + p->w->offset_w;
    p->w->ents[e].y = y + p->w->offset_h;
}

/* random.shuffle(lst); lst[:2] */
static void head2(orc_xworld *w, int n, int *d0, int *d1) {
    *d0 = orc_xw_draw_below(w, n);
    *d1 = n >= 2 ? orc_xw_draw_below(w, n - 1) : 0;
}

/* _get_surrounding_empty_grids(distance_threshold=1.0, refer): available 4-neighbours, row-major order */
static int empty_neighbours(const penv *p, pcell c, pcell *out) {
    int n = 0;
    static const int d[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};    /* (y,x)-sorted: up, left, right, down */
    for (int k = 0; k < 4; ++k) {
        int x = c.x + d[k][0], y = c.y + d[k][1];
        if (is_avail(p, x, y)) { out[n].x = x; out[n].y = y; n++; }
    }
    return n;
}

static int has_other_empty_neighbour(const penv *p, pcell c, pcell except) {
    pcell nb[4];
    int n = empty_neighbours(p, c, nb);
    for (int i = 0; i < n; ++i) if (!(nb[i].x == except.x && nb[i].y == except.y)) return 1;
    return 0;
}

/* _get_p_tiles, xworld3d_task.py:226-250 */
static int p_tiles(const penv *p, ptile *out) {
    int n = 0;
    static const int d[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int y = 0; y < p->Y; ++y)
        for (int x = 0; x < p->X; ++x)
            for (int k = 0; k < 3; ++k) {
                pcell p1 = {x, y}, p2 = {x + d[k][0], y + d[k][1]};
                if (is_avail(p, p1.x, p1.y) && is_avail(p, p2.x, p2.y)) {
                    if (has_other_empty_neighbour(p, p2, p1)) { out[n].a = p1; out[n].b = p2; n++; }
                    if (has_other_empty_neighbour(p, p1, p2)) { out[n].a = p2; out[n].b = p1; n++; }
                }
            }
    return n;
}

/* _get_t_tiles, xworld3d_task.py:252-273 */
static int t_tiles(const penv *p, ptile *out) {
    int n = 0;
    for (int y = 0; y < p->Y; ++y)
        for (int x = 0; x < p->X; ++x) {
            if (!is_avail(p, x, y)) continue;
            if (is_avail(p, x - 1, y) && is_avail(p, x + 1, y) && (is_avail(p, x, y - 1) || is_avail(p, x, y + 1))) {
                out[n].a.x = x - 1; out[n].a.y = y; out[n].b.x = x + 1; out[n].b.y = y; n++;
            }
            if (is_avail(p, x, y - 1) && is_avail(p, x, y + 1) && (is_avail(p, x - 1, y) || is_avail(p, x + 1, y))) {
                out[n].a.x = x; out[n].a.y = y - 1; out[n].b.x = x; out[n].b.y = y + 1; n++;
            }
        }
    return n;
}

/* _get_l_tiles, xworld3d_task.py:300-322 (straight triples, the diagonal test is commented out upstream) */
static int l_tiles(const penv *p, ptile *out) {
    int n = 0;
    static const int d[2][2] = {{0, 1}, {1, 0}};
    for (int y = 0; y < p->Y; ++y)
        for (int x = 0; x < p->X; ++x)
            for (int k = 0; k < 2; ++k) {
                pcell p1 = {x, y}, p2 = {x + d[k][0], y + d[k][1]}, p3 = {x + 2 * d[k][0], y + 2 * d[k][1]};
                if (is_avail(p, p1.x, p1.y) && is_avail(p, p2.x, p2.y) && is_avail(p, p3.x, p3.y)) {
                    out[n].a = p1; out[n].b = p2; n++;
                    out[n].a = p2; out[n].b = p3; n++;
                }
            }
    return n;
}

/* _propagate_agent(seeds=[seed], inclusive) = maze2d.flood_fill over blocks + goals, sorted by step (stable) */
static int propagate_agent(const penv *p, pcell seed, int inclusive, pcell *out) {
    uint8_t obst[MAXCELLS], seen[MAXCELLS];
    memset(obst, 0, sizeof obst);
    memset(seen, 0, sizeof seen);
    const orc_xworld *w = p->w;
    for (int i = 0; i < w->n_ents; ++i) {
        if (w->ents[i].type == 2) continue;                               /* the agent has been deleted */
        int x = w->ents[i].x - w->offset_w, y = w->ents[i].y - w->offset_h;
        if (in_board(p, x, y)) obst[y * p->X + x] = 1;
    }
    pcell queue[MAXCELLS];
    int head = 0, tail = 0, n = 0;
    if (inclusive) out[n++] = seed;                                       /* (seed, 0) sorts first */
    queue[tail++] = seed;
    seen[seed.y * p->X + seed.x] = 1;
    static const int mv[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
    while (head < tail) {
        pcell c = queue[head++];
        for (int m = 0; m < 4; ++m) {
            int x = c.x + mv[m][0], y = c.y + mv[m][1];
            if (in_board(p, x, y) && !seen[y * p->X + x] && !obst[y * p->X + x]) {
                seen[y * p->X + x] = 1;
                queue[tail].x = x; queue[tail].y = y; tail++;
                out[n].x = x; out[n].y = y; n++;
            }
        }
    }
    return n;
}

/* XWorld3DNavTargetDirection.__compute_triple_direction (:110-130) for the 2-D env ("opposite left and right"):
 * 0 = False, 1 = front, 2 = behind, 3 = left, 4 = right */
static int triple_direction(pcell target, pcell referent, double view_yaw) {
    const double PI = 3.1415926, PI_2 = PI / 2, PI_4 = PI / 4;
    double dx = referent.x - target.x, dy = referent.y - target.y;
    double dist = sqrt(dx * dx + dy * dy);
    if (dist == 0) return 0;
    double v1x = cos(view_yaw), v1y = sin(view_yaw);
    double v2x = dx / dist, v2y = dy / dist;
    double cos_theta = fmax(-1, fmin(1, v1x * v2x + v1y * v2y));
    double sin_theta = fmax(-1, fmin(1, v1y * v2x - v1x * v2y));
    double theta = acos(cos_theta) * copysign(1, asin(sin_theta));
    int sign = theta > 0;
    int far = 0;
    theta = fabs(theta);
    if (theta > PI_2) { far = 1; theta = PI - theta; }
    if (theta < PI_4 + 1e-3) return far ? 2 : 1;
    if (PI_2 - theta < PI_4 + 1e-3) return sign ? 4 : 3;
    return 0;
}

/* t in self.target.  NavTargetDirection evaluates (direction(g, referent, agent.yaw), near) when g is reached
 * (XWorld3DNavTargetDirection.py:74-89): with the egocentric agent the yaw -- and so the answer -- changes over time */
int orc_task_is_target(const orc_xworld *w, int ent) {
    if (w->task_kind == ORC_TASK_DIRECTION && w->dir_word != 0 && w->dir_ref_ent >= 0 && w->ents[ent].type == 0) {
        pcell gl = {w->ents[ent].x, w->ents[ent].y}, rl = {w->ents[w->dir_ref_ent].x, w->ents[w->dir_ref_ent].y};
        double ddx = gl.x - rl.x, ddy = gl.y - rl.y;
        int near = sqrt(ddx * ddx + ddy * ddy) < 1.0 + 1e-3;
        return near && triple_direction(gl, rl, w->e_yaw[w->agent_idx]) == w->dir_word;
    }
    return w->target_ent[ent];
}

/* env.entities order after delete_entity(x) ... set_entity_inst(x): x moves to the end of the list */
static void move_to_end(orc_xworld *w, int *tracked, int n_tracked, int e) {
    orc_entity tmp = w->ents[e];
    uint8_t flag = w->target_ent[e];
    double pose[3] = {w->e_yaw[e], w->e_scale[e], w->e_offset[e]};
    for (int i = e; i + 1 < w->n_ents; ++i) {
        w->ents[i] = w->ents[i + 1]; w->target_ent[i] = w->target_ent[i + 1];
        w->e_yaw[i] = w->e_yaw[i + 1]; w->e_scale[i] = w->e_scale[i + 1]; w->e_offset[i] = w->e_offset[i + 1];
    }
    w->ents[w->n_ents - 1] = tmp;
    w->target_ent[w->n_ents - 1] = flag;
    w->e_yaw[w->n_ents - 1] = pose[0]; w->e_scale[w->n_ents - 1] = pose[1]; w->e_offset[w->n_ents - 1] = pose[2];
    for (int k = 0; k < n_tracked; ++k) {
        if (tracked[k] == e) tracked[k] = w->n_ents - 1;
        else if (tracked[k] > e) tracked[k] -= 1;
    }
}

int orc_task_reachable(const orc_xworld *w, int goal_ent);     /* xworld2d.c */

static int reachable_goals(const orc_xworld *w, const penv *p, int *cand) {
    int nc = 0;
    for (int k = 0; k < p->ng; ++k) if (orc_task_reachable(w, p->goals[k])) cand[nc++] = p->goals[k];
    return nc;
}

/* the three tasks that rearrange the map share everything but the tile kind and the target rule */
static void idle_rearranging(orc_xworld *w, penv *p, int kind) {
    int agent = w->agent_idx;
    pcell a0 = {px(p, agent), py_(p, agent)};
    if (p->ng < 2) return;                                               /* assert len(goals) >= 2 */
    p->avail[a0.y * p->X + a0.x] = 1;                                    /* self._delete_entity(agent) */
    int d0, d1;
    head2(w, p->ng, &d0, &d1);                                           /* random.shuffle(goals); g1, g2 = goals[:2] */
    int g1 = p->goals[d0];
    int rest[MAXENT], nr = 0;
    for (int k = 0; k < p->ng; ++k) if (k != d0) rest[nr++] = p->goals[k];
    int g2 = rest[d1];
    pcell o1 = {px(p, g1), py_(p, g1)}, o2 = {px(p, g2), py_(p, g2)};
    p->avail[o1.y * p->X + o1.x] = 1;                                    /* delete g1, g2 "to make space" */
    p->avail[o2.y * p->X + o2.x] = 1;
    ptile tiles[MAXCELLS * 6];                                           /* on the stack: rollouts run in parallel threads */
    int nt = kind == ORC_TASK_NEAR ? p_tiles(p, tiles) : (kind == ORC_TASK_BETWEEN ? t_tiles(p, tiles) : l_tiles(p, tiles));
    if (nt == 0) goto crowded;
    {
        int t0, t1;
        head2(w, nt, &t0, &t1);                                          /* random.shuffle(tiles); tiles[0] */
        pcell l1 = tiles[t0].a, l2 = tiles[t0].b;
        set_loc(p, g1, l1.x, l1.y); p->avail[l1.y * p->X + l1.x] = 0;    /* _set_entity_inst(g1), (g2) */
        set_loc(p, g2, l2.x, l2.y); p->avail[l2.y * p->X + l2.x] = 0;
        pcell seed;
        int inclusive = 0, target = g1, referent = g2, direction = 0;
        if (kind == ORC_TASK_NEAR) {
            seed = l2;                                                   /* _propagate_agent([g2.loc]) */
        } else if (kind == ORC_TASK_BETWEEN) {
            seed.x = (l1.x + l2.x) / 2; seed.y = (l1.y + l2.y) / 2;      /* _middle_loc (Python-2 integer division) */
        } else {
            pcell eg[4];
            int ne = empty_neighbours(p, l1, eg);
            if (ne == 0) { ne = empty_neighbours(p, l2, eg); target = g2; referent = g1; }
            if (ne == 0) goto crowded_placed;                            /* assert empty_grids, "get_l_tiles() is buggy" */
            pcell e = eg[orc_xw_draw_below(w, ne)];                      /* random.choice(empty_grids) */
            pcell tl = target == g1 ? l1 : l2, rl = referent == g1 ? l1 : l2;
            direction = triple_direction(tl, rl, atan2((double)(tl.y - e.y), (double)(tl.x - e.x)));
            seed = e; inclusive = 1;                                     /* _propagate_agent([e], inclusive=True) */
        }
        pcell cells[MAXCELLS + 1];
        int na = propagate_agent(p, seed, inclusive, cells);
        if (na == 0) goto crowded_placed;                                /* assert new_a */
        pcell al = cells[orc_xw_draw_below(w, na)];                      /* agent.loc, _ = random.choice(new_a) */
        set_loc(p, agent, al.x, al.y);
        /* self._record_target(...) */
        memset(w->target_ent, 0, sizeof w->target_ent);
        if (kind == ORC_TASK_NEAR) {
            /* _get_surrounding_goals(refer=g1.loc): dist < 1.5 + 1e-3, goals AT the referent location skipped */
            for (int k = 0; k < p->ng; ++k) {
                int g = p->goals[k];
                int gx = px(p, g), gy = py_(p, g);
                if (gx == l1.x && gy == l1.y) continue;
                double ddx = gx - l1.x, ddy = gy - l1.y;
                if (sqrt(ddx * ddx + ddy * ddy) < 1.5 + 1e-3) w->target_ent[g] = 1;
            }
        } else if (kind == ORC_TASK_BETWEEN) {
            w->between_x = seed.x + w->offset_w; w->between_y = seed.y + w->offset_h;
        } else {
            /* navigation_reward (:74-89): a reached goal g wins iff (direction(g, referent, agent yaw), near) ==
             * (direction, True); static because full-observation yaw is the constant 1.5707963 */
            pcell rl = referent == g1 ? l1 : l2;
            for (int k = 0; k < p->ng; ++k) {
                int g = p->goals[k];
                pcell gl = {px(p, g), py_(p, g)};
                double ddx = gl.x - rl.x, ddy = gl.y - rl.y;
                int near = sqrt(ddx * ddx + ddy * ddy) < 1.0 + 1e-3;
                if (near && direction != 0 && triple_direction(gl, rl, w->e_yaw[agent]) == direction) w->target_ent[g] = 1;
            }
        }
        /* env.entities: g1, g2, agent were deleted and re-added -> they move to the end, in that order */
        int tr[3] = {g1, g2, agent};
        move_to_end(w, tr, 3, tr[0]);
        move_to_end(w, tr, 3, tr[1]);
        move_to_end(w, tr, 3, tr[2]);
        if (kind == ORC_TASK_DIRECTION) { w->dir_ref_ent = referent == g1 ? tr[0] : tr[1]; w->dir_word = direction; }
        /* Near: G -> g1.name; Between: G1 -> g1.name, G2 -> g2.name; Direction: G -> referent.name */
        w->sent_a = w->ents[kind == ORC_TASK_DIRECTION ? (referent == g1 ? tr[0] : tr[1]) : tr[0]].name_id;
        if (kind == ORC_TASK_BETWEEN) w->sent_b = w->ents[tr[1]].name_id;
        orc_xw_rebuild_map(w);                                           /* env_changed -> XWorld::reset(false) */
        return;
    }
crowded_placed:
    set_loc(p, g1, o1.x, o1.y);
    set_loc(p, g2, o2.x, o2.y);
crowded:
    /* reference: assert ..., "map too crowded?" -- keep the generated map, no target */
    memset(w->target_ent, 0, sizeof w->target_ent);
    orc_xw_rebuild_map(w);
}

/* ---- the 2-D-native group (confs/walls.json "XWorldNav"; rule D14b) -------------------------------------------
 * XWorldNavTarget.idle (XWorldNavTarget.py:22-33), XWorldNavColorTarget.idle (:8-20): targets = [coloured] goals
 * the agent can reach with only the BLOCKS as obstacles; random.choice -> self.target = that goal's loc.
 * XWorldNavNear.idle / XWorldNavBetween.idle build their candidate cells as 2-tuples while entity locs are
 * 3-tuples (xworld_task.py:327,341-342 vs xworld_env.py:362), so bfs() never meets its `end` and no target is
 * ever found: both stay in "idle" with reward 0 (reproduced by tests/golden/tasks2d.json). */
static void idle_2d(orc_xworld *w, penv *p) {
    w->steps_in_cur_task = 0;                                            /* TaskGroup::run_stage: busy_task_->reset() */
    w->target2d_x = w->target2d_y = -1;
    w->teacher_reward += 0.0;
    w->stage = ORC_STAGE_IDLE;
    if (w->task_kind != ORC_TASK2D_TARGET && w->task_kind != ORC_TASK2D_COLOR) return;
    int cand[MAXENT], nc = 0;
    for (int k = 0; k < p->ng; ++k) {
        int g = p->goals[k];
        if (w->task_kind == ORC_TASK2D_COLOR && !w->info[w->ents[g].icon].colored) continue;   /* _get_colored_goals */
        if (orc_task_reachable_ex(w, g, 0)) cand[nc++] = g;
    }
    if (nc == 0) return;                                                 /* ["idle", 0, ""] */
    int sel = cand[orc_xw_draw_below(w, nc)];                            /* random.choice(targets) */
    w->target2d_x = w->ents[sel].x; w->target2d_y = w->ents[sel].y;     /* self._record_target(sel_goal.loc) */
    w->stage = ORC_STAGE_NAV;                                            /* ["simple_navigation_reward", 0.0, ...] */
}

/* XWorldTask.simple_navigation_reward, games/xworld/tasks/xworld_task.py:184-223 */
void orc_task2d_navigation_reward(orc_xworld *w) {
    double reward = -0.1;                                                /* time_penalty */
    if (!w->last_action_success) reward += -0.2;                         /* failed_action_penalty */
    const orc_entity *a = &w->ents[w->agent_idx];
    int next_stage = ORC_STAGE_NAV;
    w->steps_in_cur_task += 1;
    int on_goal = 0;
    for (int i = 0; i < w->n_ents; ++i)
        if (w->ents[i].type == 0 && w->ents[i].x == a->x && w->ents[i].y == a->y) on_goal = 1;
    if (w->cfg.task_mode == ORC_TASKMODE_ONE_CHANNEL &&
        w->steps_in_cur_task >= w->height * w->width / 2) {             /* get_max_dims(); Python-2 int division */
        w->steps_in_cur_task = 0;
        orc_xw_record_result(w, w->task_kind, 0);                        /* _record_failure */
        w->perf[w->task_kind][3] += 1;                                   /* (a time-up) */
        next_stage = ORC_STAGE_IDLE;                                     /* "S -> timeup" */
    } else if (a->x == w->target2d_x && a->y == w->target2d_y) {
        w->steps_in_cur_task = 0;
        orc_xw_record_result(w, w->task_kind, 1);                        /* _record_success */
        w->event = ORC_EV_CORRECT;
        reward += 1.0;
        next_stage = ORC_STAGE_IDLE;
    } else if (on_goal) {
        reward += -1.0;
    }
    w->teacher_reward += reward;
    w->stage = next_stage;
}

void orc_task_idle(orc_xworld *w) {
    penv p;
    penv_init(&p, w);
    memset(w->target_ent, 0, sizeof w->target_ent);
    w->between_x = w->between_y = -1;
    w->target_name = -1;
    w->dir_ref_ent = -1; w->dir_word = 0;
    w->sent_a = w->sent_b = -1;
    /* TaskGroup::run_stage: idx = get_rand_ind(task_list_.size()) */
    int n_tasks = w->act_n_tasks > 0 ? w->act_n_tasks : 1;
    int t;
    if (w->act_schedule == 1 && w->act_n_tasks > 0) {
        /* util::simple_importance_sampling (simulator_util.cpp:57-86): float uniform in [0, float(acc.back())), the first
         * task whose accumulated weight is >= it; the draw is the 24-bit integer behind orc_stream_unit */
        double acc[8], total = 0;
        for (int i = 0; i < n_tasks; ++i) { total += w->act_weights[i]; acc[i] = total; }
        float val = ((float)orc_xw_draw_below(w, 1 << 24) * (1.0f / 16777216.0f)) * (float)total;
        if (w->cfg.simulator_seed) val = orc_minstd_rand_range(&w->reng, (float)total);      /* the reference's own engine */
        t = n_tasks - 1;
        for (int i = 0; i < n_tasks; ++i) if ((double)val <= acc[i]) { t = i; break; }
    } else {
        t = orc_xw_draw_below(w, n_tasks);
        if (w->cfg.simulator_seed) t = orc_minstd_rand_ind(&w->reng, n_tasks);
    }
    w->task_kind = w->act_n_tasks > 0 ? w->act_tasks[t] : ORC_TASK_TARGET;
    if (w->task_kind >= ORC_TASK2D_TARGET) { idle_2d(w, &p); return; }
    if (w->task_kind == ORC_TASK_TARGET || w->task_kind == ORC_TASK_AVOID) {
        int cand[MAXENT];
        int nc = reachable_goals(w, &p, cand);
        if (nc > 0) {                                                    /* else: assert targets, "map too crowded?" */
            int sel = cand[orc_xw_draw_below(w, nc)];                    /* sel_goal = random.choice(targets) */
            if (w->task_kind == ORC_TASK_TARGET) {
                w->target_name = w->ents[sel].name_id;
                w->sent_a = w->target_name;                               /* _bind("G -> '" + sel_goal.name + "'") */
                for (int k = 0; k < p.ng; ++k)
                    if (w->ents[p.goals[k]].name_id == w->target_name) w->target_ent[p.goals[k]] = 1;
            } else {
                int refs[MAXENT], nr = 0;
                for (int k = 0; k < p.ng; ++k)
                    if (w->ents[p.goals[k]].name_id != w->ents[sel].name_id) refs[nr++] = p.goals[k];
                if (nr > 0) {                                            /* else: assert referents, "Identical object names?" */
                    int referent = refs[orc_xw_draw_below(w, nr)];
                    w->sent_a = w->ents[referent].name_id;                   /* _bind("G -> '" + referent.name + "'") */
                    for (int k = 0; k < p.ng; ++k)
                        if (w->ents[p.goals[k]].name_id != w->ents[referent].name_id) w->target_ent[p.goals[k]] = 1;
                }
            }
        }
    } else {
        idle_rearranging(w, &p, w->task_kind);
    }
    w->teacher_reward += 0.0;
    w->stage = ORC_STAGE_NAV;
}
