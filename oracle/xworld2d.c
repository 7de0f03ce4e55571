/*
 * oracle/xworld2d.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates, one environment at a time and in the reference's own data shapes
 * (entity list + a cube of item stacks per cell + a 64 px/cell canvas):
 *
 *   dynamics   games/xworld/xworld/xitem.cpp:80-155 (XAgent::act, full-observation
 *              4-action set), xmap.cpp:51-101 (add/remove/move_item),
 *              xworld.cpp:109-166 (reset/act), xworld_simulator.cpp:124-265
 *   teacher    teacher.cpp:202-251, teaching_task.cpp:64-116,176-222 (ordering),
 *              games/xworld3d/tasks/XWorld3DNavTarget.py:28-60 and
 *              xworld3d_task.py:98-124,170-180,328-342,451-482 (reward / done rule)
 *   maps       games/xworld/maps/xworld_env.py:95-101,118-150,152-225,412-493,
 *              XWorldNav.py:16-67, XWorldWalls.py:14-36, python/maze2d.py:43-114
 *   render     xmap.cpp:125-146,201-205 (to_image, full observation),
 *              xitem.cpp:33-63 (identity warp for yaw=1.5707963, scale=1, offset=0),
 *              xworld_simulator.cpp:278-307 (get_screen_rgb), :508-545 (down_sample_image)
 *              with OpenCV 3.2.0 resize(INTER_LINEAR, 8U) / cvtColor(BGR2GRAY) restated
 *              from the library's published algorithm (imgproc/src/imgwarp.cpp, color.cpp)
 *   caller     simulator_interface.cpp:95-143, simulator.cpp:36-117,152-161
 *
 * The reference draws its randomness from CPython-2 `random`, which it never
 * seeds and whose streams cannot be reproduced; map generation here follows
 * the reference's *algorithm* step by step but takes its decisions from the
 * build's Philox stream "xwb-rng-v1" in the order documented in DESIGN.md
 * ("xwb-mapgen-v1").  The reference-generated golden maps are replayed through
 * orc_xw_load_map().
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "xworld_internal.h"

/* ------------------------------------------------------------- helpers ---- */
static void cube_clear(orc_xworld *w) { memset(w->cube_n, 0, sizeof w->cube_n); }

/* XMap::add_item, xmap.cpp:51-63 */
static void map_add_item(orc_xworld *w, int e) {
    int x = w->ents[e].x, y = w->ents[e].y;
    for (int i = 0; i < w->cube_n[y][x]; ++i)
        if (w->cube[y][x][i] == e) return;
    if (w->cube_n[y][x] >= MAXSTACK) abort();
    w->cube[y][x][w->cube_n[y][x]++] = e;
}

/* XMap::remove_item, xmap.cpp:65-74 */
static void map_remove_item(orc_xworld *w, int e) {
    int x = w->ents[e].x, y = w->ents[e].y;
    for (int i = 0; i < w->cube_n[y][x]; ++i) {
        if (w->cube[y][x][i] == e) {
            for (int k = i; k + 1 < w->cube_n[y][x]; ++k) w->cube[y][x][k] = w->cube[y][x][k + 1];
            w->cube_n[y][x]--;
            break;
        }
    }
}

/* XMap::move_item, xmap.cpp:76-101.  XItem::is_reachable() is always false (xitem.h:137). */
static int map_move_item(orc_xworld *w, int item, int tx, int ty, int *contact, int *n_contact) {
    *n_contact = 0;
    if (tx < 0 || ty < 0 || tx >= w->width || ty >= w->height) return 0;
    int flag = 1;
    for (int i = 0; i < w->cube_n[ty][tx]; ++i) {
        int other = w->cube[ty][tx][i];
        int reachable = 0;
        if (!reachable && other != item) contact[(*n_contact)++] = other;
        flag &= reachable;
    }
    if (flag) {
        map_remove_item(w, item);
        w->ents[item].x = tx; w->ents[item].y = ty;
        map_add_item(w, item);
        return 1;
    }
    return 0;
}

/* XAgent::act, xitem.cpp:89-155.  FLAGS_visible_radius == 0: MOVE_UP / DOWN / LEFT / RIGHT; otherwise
 * MOVE_FORWARD, MOVE_BACKWARD, MOVE_LEFT_FPV, MOVE_RIGHT_FPV, TURN_LEFT, TURN_RIGHT relative to the facing direction;
 * a turn changes e_.yaw and returns the current cell (which XMap::move_item then refuses: the action "fails"). */
static void agent_act(orc_xworld *w, int action_id, int *tx, int *ty) {
    int cx = w->ents[w->agent_idx].x, cy = w->ents[w->agent_idx].y;
    if (w->cfg.visible_radius == 0) {
        if (action_id < 0 || action_id >= 4) abort();
        switch (action_id) {
            case 0: *tx = cx;     *ty = cy - 1; break;   /* MOVE_UP    */
            case 1: *tx = cx;     *ty = cy + 1; break;   /* MOVE_DOWN  */
            case 2: *tx = cx - 1; *ty = cy;     break;   /* MOVE_LEFT  */
            default:*tx = cx + 1; *ty = cy;     break;   /* MOVE_RIGHT */
        }
        return;
    }
    if (action_id < 0 || action_id >= 6) abort();
    const double PI = 3.14159265358979323846;                       /* M_PI */
    double *yaw = &w->e_yaw[w->agent_idx];
    int dir = orc_facing_dir(*yaw);                                 /* 0 right, 1 down, 2 left, 3 up */
    static const int fwd[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
    static const int left[4][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}};    /* MOVE_LEFT_FPV per facing dir */
    *tx = cx; *ty = cy;
    switch (action_id) {
        case 0: *tx = cx + fwd[dir][0];  *ty = cy + fwd[dir][1];  break;   /* MOVE_FORWARD   */
        case 1: *tx = cx - fwd[dir][0];  *ty = cy - fwd[dir][1];  break;   /* MOVE_BACKWARD  */
        case 2: *tx = cx + left[dir][0]; *ty = cy + left[dir][1]; break;   /* MOVE_LEFT_FPV  */
        case 3: *tx = cx - left[dir][0]; *ty = cy - left[dir][1]; break;   /* MOVE_RIGHT_FPV */
        case 4:                                                             /* TURN_LEFT      */
            *yaw -= PI / 2;
            if (*yaw < -PI / 2 - 1e-4) *yaw += 2 * PI;
            break;
        default:                                                            /* TURN_RIGHT     */
            *yaw += PI / 2;
            if (*yaw > PI + 1e-4) *yaw -= 2 * PI;
            break;
    }
}

/* maze2d.bfs, python/maze2d.py:43-71 (reachability only; the per-node shuffle of the
 * four moves does not change whether `end` is reached) */
int orc_bfs_reachable(int sx, int sy, int ex, int ey, int X, int Y, const uint8_t *obstacle) {
    int quex[MAXCELLS], quey[MAXCELLS], head = 0, tail = 0;
    uint8_t seen[MAXCELLS];
    memset(seen, 0, sizeof seen);
    quex[tail] = sx; quey[tail] = sy; tail++;
    seen[sy * X + sx] = 1;
    static const int mv[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
    while (head < tail) {
        int cx = quex[head], cy = quey[head]; head++;
        if (cx == ex && cy == ey) return 1;
        for (int m = 0; m < 4; ++m) {
            int nx = cx + mv[m][0], ny = cy + mv[m][1];
            if (nx >= 0 && nx < X && ny >= 0 && ny < Y && !seen[ny * X + nx] && !obstacle[ny * X + nx]) {
                seen[ny * X + nx] = 1;
                quex[tail] = nx; quey[tail] = ny; tail++;
            }
        }
    }
    return 0;
}

/* maze2d.spanning_tree_maze_generator, python/maze2d.py:74-114.
 * random.shuffle(moves) -> Fisher-Yates (i = 3..1, j = below(i+1)) on
 * [(-1,0),(1,0),(0,1),(0,-1)]; recursion unrolled with an explicit stack. */
void orc_maze_generate(orc_stream *s, int Xin, char *maze) {
    int X = Xin, pad = 0;
    if (X % 2 == 0) { pad = 1; X = X - 1; }
    int n = (X + 1) / 2;                       /* node lattice n x n */
    for (int y = 0; y < X; ++y)
        for (int x = 0; x < X; ++x)
            maze[y * Xin + x] = (x % 2 == 0 && y % 2 == 0) ? ' ' : '#';
    uint8_t visited[MAXCELLS];
    memset(visited, 0, sizeof visited);
    struct frame { int x, y; int mv[4]; int next; } stack[MAXCELLS];
    int sp = 0;
    static const int moves[4][2] = {{-1, 0}, {1, 0}, {0, 1}, {0, -1}};
    /* dfs((0,0)) */
    stack[0].x = 0; stack[0].y = 0; stack[0].next = -1; sp = 1;
    while (sp > 0) {
        struct frame *f = &stack[sp - 1];
        if (f->next < 0) {                      /* function entry */
            visited[f->y * n + f->x] = 1;
            for (int i = 0; i < 4; ++i) f->mv[i] = i;
            for (int i = 3; i >= 1; --i) {
                int j = (int)orc_stream_below(s, (uint32_t)(i + 1));
                int t = f->mv[i]; f->mv[i] = f->mv[j]; f->mv[j] = t;
            }
            f->next = 0;
        }
        if (f->next >= 4) { sp--; continue; }
        int m = f->mv[f->next++];
        int nx = f->x + moves[m][0], ny = f->y + moves[m][1];
        if (nx >= 0 && nx < n && ny >= 0 && ny < n && !visited[ny * n + nx]) {
            /* edges.add((cur, next)); rendered as maze[mid_y][mid_x] = ' ' */
            maze[(f->y + ny) * Xin + (f->x + nx)] = ' ';
            stack[sp].x = nx; stack[sp].y = ny; stack[sp].next = -1; sp++;
        }
    }
    if (pad) {
        /* maze.append([' ' if i % 2 == 0 else '#' for i in range(X)]);
           for i, m in enumerate(maze): m.append(' ' if i % 2 == 0 else '#') */
        for (int i = 0; i < X; ++i) maze[X * Xin + i] = (i % 2 == 0) ? ' ' : '#';
        for (int i = 0; i < Xin; ++i) maze[i * Xin + X] = (i % 2 == 0) ? ' ' : '#';
    }
}

/* ------------------------------------------------------ name tables ------ */
static void build_name_tables(orc_xworld *w) {
    for (int t = 0; t < 3; ++t) {
        int nn = 0;
        for (int i = 0; i < w->n_icons; ++i)
            if (w->info[i].type == t && w->info[i].name_id + 1 > nn) nn = w->info[i].name_id + 1;
        w->n_names[t] = nn;
        w->name_first[t] = (int *)calloc((size_t)nn + 1, sizeof(int));
        w->name_variants[t] = (int *)calloc((size_t)w->n_icons + 1, sizeof(int));
        int pos = 0;
        for (int nm = 0; nm < nn; ++nm) {
            w->name_first[t][nm] = pos;
            for (int i = 0; i < w->n_icons; ++i)          /* icon order == sorted path order */
                if (w->info[i].type == t && w->info[i].name_id == nm) w->name_variants[t][pos++] = i;
        }
        w->name_first[t][nn] = pos;
    }
}

static int n_variants(const orc_xworld *w, int type, int name) {
    return w->name_first[type][name + 1] - w->name_first[type][name];
}

static int variant_icon(const orc_xworld *w, int type, int name, int k) {
    return w->name_variants[type][w->name_first[type][name] + k];
}

/* ------------------------------------------------------ map generation --- */
typedef struct { int x, y; } cell;

static int cell_list_remove_at(cell *list, int n, int k) {
    for (int i = k; i + 1 < n; ++i) list[i] = list[i + 1];
    return n - 1;
}

static int cell_list_find(const cell *list, int n, int x, int y) {
    for (int i = 0; i < n; ++i) if (list[i].x == x && list[i].y == y) return i;
    return -1;
}

static void add_entity(orc_xworld *w, int type, int x, int y, int name, int icon, int serial) {
    orc_entity *e = &w->ents[w->n_ents++];
    e->type = type; e->x = x; e->y = y; e->name_id = name; e->icon = icon; e->serial = serial;
    int k = w->n_ents - 1;
    w->e_yaw[k] = 1.5707963; w->e_scale[k] = 1.0; w->e_offset[k] = 0.0;      /* Entity.__init__ defaults */
}

/* xworld_env.py:464-493 __padding_walls + :376-384 cpp_get_entities (offset shift) */
static void finish_map(orc_xworld *w) {
    int H = w->cfg.max_dim, W = w->cfg.max_dim;
    int h = w->actual_h, wd = w->actual_w, oh = w->offset_h, ow = w->offset_w;
    for (int i = 0; i < w->n_ents; ++i) { w->ents[i].x += ow; w->ents[i].y += oh; }
    int brick = variant_icon(w, 1, 0, 0);     /* self.items["block"]["brick"][0] */
    int id = H * W;
    /* add_blocks(range1 (x), range2 (y)) in itertools.product order: x outer, y inner */
    for (int x = 0; x < ow; ++x) for (int y = 0; y < h + oh; ++y) add_entity(w, 1, x, y, 0, brick, id++);
    for (int x = ow; x < W; ++x) for (int y = 0; y < oh; ++y) add_entity(w, 1, x, y, 0, brick, id++);
    for (int x = ow + wd; x < W; ++x) for (int y = oh; y < H; ++y) add_entity(w, 1, x, y, 0, brick, id++);
    for (int x = 0; x < ow + wd; ++x) for (int y = oh + h; y < H; ++y) add_entity(w, 1, x, y, 0, brick, id++);
    /* XWorld::reset, xworld.cpp:137-146: rebuild item list and the map */
    w->height = H; w->width = W;
    cube_clear(w);
    w->agent_idx = -1;
    for (int i = 0; i < w->n_ents; ++i) {
        if (w->ents[i].type == 2 && w->agent_idx < 0) w->agent_idx = i;
        map_add_item(w, i);
    }
}

static void set_dims(orc_xworld *w, int h, int wd) {
    /* xworld_env.py:118-134 set_dims */
    w->actual_h = h; w->actual_w = wd;
    w->offset_h = (w->cfg.max_dim - h) / 2;
    w->offset_w = (w->cfg.max_dim - wd) / 2;
}

/* XWorldNav._configure (XWorldNav.py:16-67) + XWorldEnv.__instantiate_entities
 * (xworld_env.py:412-452, maze_generation=True) with decisions from w->rs.
 * Draw order ("xwb-mapgen-v1", NAV):
 *   1. num_goals distinct goal names: for i: j = below(M-i); take names[j]; names[j] = names[M-1-i]
 *   2. maze DFS shuffles (orc_maze_generate)
 *   3. Fisher-Yates shuffle of the '#' cells listed row-major: i = n-1..1, j = below(i+1)
 *   4. entities in order goals, blocks, agent:
 *        goal : loc = avail[below(n_avail)] (order-preserving remove); icon variant below(nv)
 *        block: loc = blocks.pop(); name below(#block names); variant below(nv)
 *        agent: loc = avail[below(n_avail)]; name below(#agent names); variant below(nv)
 *      (below(n) always consumes one draw, also for n <= 1)                           */
/* XWorld(3D)Task.__record_result: success_seq.append(res), at most performance_window_size = 200 kept
 * (xworld3d_task.py:129-133, xworld_task.py:87-91); _record_env_usage hands the list to the env */
void orc_xw_record_result(orc_xworld *w, int kind, int result) {
    if (kind < 0 || kind >= 9) abort();
    /* _record_success / _record_failure (xworld3d_task.py:135-142, xworld_task.py:93-99): the counters
     * Task::obtain_performance reads; only the XWorld3D tasks add steps_in_cur_task */
    w->perf[kind][result ? 0 : 1] += 1;
    if (result && kind < ORC_TASK2D_TARGET) w->perf[kind][2] += w->steps_in_cur_task;
    if (w->use_len[kind] < 200) {
        w->use_bits[kind][(w->use_head[kind] + w->use_len[kind]) % 200] = (uint8_t)result;
        w->use_len[kind]++;
        w->use_sum[kind] += result;
    } else {
        w->use_sum[kind] += result - w->use_bits[kind][w->use_head[kind]];
        w->use_bits[kind][w->use_head[kind]] = (uint8_t)result;
        w->use_head[kind] = (w->use_head[kind] + 1) % 200;
    }
}

/* XWorldNav._configure, the curriculum != 0 branch (XWorldNav.py:27-55) with XWorldEnv.get_current_usage
 * (xworld_env.py:103-110): every 100th call that finds a recorded task compares the worst task's success rate
 * over its window with FLAGS_curriculum and moves to the next of the six levels */
int orc_xw_curriculum_configure(orc_xworld *w, int *dim, int *num_goals, int *num_blocks) {
    static const int goals_seq[6] = {2, 2, 2, 4, 4, 4}, blocks_seq[6] = {0, 3, 6, 9, 12, 16};
    if (w->cfg.max_dim != 8) abort();                 /* assert len(num_goals_seq) == n_levels */
    w->cur_counter += 1;
    int any = 0;
    for (int k = 0; k < 9; ++k) if (w->use_len[k] > 0) any = 1;
    double usage = 0;
    if (w->cur_counter >= 100 && any) {
        usage = 2;
        for (int k = 0; k < 9; ++k)
            if (w->use_len[k] > 0) {
                double u = (double)w->use_sum[k] / (double)w->use_len[k];
                if (u < usage) usage = u;
            }
        w->cur_counter = 0;
    }
    if (usage >= w->cfg.curriculum && w->cur_level < 5) w->cur_level += 1;
    *dim = 3 + w->cur_level; *num_goals = goals_seq[w->cur_level]; *num_blocks = blocks_seq[w->cur_level];
    return w->cur_level;
}

void orc_xw_curriculum_state(const orc_xworld *w, int *level, int *counter) { *level = w->cur_level; *counter = w->cur_counter; }

static void gen_map_nav(orc_xworld *w) {
    int D = w->cfg.dim, num_goals = w->cfg.num_goals, num_blocks = w->cfg.num_blocks;
    if (w->cfg.curriculum != 0) orc_xw_curriculum_configure(w, &D, &num_goals, &num_blocks);
    set_dims(w, D, D);
    w->n_ents = 0;
    w->running_id = 0;
    int M = w->n_names[0];
    int names[1024];
    if (M > 1024 || num_goals > M) abort();
    for (int i = 0; i < M; ++i) names[i] = i;
    int goal_name[64];
    for (int i = 0; i < num_goals; ++i) {
        int j = (int)orc_stream_below(&w->rs, (uint32_t)(M - i));
        goal_name[i] = names[j];
        names[j] = names[M - 1 - i];
    }
    char maze[MAXCELLS];
    orc_maze_generate(&w->rs, D, maze);
    cell blocks[MAXCELLS]; int nb = 0;
    cell avail[MAXCELLS]; int na = 0;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            if (maze[i * D + j] == '#') { blocks[nb].x = j; blocks[nb].y = i; nb++; }
            else { avail[na].x = j; avail[na].y = i; na++; }
        }
    for (int i = nb - 1; i >= 1; --i) {
        int j = (int)orc_stream_below(&w->rs, (uint32_t)(i + 1));
        cell t = blocks[i]; blocks[i] = blocks[j]; blocks[j] = t;
    }
    if (num_blocks > nb) abort();    /* assert blocks, "too many blocks for a valid maze" */
    for (int i = 0; i < num_goals; ++i) {
        int k = (int)orc_stream_below(&w->rs, (uint32_t)na);
        cell c = avail[k]; na = cell_list_remove_at(avail, na, k);
        int v = (int)orc_stream_below(&w->rs, (uint32_t)n_variants(w, 0, goal_name[i]));
        add_entity(w, 0, c.x, c.y, goal_name[i], variant_icon(w, 0, goal_name[i], v), w->running_id++);
        if (w->cfg.visible_radius) {
            /* xworld_env.py:211-223: yaw ~ U[0, 4*PI_2), scale ~ U[0.5, 1], offset ~ U[0, 1 - scale]; random.uniform(a, b)
             * = a + (b - a) * random(); here random() = the stream's unit() ("xwb-mapgen-v1", egocentric extension) */
            int k = w->n_ents - 1;
            double u0 = (double)orc_stream_unit(&w->rs), u1 = (double)orc_stream_unit(&w->rs), u2 = (double)orc_stream_unit(&w->rs);
            w->e_yaw[k] = 0 + (1.5707963 * 4 - 0) * u0;
            w->e_scale[k] = 0.5 + (1 - 0.5) * u1;
            w->e_offset[k] = 0 + ((1 - w->e_scale[k]) - 0) * u2;
        }
    }
    for (int i = 0; i < num_blocks; ++i) {
        cell c = blocks[--nb];
        int nm = (int)orc_stream_below(&w->rs, (uint32_t)w->n_names[1]);
        int v = (int)orc_stream_below(&w->rs, (uint32_t)n_variants(w, 1, nm));
        add_entity(w, 1, c.x, c.y, nm, variant_icon(w, 1, nm, v), w->running_id++);
    }
    {
        int k = (int)orc_stream_below(&w->rs, (uint32_t)na);
        cell c = avail[k]; na = cell_list_remove_at(avail, na, k);
        int nm = (int)orc_stream_below(&w->rs, (uint32_t)w->n_names[2]);
        int v = (int)orc_stream_below(&w->rs, (uint32_t)n_variants(w, 2, nm));
        add_entity(w, 2, c.x, c.y, nm, variant_icon(w, 2, nm, v), w->running_id++);
        if (w->cfg.visible_radius)                      /* xworld_env.py:208-210: random.choice(range(-1, 3)) * PI_2 */
            w->e_yaw[w->n_ents - 1] = (double)(-1 + (int)orc_stream_below(&w->rs, 4)) * 1.5707963;
    }
    finish_map(w);
}

/* XWorldWalls._configure (XWorldWalls.py:14-36) + __instantiate_entities (maze off).
 * Draw order ("xwb-mapgen-v1", WALLS): row = below(h); column = below(w); then entities in
 * order agent, goals, blocks: loc (if unset) = avail[below(n_avail)], name below(#names of type),
 * variant below(nv).  avail is the row-major cell list minus the wall cells.          */
static void gen_map_walls(orc_xworld *w) {
    /* without maze generation set_property() keeps the Entity default yaw 1.5707963, which check_or_get_value
     * rejects for the agent when visible_radius > 0 (xworld_env.py:210, py_util.py:27-29): the reference asserts */
    if (w->cfg.visible_radius) abort();
    int D = w->cfg.dim;
    set_dims(w, D, D);
    w->n_ents = 0;
    w->running_id = 0;
    cell avail[MAXCELLS]; int na = 0;
    for (int y = 0; y < D; ++y) for (int x = 0; x < D; ++x) { avail[na].x = x; avail[na].y = y; na++; }
    struct { int type, x, y, has_loc; } pend[MAXENT]; int np = 0;
    pend[np].type = 2; pend[np].has_loc = 0; np++;
    for (int i = 0; i < w->cfg.num_goals; ++i) { pend[np].type = 0; pend[np].has_loc = 0; np++; }
    int n_blocks = w->cfg.num_blocks;
    int row = (int)orc_stream_below(&w->rs, (uint32_t)D);
    int first = n_blocks < D ? n_blocks : D;
    for (int i = 0; i < first; ++i) {
        int k = cell_list_find(avail, na, i, row);
        if (k < 0) abort();
        na = cell_list_remove_at(avail, na, k);
        pend[np].type = 1; pend[np].x = i; pend[np].y = row; pend[np].has_loc = 1; np++;
    }
    n_blocks -= first;
    int column = (int)orc_stream_below(&w->rs, (uint32_t)D);
    int lim = n_blocks < D - 1 ? n_blocks : D - 1;
    for (int i = 0, j = 0; j < lim; ++i) {
        if (i != row) {
            int k = cell_list_find(avail, na, column, i);
            if (k < 0) abort();
            na = cell_list_remove_at(avail, na, k);
            pend[np].type = 1; pend[np].x = column; pend[np].y = i; pend[np].has_loc = 1; np++;
            j++;
        }
    }
    for (int i = 0; i < np; ++i) {
        int x = pend[i].x, y = pend[i].y;
        if (!pend[i].has_loc) {
            int k = (int)orc_stream_below(&w->rs, (uint32_t)na);
            x = avail[k].x; y = avail[k].y;
            na = cell_list_remove_at(avail, na, k);
        }
        int t = pend[i].type;
        int nm = (int)orc_stream_below(&w->rs, (uint32_t)w->n_names[t]);
        int v = (int)orc_stream_below(&w->rs, (uint32_t)n_variants(w, t, nm));
        add_entity(w, t, x, y, nm, variant_icon(w, t, nm, v), w->running_id++);
    }
    finish_map(w);
}

/* --------------------------------------------------------------- teacher -- */
/* XWorld3DTask._reachable, xworld3d_task.py:328-342 (coordinates in actual dims) */
int orc_task_reachable(const orc_xworld *w, int goal_ent) { return orc_task_reachable_ex(w, goal_ent, 1); }

/* goals_are_obstacles = 0: XWorldTask._reachable, games/xworld/tasks/xworld_task.py:347-357 (blocks only) */
int orc_task_reachable_ex(const orc_xworld *w, int goal_ent, int goals_are_obstacles) {
    uint8_t obst[MAXCELLS];
    int X = w->actual_w, Y = w->actual_h;
    memset(obst, 0, sizeof obst);
    const orc_entity *a = &w->ents[w->agent_idx];
    const orc_entity *g = &w->ents[goal_ent];
    int ax = a->x - w->offset_w, ay = a->y - w->offset_h;
    int gx = g->x - w->offset_w, gy = g->y - w->offset_h;
    if (ax == gx && ay == gy) return 1;
    for (int i = 0; i < w->n_ents; ++i) {
        const orc_entity *e = &w->ents[i];
        int ex = e->x - w->offset_w, ey = e->y - w->offset_h;
        if (ex < 0 || ey < 0 || ex >= X || ey >= Y) continue;   /* padding blocks are dropped, xworld_env.py:393 */
        if (e->type == 1) obst[ey * X + ex] = 1;
        if (goals_are_obstacles && e->type == 0 && !(ex == gx && ey == gy)) obst[ey * X + ex] = 1;
    }
    return orc_bfs_reachable(ax, ay, gx, gy, X, Y, obst);
}

/* XWorld3DTask._reach_object, xworld3d_task.py:451-454 with
 * _get_direction_and_distance (:98-124), evaluated in double exactly as Python does */
static int task_reach_object(const orc_xworld *w, int goal_ent) {
    int in_hits = 0;
    for (int i = 0; i < w->n_hits; ++i) if (w->hits[i] == goal_ent) in_hits = 1;
    const orc_entity *a = &w->ents[w->agent_idx];
    const orc_entity *g = &w->ents[goal_ent];
    double yaw = w->e_yaw[w->agent_idx];          /* xworld_env.py:42 Entity default 1.5707963 under full observation */
    double dx = g->x - a->x, dy = g->y - a->y;
    double dist = sqrt(dx * dx + dy * dy);
    double theta;
    if (dist == 0) {
        theta = 0;
    } else {
        double v1x = cos(yaw), v1y = sin(yaw);
        double v2x = dx / dist, v2y = dy / dist;
        double cos_theta = fmax(-1, fmin(1, v1x * v2x + v1y * v2y));
        double sin_theta = fmax(-1, fmin(1, v1y * v2x - v1x * v2y));
        theta = acos(cos_theta) * copysign(1, asin(sin_theta));
    }
    const double PI_py = 3.1415926;               /* xworld3d_task.py:36 */
    return fabs(theta) < PI_py / 4 && in_hits;
}

/* XWorld3DNavTarget.navigation_reward (:45-60) + _time_reward (xworld3d_task.py:472-482) */
static void task_navigation_reward(orc_xworld *w) {
    double reward = -0.01;                        /* time_penalty */
    int time_out = 0;
    w->steps_in_cur_task += 1;
    if (w->steps_in_cur_task >= w->actual_h * w->actual_w * w->cfg.max_steps_factor) {
        w->event = ORC_EV_TIMEUP;
        orc_xw_record_result(w, w->task_kind, 0);   /* _time_reward: _record_failure */
        w->perf[w->task_kind][3] += 1;
        time_out = 1;
    }
    int next_stage = ORC_STAGE_NAV;
    if (!time_out) {
        int any_reach = 0, target_reach = 0;
        for (int i = 0; i < w->n_ents; ++i) {
            if (w->ents[i].type != 0) continue;
            if (task_reach_object(w, i)) {
                any_reach = 1;
                if (orc_task_is_target(w, i)) target_reach = 1;      /* t.id in objects_reach_test */
            }
        }
        if (w->task_kind == ORC_TASK_BETWEEN) {
            /* XWorld3DNavTargetBetween.navigation_reward (XWorld3DNavTargetBetween.py:64-88): any reached goal
             * fails; otherwise success when the agent stands within threshold/2 = 0.5 of the middle cell */
            const orc_entity *a = &w->ents[w->agent_idx];
            double ddx = a->x - w->between_x, ddy = a->y - w->between_y;
            target_reach = !any_reach && sqrt(ddx * ddx + ddy * ddy) < 1.0 / 2;
        }
        if (target_reach) {
            w->event = ORC_EV_CORRECT;
            orc_xw_record_result(w, w->task_kind, 1);                /* _successful_goal: _record_success */
            reward += 1.0;                        /* correct_reward */
            next_stage = ORC_STAGE_TERMINAL;
        } else if (any_reach) {
            w->event = ORC_EV_WRONG;
            orc_xw_record_result(w, w->task_kind, 0);                /* _failed_goal: _record_failure */
            reward += -1.0;                       /* wrong_reward */
            next_stage = ORC_STAGE_TERMINAL;
        }
    } else {
        next_stage = ORC_STAGE_TERMINAL;
    }
    w->teacher_reward += reward;                  /* Task::give_reward -> add_teacher_reward */
    w->stage = next_stage;
}

static void group_save(orc_xworld *w, int g) {
    orc_group_state *s = &w->grp[g];
    s->stage = w->stage; s->steps_in_cur_task = w->steps_in_cur_task; s->target_name = w->target_name; s->task_kind = w->task_kind;
    memcpy(s->target_ent, w->target_ent, sizeof s->target_ent);
    s->between_x = w->between_x; s->between_y = w->between_y; s->sent_a = w->sent_a; s->sent_b = w->sent_b;
    s->dir_ref_ent = w->dir_ref_ent; s->dir_word = w->dir_word; s->target2d_x = w->target2d_x; s->target2d_y = w->target2d_y;
    s->last_event = w->event;
}
static void group_load(orc_xworld *w, int g) {
    const orc_group_state *s = &w->grp[g];
    w->stage = s->stage; w->steps_in_cur_task = s->steps_in_cur_task; w->target_name = s->target_name; w->task_kind = s->task_kind;
    memcpy(w->target_ent, s->target_ent, sizeof s->target_ent);
    w->between_x = s->between_x; w->between_y = s->between_y; w->sent_a = s->sent_a; w->sent_b = s->sent_b;
    w->dir_ref_ent = s->dir_ref_ent; w->dir_word = s->dir_word; w->target2d_x = s->target2d_x; w->target2d_y = s->target2d_y;
    w->act_n_tasks = g ? w->cfg.n_tasks2 : w->cfg.n_tasks;
    w->act_tasks = g ? w->cfg.tasks2 : w->cfg.tasks;
    w->act_schedule = g ? w->cfg.task_schedule2 : w->cfg.task_schedule;
    w->act_weights = g ? w->cfg.task_weights2 : w->cfg.task_weights;
}

static void teacher_run_group(orc_xworld *w, int idle_pick);

static int teacher_exclusive(const orc_xworld *w) {
    return w->cfg.task_groups_exclusive && w->cfg.task_mode != ORC_TASKMODE_LANG_ACQ;      /* simulator_interface.cpp:46-48 */
}

/* Teacher::nondeterministic_sort_task_groups, teacher.cpp:143-163: position i takes one of the remaining groups with
 * probability proportional to its weight -- util::simple_importance_sampling over the accumulated remaining weights, also
 * for the last position (one weight: index 0, the draw is still made).  Decisions ("xwb-taskgen-v1"): stream 4, block =
 * num_steps, one unit() per position; under cfg.simulator_seed the env's own minstd engine, as in the reference. */
static void teacher_sort_groups(orc_xworld *w) {
    orc_stream gs;
    orc_stream_init(&gs, w->cfg.seed, w->env_gid, w->episode, 4);
    gs.ctr[0] = (uint32_t)w->num_steps;
    for (int i = 0; i < w->n_groups; ++i) {
        double acc[2], total = 0;
        for (int j = i; j < w->n_groups; ++j) { total += w->cfg.group_weight[w->grp_order[j]]; acc[j - i] = total; }
        int idx;
        if (w->forced) {
            idx = orc_xw_draw_below(w, w->n_groups - i);
        } else {
            float val = orc_stream_unit(&gs) * (float)total;             /* get_rand_range_val(float(acc.back())) */
            if (w->cfg.simulator_seed) val = orc_minstd_rand_range(&w->reng, (float)total);
            idx = w->n_groups - i - 1;
            for (int j = 0; j < w->n_groups - i; ++j) if ((double)val <= acc[j]) { idx = j; break; }
        }
        int t = w->grp_order[i]; w->grp_order[i] = w->grp_order[i + idx]; w->grp_order[i + idx] = t;
    }
}

/* Teacher::teach, teacher.cpp:207-230.  task_groups_exclusive_ == false: every group's stage in conf order.  Rewards add
 * up in the buffer (add_teacher_reward); Task::py_stage ends with record_event_in_buffer(task.get_event()), so the buffer
 * holds the LAST group's event, "" included; game_events_ is cleared by the first py_stage that reads it.
 * task_groups_exclusive_ == true: the groups are re-sorted, then one group runs -- the last one of the sorted list that is
 * not idle (the reference's loop has no break), else the first. */
static void teacher_teach(orc_xworld *w, int idle_pick) {
    /* before_teach: clear_teacher_env_buffer */
    w->teacher_reward = 0; w->event = ORC_EV_NONE;
    if (teacher_exclusive(w)) {
        teacher_sort_groups(w);
        if (w->n_groups > 1) {
            int pick = -1;
            for (int k = 0; k < w->n_groups; ++k)
                if (w->grp[w->grp_order[k]].stage != ORC_STAGE_IDLE) pick = w->grp_order[k];   /* TaskGroup::is_idle */
            if (pick < 0) pick = w->grp_order[0];
            group_load(w, pick);
            w->event = ORC_EV_NONE;
            teacher_run_group(w, idle_pick);
            const int ev = w->event;
            group_save(w, pick);
            group_load(w, 0);                     /* accessors read group 0 */
            w->event = ev;
            return;
        }
    }
    if (w->n_groups <= 1) {
        group_load(w, 0);                         /* (the working fields already are group 0's: sets the task list) */
        teacher_run_group(w, idle_pick);
        group_save(w, 0);
        return;
    }
    int last_event = ORC_EV_NONE;
    for (int g = 0; g < w->n_groups; ++g) {
        if (g > 0) group_save(w, g - 1);
        if (g > 0) group_load(w, g);
        else group_load(w, 0);
        w->event = ORC_EV_NONE;
        teacher_run_group(w, idle_pick);
        last_event = w->event;                    /* record_event_in_buffer: overwrites */
    }
    group_save(w, w->n_groups - 1);
    group_load(w, 0);                             /* accessors read group 0 */
    w->event = last_event;
}

static void teacher_run_group(orc_xworld *w, int idle_pick) {
    switch (w->stage) {
        case ORC_STAGE_IDLE:
            (void)idle_pick;
            /* an idle stage at step time: the 2-D-native tasks come back to "idle" (decisions: the words of stream 2,
             * block = num_steps); under exclusive scheduling an XWorld3DNav* group can be picked idle in mid-episode: its
             * (many) decisions are the successive words of a stream of that step's own, id 5 | num_steps << 8
             * ("xwb-taskgen-v1") */
            if (w->num_steps > 0) {
                if (w->act_n_tasks > 0 && w->act_tasks[0] >= ORC_TASK2D_TARGET) {
                    orc_stream_init(&w->rs, w->cfg.seed, w->env_gid, w->episode, 2);
                    w->rs.ctr[0] = (uint32_t)w->num_steps;
                } else {
                    orc_stream_init(&w->rs, w->cfg.seed, w->env_gid, w->episode, 5u | ((uint32_t)w->num_steps << 8));
                }
            }
            orc_task_idle(w);
            break;
        case ORC_STAGE_NAV:
            if (w->task_kind >= ORC_TASK2D_TARGET) orc_task2d_navigation_reward(w);
            else task_navigation_reward(w);
            break;
        default: w->teacher_reward += 0; break;   /* terminal(): ["terminal", 0, ""] */
    }
    /* py_stage consumed game_events_ (get_events_of_game) */
    w->n_hits = 0;
}

/* ---------------------------------------------------------------- render -- */
/* OpenCV 3.2.0 imgproc resize(), INTER_LINEAR, CV_8U, fixed point (INTER_RESIZE_COEF_BITS = 11) */
static int cv_round_f(float v) { return (int)lrintf(v); }     /* cvRound: round half to even */
static short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

void orc_cv_resize_linear_8u(const uint8_t *src, int sh, int sw, int cn, uint8_t *dst, int dh, int dw) {
    if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * cn); return; }
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw * cn);
    short *ialpha = (short *)malloc(sizeof(short) * (size_t)dw * cn * 2);
    int *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    short *ibeta = (short *)malloc(sizeof(short) * (size_t)dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        float cbuf0 = 1.f - fx, cbuf1 = fx;
        for (int k = 0; k < cn; ++k) {
            xofs[dx * cn + k] = sx * cn + k;
            ialpha[(dx * cn + k) * 2 + 0] = sat_short(cv_round_f(cbuf0 * 2048));
            ialpha[(dx * cn + k) * 2 + 1] = sat_short(cv_round_f(cbuf1 * 2048));
        }
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2 + 0] = sat_short(cv_round_f((1.f - fy) * 2048));
        ibeta[dy * 2 + 1] = sat_short(cv_round_f(fy * 2048));
    }
    int width = dw * cn;
    int xmaxc = xmax * cn;
    int *row0 = (int *)malloc(sizeof(int) * (size_t)width);
    int *row1 = (int *)malloc(sizeof(int) * (size_t)width);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy];
        int *rows[2] = {row0, row1};
        for (int k = 0; k < 2; ++k) {
            int sy = sy0 + k;
            sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;            /* clip(sy, 0, ssize.height) */
            const uint8_t *S = src + (size_t)sy * sw * cn;
            int *D = rows[k];
            int dx = 0;
            for (; dx < xmaxc; ++dx) {                             /* HResizeLinear */
                int sx = xofs[dx];
                D[dx] = S[sx] * ialpha[dx * 2] + S[sx + cn] * ialpha[dx * 2 + 1];
            }
            for (; dx < width; ++dx) D[dx] = (int)S[xofs[dx]] * 2048;
        }
        short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *out = dst + (size_t)dy * width;
        for (int x = 0; x < width; ++x)                            /* VResizeLinear<uchar,...> */
            out[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(row0); free(row1); free(xofs); free(ialpha); free(yofs); free(ibeta);
}

/* OpenCV 3.2.0 cvtColor(COLOR_BGR2GRAY), CV_8U: RGB2Gray<uchar>, yuv_shift = 14 */
void orc_cv_bgr2gray_8u(const uint8_t *src, int n_pixels, uint8_t *dst) {
    for (int i = 0; i < n_pixels; ++i) {
        int b = src[i * 3], g = src[i * 3 + 1], r = src[i * 3 + 2];
        dst[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
}

/* XWorldSimulator::get_screen, xworld_simulator.cpp:278-285 */
void orc_xw_get_screen(const orc_xworld *w, uint8_t *out) {
    if (!w->icons64) abort();
    int H = w->height, W = w->width;
    int ih = H * ITEM_SIZE, iw = W * ITEM_SIZE;
    /* XMap::to_image, xmap.cpp:125-146: canvas filled with 255, items copied in stack order */
    uint8_t *world = (uint8_t *)malloc((size_t)ih * iw * 3);
    if (w->cfg.visible_radius > 0) {
        /* egocentric: the r*64-pixel view (xworld_ego.c), then get_screen_rgb's resize to img_height_ x img_width_ */
        int r = w->cfg.visible_radius, S = r * ITEM_SIZE;
        uint8_t *view = (uint8_t *)malloc((size_t)S * S * 3);
        orc_xw_ego_view(w, r, view);
        orc_cv_resize_linear_8u(view, S, S, 3, world, ih, iw);
        free(view);
    } else {
    memset(world, 255, (size_t)ih * iw * 3);
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j)
            for (int k = 0; k < w->cube_n[i][j]; ++k) {
                const uint8_t *icon = w->icons64 + (size_t)w->ents[w->cube[i][j][k]].icon * ITEM_SIZE * ITEM_SIZE * 3;
                for (int r = 0; r < ITEM_SIZE; ++r)
                    memcpy(world + ((size_t)(i * ITEM_SIZE + r) * iw + (size_t)j * ITEM_SIZE) * 3,
                           icon + (size_t)r * ITEM_SIZE * 3, ITEM_SIZE * 3);
            }
    }
    /* get_screen_rgb (:287-307): same-size resize, then interleaved BGR -> planar */
    uint8_t *rgbs = (uint8_t *)malloc((size_t)ih * iw * 3);
    for (int i = 0; i < ih; ++i)
        for (int j = 0; j < iw; ++j)
            for (int c = 0; c < 3; ++c)
                rgbs[(size_t)c * iw * ih + (size_t)i * iw + j] = world[((size_t)i * iw + j) * 3 + c];
    /* down_sample_image (:508-545): planar -> interleaved, resize, [gray], planar out */
    uint8_t *img = world;   /* reuse */
    for (int h = 0; h < ih; ++h)
        for (int x = 0; x < iw; ++x)
            for (int c = 0; c < 3; ++c)
                img[((size_t)h * iw + x) * 3 + c] = rgbs[(size_t)c * iw * ih + (size_t)h * iw + x];
    int oh = w->img_h_out, ow = w->img_w_out;
    uint8_t *img_out = (uint8_t *)malloc((size_t)oh * ow * 3);
    orc_cv_resize_linear_8u(img, ih, iw, 3, img_out, oh, ow);
    if (!w->cfg.color) {
        uint8_t *gray = (uint8_t *)malloc((size_t)oh * ow);
        orc_cv_bgr2gray_8u(img_out, oh * ow, gray);
        memcpy(out, gray, (size_t)oh * ow);
        free(gray);
    } else {
        for (int h = 0; h < oh; ++h)
            for (int x = 0; x < ow; ++x)
                for (int c = 0; c < 3; ++c)
                    out[(size_t)c * ow * oh + (size_t)h * ow + x] = img_out[((size_t)h * ow + x) * 3 + c];
    }
    free(img_out); free(rgbs); free(world);
}

static size_t screen_size(const orc_xworld *w) {
    return (size_t)w->img_h_out * w->img_w_out * w->channels;
}

static void make_context_screens(orc_xworld *w) {
    if (!w->icons64) return;
    size_t sz = screen_size(w);
    memmove(w->screens, w->screens + sz, sz * (size_t)(w->cfg.context - 1));
    orc_xw_get_screen(w, w->screens + sz * (size_t)(w->cfg.context - 1));
}

static void init_screen(orc_xworld *w) {
    if (!w->icons64) return;
    memset(w->screens, 0, screen_size(w) * (size_t)w->cfg.context);
    make_context_screens(w);
}

/* ---------------------------------------------------------------- public -- */
orc_xworld *orc_xw_create(const orc_xw_cfg *cfg, int n_icons, const orc_icon_info *info,
                          const uint8_t *icons64) {
    orc_xworld *w = (orc_xworld *)calloc(1, sizeof *w);
    w->cfg = *cfg;
    if (w->cfg.context < 1) w->cfg.context = 1;
    if (w->cfg.max_dim > MAXDIM || w->cfg.dim > w->cfg.max_dim) abort();
    w->n_icons = n_icons;
    w->info = (orc_icon_info *)malloc(sizeof(orc_icon_info) * (size_t)n_icons);
    memcpy(w->info, info, sizeof(orc_icon_info) * (size_t)n_icons);
    w->icons64 = icons64;
    build_name_tables(w);
    /* XWorldSimulator::init, xworld_simulator.cpp:48-77: full observation, block_size 12 */
    w->height = w->width = w->cfg.max_dim;
    w->img_h_out = w->height * 12;
    w->img_w_out = w->width * 12;
    if (w->cfg.visible_radius > 0) {                                /* :62-68 */
        if (w->cfg.visible_radius > w->cfg.max_dim) w->cfg.visible_radius = w->cfg.max_dim;
        int block_size = 84 / w->cfg.visible_radius;
        w->img_h_out = w->img_w_out = w->cfg.visible_radius * block_size;
    }
    w->channels = w->cfg.color ? 3 : 1;
    w->screens = (uint8_t *)calloc(screen_size(w) * (size_t)w->cfg.context, 1);
    w->last_action_success = 1;       /* GameSimulator ctor default, simulator.cpp:33-34 */
    w->cur_level = w->cfg.start_level;
    if (w->cfg.curriculum != 0) {
        if (w->cfg.map_kind != ORC_MAP_NAV) w->cfg.curriculum = 0;            /* XWorldWalls never reads the flag */
        else if (w->cur_level < 0 || w->cur_level > 5 || w->cfg.max_dim != 8) abort();
    }
    return w;
}

void orc_xw_destroy(orc_xworld *w) {
    if (!w) return;
    for (int t = 0; t < 3; ++t) { free(w->name_first[t]); free(w->name_variants[t]); }
    free(w->info); free(w->screens); free(w);
}

void orc_xw_rebuild_map(orc_xworld *w) {
    cube_clear(w);
    w->agent_idx = -1;
    for (int i = 0; i < w->n_ents; ++i) {
        if (w->ents[i].type == 2 && w->agent_idx < 0) w->agent_idx = i;
        map_add_item(w, i);
    }
}

int orc_xw_draw_below(orc_xworld *w, int n) {
    if (w->forced) {
        if (w->forced_at >= w->n_forced) abort();
        int v = w->forced[w->forced_at++];
        if (n > 0 && (v < 0 || v >= n)) abort();
        return v;
    }
    return (int)orc_stream_below(&w->rs, (uint32_t)n);
}

static void after_map(orc_xworld *w, int idle_pick) {
    /* XWorldSimulator::reset_game (:143-157), GameSimulator::reset_game */
    w->n_hits = 0;
    w->num_steps = 0;             /* last_action_success_ is NOT touched by reset_game */
    /* Teacher::reset_after_game_reset + teach(): lazy Task::reset then idle stage */
    w->n_groups = w->cfg.n_tasks2 > 0 ? 2 : 1;
    if (w->episode == 0) { w->grp_order[0] = 0; w->grp_order[1] = 1; }    /* a new env's teacher: conf order */
    for (int g = w->n_groups - 1; g >= 0; --g) {      /* TaskGroup::reset for every group; group 0's ends up in the working fields */
        w->stage = ORC_STAGE_IDLE;
        w->steps_in_cur_task = 0;
        w->target_name = -1;
        w->target2d_x = w->target2d_y = -1;
        w->task_kind = ORC_TASK_TARGET;
        memset(w->target_ent, 0, sizeof w->target_ent);
        w->between_x = w->between_y = -1; w->sent_a = w->sent_b = -1; w->dir_ref_ent = -1; w->dir_word = 0;
        w->event = ORC_EV_NONE;
        group_save(w, g);
    }
    teacher_teach(w, idle_pick);
    init_screen(w);
}

void orc_xw_reset_game(orc_xworld *w, uint32_t env_gid, uint32_t episode) {
    w->env_gid = env_gid; w->episode = episode;
    w->forced = NULL; w->n_forced = 0;
    /* one object may stand in for many envs in turn: an env's engine starts with its first episode */
    if (w->cfg.simulator_seed && episode == 0)
        orc_minstd_seed_thread(&w->reng, w->cfg.simulator_seed, w->cfg.thread_base + (int)env_gid + 1);
    orc_stream_init(&w->rs, w->cfg.seed, env_gid, episode, 0);
    if (w->cfg.map_kind == ORC_MAP_WALLS) gen_map_walls(w);
    else gen_map_nav(w);
    after_map(w, -1);
}

void orc_xw_load_map_ex(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                        const int *decisions, int n_decisions, uint32_t env_gid, uint32_t episode) {
    w->forced = decisions; w->n_forced = n_decisions; w->forced_at = 0;
    orc_xw_load_map(w, n_entities, ents, dim, -1, env_gid, episode);
    if (decisions && w->forced_at != n_decisions) abort();       /* every decision must have been consumed */
    w->forced = NULL; w->n_forced = 0;
}

void orc_xw_load_map_forced(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                            const int *decisions, int n_decisions, uint32_t env_gid, uint32_t episode) {
    w->forced = decisions; w->n_forced = n_decisions; w->forced_at = 0;
    orc_xw_load_map(w, n_entities, ents, dim, -1, env_gid, episode);
}

void orc_xw_set_pose(orc_xworld *w, int ent, double yaw, double scale, double offset) {
    if (ent < 0 || ent >= w->n_ents) abort();
    w->e_yaw[ent] = yaw; w->e_scale[ent] = scale; w->e_offset[ent] = offset;
}
void orc_xw_get_pose(const orc_xworld *w, int ent, double *yaw, double *scale, double *offset) {
    *yaw = w->e_yaw[ent]; *scale = w->e_scale[ent]; *offset = w->e_offset[ent];
}
double orc_xw_agent_yaw(const orc_xworld *w) { return w->e_yaw[w->agent_idx]; }
void orc_xw_agent_masking(const orc_xworld *w, int *x_st, int *y_st, uint8_t *shadow) {
    const orc_entity *a = &w->ents[w->agent_idx];
    orc_xw_image_masking(w, a->x, a->y, w->e_yaw[w->agent_idx], w->cfg.visible_radius, x_st, y_st, shadow);
}
void orc_xw_refresh_screen(orc_xworld *w) { init_screen(w); }
/* XMap::to_image(agent, false, r): the r*64-pixel egocentric view BEFORE the two resizes, interleaved B,G,R */
void orc_xw_agent_view(const orc_xworld *w, uint8_t *view) {
    if (w->cfg.visible_radius <= 0 || !w->icons64) abort();
    orc_xw_ego_view(w, w->cfg.visible_radius, view);
}
/* XItem::get_item_image of one entity (its pose applied), 64 x 64 interleaved B,G,R */
void orc_xw_entity_image(const orc_xworld *w, int ent, uint8_t *out) {
    if (ent < 0 || ent >= w->n_ents || !w->icons64) abort();
    orc_xw_item_image(w, ent, out);
}
void orc_xw_stage_poses(orc_xworld *w, const double *poses, int n_entities) { w->staged_poses = poses; w->n_staged_poses = n_entities; }

void orc_xw_sentence_names(const orc_xworld *w, int *a, int *b) { *a = w->sent_a; *b = w->sent_b; }

void orc_xw_direction_target(const orc_xworld *w, int *x, int *y, int *word) {
    *x = *y = -1; *word = 0;
    if (w->task_kind == ORC_TASK_DIRECTION && w->dir_ref_ent >= 0) { *x = w->ents[w->dir_ref_ent].x; *y = w->ents[w->dir_ref_ent].y; *word = w->dir_word; }
}

int orc_xw_forced_left(const orc_xworld *w) { return w->forced ? w->n_forced - w->forced_at : 0; }
void orc_xw_target2d(const orc_xworld *w, int *x, int *y) { *x = w->target2d_x; *y = w->target2d_y; }

void orc_xw_load_map(orc_xworld *w, int n_entities, const orc_entity *ents, int dim,
                     int target_pick, uint32_t env_gid, uint32_t episode) {
    /* target_pick >= 0: legacy form for a NavTarget-only config = decisions {task 0, pick} */
    int legacy[2] = {0, target_pick};
    if (target_pick >= 0 && !w->forced) {
        w->forced = legacy; w->n_forced = 2; w->forced_at = 0;
        orc_xw_load_map(w, n_entities, ents, dim, -1, env_gid, episode);
        w->forced = NULL; w->n_forced = 0;
        return;
    }
    w->env_gid = env_gid; w->episode = episode;
    orc_stream_init(&w->rs, w->cfg.seed, env_gid, episode, 0);
    set_dims(w, dim, dim);
    w->n_ents = 0;
    for (int i = 0; i < n_entities; ++i)
        add_entity(w, ents[i].type, ents[i].x, ents[i].y, ents[i].name_id, ents[i].icon, ents[i].serial);
    if (w->staged_poses) {
        if (w->n_staged_poses != n_entities) abort();
        for (int i = 0; i < n_entities; ++i) {
            w->e_yaw[i] = w->staged_poses[3 * i]; w->e_scale[i] = w->staged_poses[3 * i + 1]; w->e_offset[i] = w->staged_poses[3 * i + 2];
        }
        w->staged_poses = NULL;
    }
    finish_map(w);
    after_map(w, target_pick);
}

/* SimulatorInterface::take_actions, simulator_interface.cpp:126-137 */
float orc_xw_take_actions(orc_xworld *w, int action, int act_rep) {
    float r = 0;
    /* GameSimulator::take_actions, simulator.cpp:98-108 */
    float reward = 0;
    w->num_steps++;
    for (int i = 0; i < act_rep; ++i) {
        /* XWorldSimulator::take_action (:200-265): TeachingEnvironment::take_action clears the
         * teacher buffer; move; record collision events; returns 0 */
        w->teacher_reward = 0; w->event = ORC_EV_NONE;
        int tx, ty, contact[MAXSTACK], nc;
        agent_act(w, action, &tx, &ty);
        w->last_action_success = map_move_item(w, w->agent_idx, tx, ty, contact, &nc);
        for (int k = 0; k < nc; ++k) w->hits[w->n_hits++] = contact[k];
        reward += 0;
    }
    r += reward;
    teacher_teach(w, -1);
    r = (float)((double)r + w->teacher_reward);    /* r += teacher_->give_reward() (double) */
    make_context_screens(w);
    return r;
}

/* AgentSpecificSimulator::game_over (simulator.cpp:158-161) | XWorldSimulator::game_over (:165-198) */
int orc_xw_game_over(const orc_xworld *w) {
    int base = (w->cfg.max_steps > 0 && w->num_steps >= w->cfg.max_steps) ? ORC_MAX_STEP : ORC_ALIVE;
    int code = ORC_ALIVE;
    if (w->cfg.task_mode == ORC_TASKMODE_LANG_ACQ) {
        if (w->event == ORC_EV_CORRECT) code = ORC_SUCCESS;        /* event.find("correct") */
        else if (w->event == ORC_EV_WRONG) code = ORC_DEAD;        /* event.find("wrong")   */
        else if (w->event == ORC_EV_TIMEUP) code = ORC_MAX_STEP;   /* event == "time_up"    */
    }
    return base | code;
}

int orc_xw_get_lives(const orc_xworld *w) { return orc_xw_game_over(w) ? 0 : 1; }   /* :506 */
int orc_xw_num_actions(const orc_xworld *w) { return w->cfg.visible_radius ? 6 : 4; }   /* xitem.cpp:80-87 */
int64_t orc_xw_num_steps(const orc_xworld *w) { return w->num_steps; }
int orc_xw_last_action_success(const orc_xworld *w) { return w->last_action_success; }
int orc_xw_event(const orc_xworld *w) { return w->event; }
void orc_xw_group_state(const orc_xworld *w, int g, int *kind, int *stage, int *steps_in_task, int *event,
                        int *target2d_x, int *target2d_y) {
    const orc_group_state *s = &w->grp[g < 0 || g >= w->n_groups ? 0 : g];
    *kind = s->task_kind; *stage = s->stage; *steps_in_task = s->steps_in_cur_task; *event = s->last_event;
    *target2d_x = s->target2d_x; *target2d_y = s->target2d_y;
}
void orc_xw_get_performance(const orc_xworld *w, int64_t out[9][4]) { memcpy(out, w->perf, sizeof w->perf); }
int orc_xw_group_first(const orc_xworld *w) { return w->grp_order[0]; }
int orc_xw_stage(const orc_xworld *w) { return w->stage; }
int orc_xw_target_name(const orc_xworld *w) { return w->target_name; }
int orc_xw_task_kind(const orc_xworld *w) { return w->task_kind; }
void orc_xw_between_cell(const orc_xworld *w, int *x, int *y) { *x = w->between_x; *y = w->between_y; }
/* 1 where the cell's top item is a target goal (self.target), max_dim*max_dim, row-major [y][x] */
void orc_xw_get_target_cells(const orc_xworld *w, uint8_t *out) {
    for (int y = 0; y < w->height; ++y)
        for (int x = 0; x < w->width; ++x) {
            int n = w->cube_n[y][x];
            out[y * w->width + x] = (uint8_t)(n ? orc_task_is_target(w, w->cube[y][x][n - 1]) : 0);
        }
}
int orc_xw_steps_in_task(const orc_xworld *w) { return w->steps_in_cur_task; }
int orc_xw_n_entities(const orc_xworld *w) { return w->n_ents; }
void orc_xw_get_entities(const orc_xworld *w, orc_entity *out) {
    memcpy(out, w->ents, sizeof(orc_entity) * (size_t)w->n_ents);
}
void orc_xw_agent_xy(const orc_xworld *w, int *x, int *y) {
    *x = w->ents[w->agent_idx].x; *y = w->ents[w->agent_idx].y;
}
void orc_xw_get_grid(const orc_xworld *w, int32_t *out) {
    for (int y = 0; y < w->height; ++y)
        for (int x = 0; x < w->width; ++x) {
            int n = w->cube_n[y][x];
            out[y * w->width + x] = n ? w->ents[w->cube[y][x][n - 1]].icon + 1 : 0;
        }
}
void orc_xw_screen_dims(const orc_xworld *w, int *h, int *wd, int *c) {
    *h = w->img_h_out; *wd = w->img_w_out; *c = w->channels;
}
void orc_xw_get_state_screen(const orc_xworld *w, uint8_t *out) {
    memcpy(out, w->screens, screen_size(w) * (size_t)w->cfg.context);
}

/* ---- batch driver (examples/test_xworld.cpp:34-61 loop shape) ---- */
uint64_t orc_xw_rollout(int n_envs, const orc_xw_cfg *cfg, int n_icons, const orc_icon_info *info,
                        const uint8_t *icons64, int steps, uint32_t policy_seed,
                        uint32_t env_gid0, int render, orc_rollout_stats *st, const orc_rollout_out *out) {
    uint64_t n_steps = 0;
    orc_rollout_stats s;
    memset(&s, 0, sizeof s);
    orc_xworld *w = orc_xw_create(cfg, n_icons, info, render ? icons64 : NULL);
    size_t sz = screen_size(w) * (size_t)w->cfg.context;
    uint8_t *obs = (uint8_t *)malloc(sz ? sz : 1);
    for (int e = 0; e < n_envs; ++e) {
        uint32_t episode = 0;
        /* one object stands in for every env in turn: each env has a curriculum history of its own */
        w->cur_level = w->cfg.start_level; w->cur_counter = 0;
        memset(w->use_len, 0, sizeof w->use_len); memset(w->use_sum, 0, sizeof w->use_sum); memset(w->use_head, 0, sizeof w->use_head);
        orc_xw_reset_game(w, env_gid0 + (uint32_t)e, episode);
        for (int t = 0; t < steps; ++t) {
            if (orc_xw_game_over(w) != ORC_ALIVE) {
                episode++;
                orc_xw_reset_game(w, env_gid0 + (uint32_t)e, episode);
                s.resets++;
            }
            if (render) orc_xw_get_state_screen(w, obs);
            int a = orc_policy_action(policy_seed, env_gid0 + (uint32_t)e, (uint32_t)t, orc_xw_num_actions(w));
            float r = orc_xw_take_actions(w, a, 1);
            int code = orc_xw_game_over(w);
            s.reward_sum += r;
            if (out) {
                size_t k = (size_t)t * (size_t)n_envs + (size_t)e;
                if (out->rewards) out->rewards[k] = r;
                if (out->codes) out->codes[k] = (uint8_t)code;
                if (out->obs_ck && render) out->obs_ck[k] = orc_obs_checksum(obs, sz);
            }
            n_steps++;
        }
    }
    memcpy(s.task_perf, w->perf, sizeof s.task_perf);
    free(obs);
    orc_xw_destroy(w);
    if (st) *st = s;
    return n_steps;
}
