/*
 * oracle/simple_race.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates games/simple_race/simple_race_simulator.{h,cpp} (dynamics, reward,
 * observation; the OpenCV drawing / GUI code is out of scope) plus the
 * GameSimulator base behaviour, in SimulatorInterface call order.
 *
 * Arithmetic contract.  The reference only compiles with pre-GCC-6 headers
 * (simple_race_simulator.cpp:271,416 use std::min(1.0d, ...)), where the
 * unqualified cos/sin/sqrt/acos/floor/fabs/round resolve to the C *double*
 * functions.  Every expression below therefore spells out where a value is
 * float and where it is double, exactly as C++'s usual arithmetic conversions
 * give for the reference source; cv::Point2f members are float and
 * cv::Point_ operators narrow with saturate_cast<float>.  This file must be
 * compiled with -ffp-contract=off (no FMA contraction), see oracle/Makefile.
 *
 * Pinning: the reference has no SimpleRace test and its C++ cannot be built
 * here without stand-in headers -> pinned only by the known-answer values in
 * SURVEY.md 8(a) (tests/test_oracle_simple_race.py).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.1415926            /* simple_race_simulator.h:39 (double literal) */
static const int WINDOW_WIDTH = 480;   /* simple_race_simulator.cpp:34 */
static const int WINDOW_HEIGHT = 720;  /* simple_race_simulator.cpp:35 */

typedef struct { float x, y; } pt2f;   /* cv::Point2f */

struct orc_simple_race {
    orc_race_cfg cfg;
    /* Track (one track in the pool, SimpleRaceGame ctor cpp:444-458) */
    float width;                       /* Track::_width */
    /* StraightTrack */
    pt2f  mid_pos, start_pos, end_pos;
    float length;
    /* CircleTrack */
    pt2f  center;
    float inner_radius, outer_radius;
    /* CircleCar / BaseCar */
    pt2f  pos;
    float angle;
    /* RaceEngine */
    float last_reward;
    int   steps;
    float delta_fwd, delta_ang;
    int   legal_actions[9];
    int   n_legal;
    /* GameSimulator */
    int64_t num_steps;
    float  *screens;                   /* context * 4, oldest first */
    /* thread-local RNG of the reference (random mode) */
    orc_minstd reng;
};

/* ---------------------------------------------------------------- Track -- */
/* StraightTrack::StraightTrack, cpp:105-110 */
static void straight_init(orc_simple_race *g, float x, float y, float length, float width) {
    g->mid_pos.x = x; g->mid_pos.y = y;
    g->length = length;
    g->width = width;
    pt2f d0 = {0.0f, (float)(0.4 * (double)g->length)};
    pt2f d1 = {0.0f, (float)(0.6 * (double)g->length)};
    g->start_pos.x = g->mid_pos.x - d0.x; g->start_pos.y = g->mid_pos.y - d0.y;
    g->end_pos.x = g->mid_pos.x + d1.x;   g->end_pos.y = g->mid_pos.y + d1.y;
}

/* CircleTrack::CircleTrack, cpp:55-59 */
static void circle_init(orc_simple_race *g, float cx, float cy, float inner_radius, float width) {
    g->center.x = cx; g->center.y = cy;
    g->inner_radius = inner_radius;
    g->width = width;
    g->outer_radius = inner_radius + g->width;
}

/* cv::norm(Point_<float>) -> double */
static double cv_norm(pt2f p) {
    return sqrt((double)p.x * p.x + (double)p.y * p.y);
}

/* StraightTrack::out_of_bound cpp:182-186 ; CircleTrack::out_of_bound cpp:75-79 */
static int track_out_of_bound(const orc_simple_race *g, pt2f pos) {
    if (g->cfg.track_type == 1) {
        pt2f d = {pos.x - g->center.x, pos.y - g->center.y};
        float r = (float)cv_norm(d);
        return r < g->inner_radius || r > g->outer_radius;
    }
    return (pos.x < g->mid_pos.x - g->width / 2) || (pos.x > g->mid_pos.x + g->width / 2) ||
           (pos.y < g->start_pos.y) || (pos.y > g->end_pos.y);
}

/* StraightTrack::race_finish cpp:188-190 ; Track::race_finish h:53 */
static int track_race_finish(const orc_simple_race *g, pt2f pos) {
    if (g->cfg.track_type == 1) return 0;
    return pos.y > g->end_pos.y;
}

/* horizontal_displacement: straight cpp:202-204, circle cpp:92-95 */
static float track_h_disp(const orc_simple_race *g, pt2f p) {
    if (g->cfg.track_type == 1) {
        pt2f rel = {p.x - g->center.x, p.y - g->center.y};
        return (float)((2 * cv_norm(rel) - (double)g->inner_radius - (double)g->outer_radius) /
                       (double)g->width);
    }
    return 2 * (p.x - g->mid_pos.x) / g->width;
}

/* vertical_displacement: straight cpp:210-212 ; Track default 0 (h:60) */
static float track_v_disp(const orc_simple_race *g, pt2f p) {
    if (g->cfg.track_type == 1) return 0;
    return 2 * (p.y - g->mid_pos.y) / g->length;
}

/* get_tangent_vec: straight cpp:218-220, circle cpp:101-104 */
static pt2f track_tangent(const orc_simple_race *g, pt2f p) {
    pt2f t;
    if (g->cfg.track_type == 1) {
        pt2f u = {g->center.y - p.y, p.x - g->center.x};
        double s = 1 / cv_norm(u);
        t.x = (float)((double)u.x * s);
        t.y = (float)((double)u.y * s);
        return t;
    }
    t.x = 0.0f; t.y = 1.0f;
    return t;
}

/* get_start_pos: straight cpp:192-200, circle cpp:81-90.  `u` supplies
 * util::get_rand_range_val(1.0) values in call order. */
static pt2f track_start_pos(const orc_simple_race *g, int random, float u_first, float u_second) {
    pt2f p;
    if (g->cfg.track_type == 1) {
        if (!random) {
            p.x = (g->inner_radius + g->width / 2) + g->center.x;
            p.y = 0.0f + g->center.y;
        } else {
            float theta = (float)((double)(u_first * 2) * PI);
            float r = g->inner_radius + u_second * g->width;
            pt2f q = {(float)((double)r * orc_trig_cos((double)theta)), (float)((double)r * orc_trig_sin((double)theta))};
            p.x = q.x + g->center.x;
            p.y = q.y + g->center.y;
        }
        return p;
    }
    if (!random) return g->start_pos;
    float dy = u_first * g->length / 2;
    float dx = (float)(((double)u_second - 0.5) * (double)g->width);
    p.x = dx + g->start_pos.x;
    p.y = dy + g->start_pos.y;
    return p;
}

/* ------------------------------------------------------------------ Car -- */
/* BaseCar::move, cpp:227-235 */
static void car_move(orc_simple_race *g, float d, float da) {
    g->angle += da;
    if ((double)g->angle > 2 * PI)
        g->angle = (float)((double)g->angle - 2 * PI);
    else if (g->angle < 0)
        g->angle = (float)((double)g->angle + 2 * PI);
    pt2f dir = {(float)orc_trig_cos((double)g->angle), (float)orc_trig_sin((double)g->angle)};
    pt2f step = {d * dir.x, d * dir.y};   /* float * Point2f */
    g->pos.x += step.x;
    g->pos.y += step.y;
}

/* BaseCar::set_angle(bool), cpp:237-243 */
static void car_set_angle(orc_simple_race *g, int random, float u) {
    if (random) g->angle = (float)((double)(u * 2) * PI);
    else g->angle = (float)(PI / 2);
}

/* ----------------------------------------------------------- RaceEngine -- */
/* RaceEngine::get_reward, cpp:386-410 */
static float engine_get_reward(const orc_simple_race *g, float forward, float angle) {
    pt2f p = g->pos;
    pt2f t = track_tangent(g, p);
    float vx = (float)orc_trig_cos((double)angle), vy = (float)orc_trig_sin((double)angle);
    float reward_speed = (vx * t.x + vy * t.y) * forward;
    float reward_finish = track_race_finish(g, p) ? 2.0f : 0.0f;
    float reward_boundary = 0;
    if (!g->cfg.difficulty_hard) {
        reward_boundary = (float)(-fabs((double)track_h_disp(g, p)));
    } else {
        int hit_boundary = track_out_of_bound(g, p) && !track_race_finish(g, p);
        reward_boundary = hit_boundary ? -2.0f : 0.0f;
    }
    float reward = reward_finish + reward_boundary + reward_speed;
    return (float)((double)reward * g->cfg.reward_scale);
}

/* RaceEngine::act, cpp:290-341 (lock_step keyboard branch is GUI, out of scope) */
static float engine_act(orc_simple_race *g, int a) {
    g->steps++;
    int action_id = a;
    float d_forward = 0.0f, d_turn = 0.0f;
    switch (action_id % 3) {
        case 0: break;
        case 1: d_forward = g->delta_fwd; break;
        case 2: d_forward = -g->delta_fwd;
    }
    action_id /= 3;
    switch (action_id % 3) {
        case 0: break;
        case 1: d_turn = g->delta_ang; break;
        case 2: d_turn = -g->delta_ang;
    }
    car_move(g, d_forward, d_turn);
    g->last_reward = engine_get_reward(g, d_forward, g->angle);
    return g->last_reward;
}

/* RaceEngine::get_screen, cpp:412-430 */
static void engine_get_screen(const orc_simple_race *g, float *state) {
    pt2f t = track_tangent(g, g->pos);
    float a = g->angle;
    double c = (double)t.x * orc_trig_cos((double)a) + (double)t.y * orc_trig_sin((double)a);
    float cos_theta = (float)fmax(-1.0, fmin(1.0, c));
    float sin_theta = (float)sqrt((double)(1 - cos_theta * cos_theta));
    if (orc_trig_cos((double)a) * (double)t.y + orc_trig_sin((double)a) * (double)t.x < 0) sin_theta = -sin_theta;
    state[0] = cos_theta;
    state[1] = sin_theta;
    state[2] = track_h_disp(g, g->pos);
    state[3] = track_v_disp(g, g->pos);
}

/* RaceEngine::reset_game, cpp:267-284.  Draw order in random mode:
 * track index, start-pos first, start-pos second, angle. */
static void engine_reset_with(orc_simple_race *g, float u_track, float u_a, float u_b, float u_angle) {
    (void)u_track;   /* one track in the pool: round(max(0, min(floor(u*1), 0))) == 0 */
    g->pos = track_start_pos(g, g->cfg.random, u_a, u_b);
    car_set_angle(g, g->cfg.random, u_angle);
    g->steps = 0;
    g->last_reward = 0.0f;
}

static void engine_reset(orc_simple_race *g) {
    float u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    if (g->cfg.random) {
        u0 = orc_minstd_rand_range(&g->reng, 1.0f);
        u1 = orc_minstd_rand_range(&g->reng, 1.0f);
        u2 = orc_minstd_rand_range(&g->reng, 1.0f);
        u3 = orc_minstd_rand_range(&g->reng, 1.0f);
    }
    engine_reset_with(g, u0, u1, u2, u3);
}

/* ---------------------------------------------------- GameSimulator base -- */
static void make_context_screens(orc_simple_race *g) {
    float cur[4];
    engine_get_screen(g, cur);
    memmove(g->screens, g->screens + 4, sizeof(float) * 4 * (size_t)(g->cfg.context - 1));
    memcpy(g->screens + 4 * (g->cfg.context - 1), cur, sizeof cur);
}

static void init_screen(orc_simple_race *g) {
    memset(g->screens, 0, sizeof(float) * 4 * (size_t)g->cfg.context);
    make_context_screens(g);
}

/* ------------------------------------------------------------- public ---- */
void orc_race_default_cfg(orc_race_cfg *c) {
    memset(c, 0, sizeof *c);
    c->track_type = 0;
    c->track_width = 20.0f;      /* cpp:18 */
    c->track_length = 100.0f;    /* cpp:19 */
    c->track_radius = 30.0f;     /* cpp:20 */
    c->race_full_manouver = 0;   /* cpp:21-23 */
    c->random = 0;               /* cpp:24 */
    c->difficulty_hard = 0;      /* cpp:25 */
    c->reward_scale = 1.0;       /* cpp:26 */
    c->max_steps = 0;
    c->context = 1;
    c->simulator_seed = 0;
    c->nth_thread = 1;
}

orc_simple_race *orc_race_create(const orc_race_cfg *c) {
    orc_simple_race *g = (orc_simple_race *)calloc(1, sizeof *g);
    g->cfg = *c;
    if (g->cfg.context < 1) g->cfg.context = 1;
    /* RaceEngine::RaceEngine cpp:257-261 */
    g->delta_ang = (float)(PI / 10);
    g->delta_fwd = 1;
    /* get_action_set cpp:432-440 */
    if (c->race_full_manouver) {
        for (int i = 0; i < 9; ++i) g->legal_actions[i] = i;
        g->n_legal = 9;
    } else {
        g->legal_actions[0] = 4; g->legal_actions[1] = 7;
        g->n_legal = 2;
    }
    /* SimpleRaceGame::SimpleRaceGame cpp:444-458 */
    float cx = (float)(WINDOW_WIDTH / 2), cy = (float)(WINDOW_HEIGHT / 2);
    if (c->track_type == 1) {
        float r_in = (float)c->track_radius, width = (float)c->track_width;
        circle_init(g, cx, cy, r_in, width);
    } else {
        float length = (float)c->track_length, width = (float)c->track_width;
        straight_init(g, cx, cy, length, width);
    }
    /* CircleCar::CircleCar() -> BaseCar(): pos (0,0), angle PI/2 */
    g->pos.x = 0.0f; g->pos.y = 0.0f; g->angle = (float)(PI / 2);
    if (c->simulator_seed) orc_minstd_seed_thread(&g->reng, c->simulator_seed, c->nth_thread);
    else orc_minstd_seed(&g->reng, 1);
    g->screens = (float *)calloc(4 * (size_t)g->cfg.context, sizeof(float));
    engine_reset(g);      /* ctor calls reset_game(), cpp:457 */
    g->num_steps = 0;
    return g;
}

void orc_race_destroy(orc_simple_race *g) {
    if (!g) return;
    free(g->screens); free(g);
}

/* SimulatorInterface::reset_game: SimpleRaceGame::reset_game cpp:460-463 -> init_screen */
void orc_race_reset_game(orc_simple_race *g) {
    engine_reset(g);
    g->num_steps = 0;
    init_screen(g);
}

void orc_race_reset_game_with(orc_simple_race *g, float u_track, float u_a, float u_b, float u_angle) {
    engine_reset_with(g, u_track, u_a, u_b, u_angle);
    g->num_steps = 0;
    init_screen(g);
}

/* SimulatorInterface::take_actions -> GameSimulator::take_actions -> SimpleRaceGame::take_action cpp:469-476 */
float orc_race_take_actions(orc_simple_race *g, int action, int act_rep) {
    float reward = 0;
    g->num_steps++;
    if (action < 0 || action >= g->n_legal) abort();
    for (int i = 0; i < act_rep; ++i) reward += engine_act(g, g->legal_actions[action]);
    float r = 0;
    r += reward;
    make_context_screens(g);
    return r;
}

/* SimpleRaceGame::game_over cpp:465-467 ; RaceEngine::game_over cpp:286-288 */
int orc_race_game_over(const orc_simple_race *g) {
    int base = (g->cfg.max_steps > 0 && g->num_steps >= g->cfg.max_steps) ? ORC_MAX_STEP : ORC_ALIVE;
    return base | (track_out_of_bound(g, g->pos) ? ORC_DEAD : ORC_ALIVE);
}

int orc_race_get_lives(const orc_simple_race *g) { (void)g; return 1; }   /* cpp:503 */
int orc_race_num_actions(const orc_simple_race *g) { return g->n_legal; }
int64_t orc_race_num_steps(const orc_simple_race *g) { return g->num_steps; }

void orc_race_get_car(const orc_simple_race *g, float *x, float *y, float *angle) {
    *x = g->pos.x; *y = g->pos.y; *angle = g->angle;
}

void orc_race_set_car(orc_simple_race *g, float x, float y, float angle) {
    g->pos.x = x; g->pos.y = y; g->angle = angle;
}

void orc_race_get_screen(const orc_simple_race *g, float *out4) { engine_get_screen(g, out4); }

void orc_race_get_state_screen(const orc_simple_race *g, float *out) {
    memcpy(out, g->screens, sizeof(float) * 4 * (size_t)g->cfg.context);
}

/* ---- batch driver (examples/test_simple_race.cpp:26-53 loop shape).  In random mode the reset
 * uniforms come from the xwb-rng-v1 stream (seed, env, episode, 0): track, start #1, start #2, angle. ---- */
static void race_reset_stream(orc_simple_race *g, uint32_t seed, uint32_t gid, uint32_t episode) {
    if (!g->cfg.random) { orc_race_reset_game(g); return; }
    orc_stream rs;
    orc_stream_init(&rs, seed, gid, episode, 0);
    float u0 = orc_stream_unit(&rs), u1 = orc_stream_unit(&rs), u2 = orc_stream_unit(&rs), u3 = orc_stream_unit(&rs);
    orc_race_reset_game_with(g, u0, u1, u2, u3);
}

uint64_t orc_race_rollout(int n_envs, const orc_race_cfg *cfg, uint32_t seed, int steps, uint32_t policy_seed,
                          uint32_t env_gid0, orc_rollout_stats *st, const orc_rollout_out *out) {
    uint64_t n_steps = 0;
    orc_rollout_stats s;
    memset(&s, 0, sizeof s);
    size_t osz = sizeof(float) * 4 * (size_t)(cfg->context < 1 ? 1 : cfg->context);
    float *obs = (float *)malloc(osz);
    for (int e = 0; e < n_envs; ++e) {
        orc_simple_race *g = orc_race_create(cfg);
        uint32_t episode = 0;
        race_reset_stream(g, seed, env_gid0 + (uint32_t)e, episode);
        int na = orc_race_num_actions(g);
        for (int t = 0; t < steps; ++t) {
            if (orc_race_game_over(g) != ORC_ALIVE) {
                episode++;
                race_reset_stream(g, seed, env_gid0 + (uint32_t)e, episode);
                s.resets++;
            }
            orc_race_get_state_screen(g, obs);
            int a = orc_policy_action(policy_seed, env_gid0 + (uint32_t)e, (uint32_t)t, na);
            float r = orc_race_take_actions(g, a, 1);
            int code = orc_race_game_over(g);
            s.reward_sum += r;
            if (out) {
                size_t k = (size_t)t * (size_t)n_envs + (size_t)e;
                if (out->rewards) out->rewards[k] = r;
                if (out->codes) out->codes[k] = (uint8_t)code;
                if (out->obs_ck) out->obs_ck[k] = orc_obs_checksum(obs, osz);
            }
            n_steps++;
        }
        orc_race_destroy(g);
    }
    free(obs);
    if (st) *st = s;
    return n_steps;
}
