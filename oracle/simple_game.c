/*
 * oracle/simple_game.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates games/simple_game/simple_game_simulator.{h,cpp} plus the
 * GameSimulator base behaviour it inherits (simulator.cpp:36-117,
 * simulator.h:68-74) in the call order of SimulatorInterface
 * (simulator_interface.cpp:95-143).  One struct == one reference object.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

struct orc_simple_game {
    /* SimpleGameEngine members, simple_game_simulator.h:55-58 */
    int      array_size;
    uint8_t *state_vec;     /* GameFrame _state_vec */
    float   *rewards;       /* std::vector<float> _rewards */
    int      cur_pos;
    /* GameSimulator members, simulator.h:193-198 */
    int64_t  num_steps;
    uint8_t *screens;       /* context * array_size, oldest first (simulator.cpp:51-60) */
    /* flags */
    int      max_steps;     /* FLAGS_max_steps */
    int      context;       /* FLAGS_context */
};

static const float MOVE_REWARD = -0.1f;   /* simple_game_simulator.h:52 */
static const float DEST_REWARD = 4.0f;    /* simple_game_simulator.h:53 */

/* SimpleGameEngine::reset_game, simple_game_simulator.cpp:31-38 */
static void engine_reset_game(orc_simple_game *g) {
    g->cur_pos = g->array_size / 2;
    memset(g->state_vec, 0, (size_t)g->array_size);
    g->state_vec[g->cur_pos] = 1;
    for (int i = 0; i < g->array_size; ++i) g->rewards[i] = 0.0f;
    g->rewards[g->array_size - 1] = DEST_REWARD / 2;
    g->rewards[0] = DEST_REWARD;
}

/* SimpleGameEngine::game_over, simple_game_simulator.cpp:40-42 */
static int engine_game_over(const orc_simple_game *g) {
    return g->cur_pos <= 0 || g->cur_pos >= g->array_size - 1;
}

/* valid_range, simple_game_simulator.h:47-49 */
static int valid_range(const orc_simple_game *g) {
    return g->cur_pos >= 0 && g->cur_pos < g->array_size;
}

/* SimpleGameEngine::get_reward, simple_game_simulator.cpp:69-76 */
static float engine_get_reward(orc_simple_game *g) {
    float reward = MOVE_REWARD;
    if (valid_range(g) && g->rewards[g->cur_pos] != 0.0) {
        reward = g->rewards[g->cur_pos];
        g->rewards[g->cur_pos] = 0.0f;
    }
    return reward;
}

/* SimpleGameEngine::act, simple_game_simulator.cpp:44-63 */
static float engine_act(orc_simple_game *g, int action_id) {
    if (engine_game_over(g)) return engine_get_reward(g);
    switch (action_id) {
        case 0:
            g->state_vec[g->cur_pos] = 0;
            --g->cur_pos;
            break;
        case 1:
            g->state_vec[g->cur_pos] = 0;
            ++g->cur_pos;
            break;
        default:
            abort(); /* LOG(FATAL) << "undefined action_id" */
    }
    if (valid_range(g)) g->state_vec[g->cur_pos] = 1;
    return engine_get_reward(g);
}

/* GameSimulator::make_context_screens + shift_context<uint8_t>, simulator.cpp:51-85 */
static void make_context_screens(orc_simple_game *g) {
    size_t sz = (size_t)g->array_size;
    memmove(g->screens, g->screens + sz, sz * (size_t)(g->context - 1));
    memcpy(g->screens + sz * (size_t)(g->context - 1), g->state_vec, sz);
}

/* GameSimulator::init_screen, simulator.cpp:110-113 (zeros, then one shift) */
static void init_screen(orc_simple_game *g) {
    memset(g->screens, 0, (size_t)g->context * (size_t)g->array_size);
    make_context_screens(g);
}

orc_simple_game *orc_sg_create(int array_size, int max_steps, int context) {
    orc_simple_game *g = (orc_simple_game *)calloc(1, sizeof *g);
    g->array_size = array_size;
    g->max_steps = max_steps;
    g->context = context < 1 ? 1 : context;
    g->state_vec = (uint8_t *)calloc((size_t)array_size, 1);
    g->rewards = (float *)calloc((size_t)array_size, sizeof(float));
    g->screens = (uint8_t *)calloc((size_t)g->context * (size_t)array_size, 1);
    /* SimpleGame::SimpleGame, simple_game_simulator.cpp:82-85 */
    engine_reset_game(g);
    g->num_steps = 0;
    return g;
}

void orc_sg_destroy(orc_simple_game *g) {
    if (!g) return;
    free(g->state_vec); free(g->rewards); free(g->screens); free(g);
}

/* SimulatorInterface::reset_game, simulator_interface.cpp:95-105:
 *   SimpleGame::reset_game (cpp:87-90) -> GameSimulator::reset_game (simulator.cpp:115-117) -> init_screen */
void orc_sg_reset_game(orc_simple_game *g) {
    engine_reset_game(g);
    g->num_steps = 0;
    init_screen(g);
}

/* SimulatorInterface::take_actions (simulator_interface.cpp:126-137)
 *   -> GameSimulator::take_actions (simulator.cpp:98-108): num_steps_++ once, act_rep x take_action
 *   -> SimpleGame::take_action (simple_game_simulator.cpp:96-103): CHECK_LT(action_id, 2)
 *   -> make_context_screens */
float orc_sg_take_actions(orc_simple_game *g, int action, int act_rep) {
    float reward = 0;
    g->num_steps++;
    if (action < 0 || action >= 2) abort();
    for (int i = 0; i < act_rep; ++i) reward += engine_act(g, action);
    float r = 0;
    r += reward;
    make_context_screens(g);
    return r;
}

/* SimpleGame::game_over, simple_game_simulator.cpp:92-94 | GameSimulator::game_over, simulator.h:68-74 */
int orc_sg_game_over(const orc_simple_game *g) {
    int base = (g->max_steps > 0 && g->num_steps >= g->max_steps) ? ORC_MAX_STEP : ORC_ALIVE;
    return base | (engine_game_over(g) ? ORC_SUCCESS : ORC_ALIVE);
}

/* SimpleGame::get_lives, simple_game_simulator.cpp:137 */
int orc_sg_get_lives(const orc_simple_game *g) { return orc_sg_game_over(g) ? 0 : 1; }

int64_t orc_sg_num_steps(const orc_simple_game *g) { return g->num_steps; }
int orc_sg_pos(const orc_simple_game *g) { return g->cur_pos; }

/* SimpleGame::get_screen, simple_game_simulator.cpp:105-110 */
void orc_sg_get_screen(const orc_simple_game *g, uint8_t *out) {
    memcpy(out, g->state_vec, (size_t)g->array_size);
}

/* get_state_data -> fill_in_reward_and_screen, simulator.cpp:87-96 */
void orc_sg_get_state_screen(const orc_simple_game *g, uint8_t *out) {
    memcpy(out, g->screens, (size_t)g->context * (size_t)g->array_size);
}

/* ---- batch driver (reference example loop, python/examples/test_simple_game.py:15-30) ---- */
uint64_t orc_sg_rollout(int n_envs, int array_size, int context, int steps, uint32_t policy_seed,
                        uint32_t env_gid0, orc_rollout_stats *st, const orc_rollout_out *out) {
    uint64_t n_steps = 0;
    orc_rollout_stats s;
    memset(&s, 0, sizeof s);
    size_t osz = (size_t)array_size * (size_t)(context < 1 ? 1 : context);
    uint8_t *obs = (uint8_t *)malloc(osz);
    for (int e = 0; e < n_envs; ++e) {
        orc_simple_game *g = orc_sg_create(array_size, 0, context);
        orc_sg_reset_game(g);
        for (int t = 0; t < steps; ++t) {
            if (orc_sg_game_over(g) != ORC_ALIVE) { orc_sg_reset_game(g); s.resets++; }
            orc_sg_get_state_screen(g, obs);
            int a = orc_policy_action(policy_seed, env_gid0 + (uint32_t)e, (uint32_t)t, 2);
            float r = orc_sg_take_actions(g, a, 1);
            int code = orc_sg_game_over(g);
            s.reward_sum += r;
            if (out) {
                size_t k = (size_t)t * (size_t)n_envs + (size_t)e;
                if (out->rewards) out->rewards[k] = r;
                if (out->codes) out->codes[k] = (uint8_t)code;
                if (out->obs_ck) out->obs_ck[k] = orc_obs_checksum(obs, osz);
            }
            n_steps++;
        }
        orc_sg_destroy(g);
    }
    free(obs);
    if (st) *st = s;
    return n_steps;
}
