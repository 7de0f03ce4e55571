// xwb_getters.hip -- host side of libxwb.so, part 3: device-pointer getters, static queries, per-env host access (the scalar
// SimulatorInterface surface), map replay hooks, StatePacket / sentences / task performance.
#include "xwb_sim.h"
#include "xwb_language.h"
#include "../../include/xwb_trig.h"
#include "../../include/xwb_minstd.h"

#include <unistd.h>

using namespace xwb;
using namespace xwb::host;

namespace {

// ---- StatePacket wire writer (data_packet.h:313-319, data_packet.cpp:143-162, memory_util.h:307-333) ----
struct Writer {
    uint8_t *p; size_t cap, n;
    void put(const void *d, size_t len) { if (p && n + len <= cap) memcpy(p + n, d, len); n += len; }
    void u64(uint64_t v) { put(&v, 8); }
    void str(const char *s) { size_t len = strlen(s); u64(len); put(s, len + 1); }
};

}  // namespace

extern "C" {

int xwb_ego_render_path(xwb_sim *s, int32_t *path) {
    if (!s || !path) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D || s->cfg.visible_radius == 0) return fail(XWB_ERR_STATE, "not an egocentric xworld batch");
    *path = xw_ego_span(s->xw) ? 1 : 0;
    return XWB_OK;
}

int xwb_obs_dev(xwb_sim *s, void **ptr, size_t *bytes_per_env) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (ptr) *ptr = s->d_obs;
    if (bytes_per_env) *bytes_per_env = s->obs_bytes_per_env;
    return XWB_OK;
}

#define XWB_GETTER(NAME, TYPE, FIELD)                                   \
    int NAME(xwb_sim *s, TYPE **ptr) {                                  \
        if (!s || !ptr) return fail(XWB_ERR_ARG, "NULL argument");      \
        *ptr = s->FIELD;                                                \
        return XWB_OK;                                                  \
    }
XWB_GETTER(xwb_reward_dev, float, d_reward)
XWB_GETTER(xwb_game_over_dev, uint8_t, d_done)
XWB_GETTER(xwb_actions_dev, int32_t, d_actions)
XWB_GETTER(xwb_num_steps_dev, int32_t, d_num_steps)
XWB_GETTER(xwb_success_dev, uint8_t, d_success)
XWB_GETTER(xwb_episode_dev, uint32_t, d_episode)
XWB_GETTER(xwb_minstd_state_dev, uint32_t, d_minstd)

uint32_t xwb_minstd_seed_thread(int32_t simulator_seed, int32_t nth_thread) {
    // simulator_util.cpp:48-50: int seed = std::hash<std::string>()(std::to_string(FLAGS_simulator_seed + (++__num_threads)));
    // reng_.seed(seed) -- libstdc++'s own hash, as in the reference's build
    const int seed = (int)std::hash<std::string>()(std::to_string(simulator_seed + nth_thread));
    return xwb_minstd_seed_value((int64_t)seed);
}
int32_t xwb_minstd_rand_ind(uint32_t *state, int32_t size) { return (state && size >= 1) ? xwb_minstd_rand_ind_state(state, size) : -1; }
float xwb_minstd_rand_range(uint32_t *state, float upper) { return state ? xwb_minstd_rand_range_state(state, upper) : 0.0f; }

int xwb_xw_grid_dev(xwb_sim *s, uint16_t **ptr) {
    if (!s || !ptr) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    *ptr = s->d_grid;
    return XWB_OK;
}

int xwb_done_count(xwb_sim *s, void *stream, int32_t *n_done) {
    if (!s || !n_done) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    if (s->cfg.game != XWB_XWORLD2D) {                     // per-workgroup counts of the last launch that reset envs
        std::vector<int32_t> part((size_t)(s->n + 255) / 256);
        HIP_TRY(hipMemcpyAsync(part.data(), s->d_reset_partial, part.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int64_t total = 0;
        for (int32_t v : part) total += v;
        *n_done = (int32_t)total;
        return XWB_OK;
    }
    HIP_TRY(hipMemcpyAsync(n_done, s->d_done_count + s->count_sel, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return XWB_OK;
}

int xwb_get_num_actions(const xwb_sim *s, int32_t *n) {
    if (!s || !n) return fail(XWB_ERR_ARG, "NULL argument");
    *n = s->num_actions;
    return XWB_OK;
}

int xwb_get_screen_out_dimensions(const xwb_sim *s, size_t *h, size_t *w, size_t *c) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (h) *h = (size_t)s->out_h;
    if (w) *w = (size_t)s->out_w;
    if (c) *c = (size_t)s->out_c;
    return XWB_OK;
}

int xwb_get_world_dimensions(const xwb_sim *s, double *X, double *Y, double *Z) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    // SimulatorInterface::get_world_dimensions: only teaching environments answer (xworld_simulator.cpp:100-104)
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "get_world_dimensions: not a teaching environment");
    if (X) *X = s->cfg.max_dim;
    if (Y) *Y = s->cfg.max_dim;
    if (Z) *Z = 0;
    return XWB_OK;
}

int xwb_num_envs(const xwb_sim *s, int32_t *n) {
    if (!s || !n) return fail(XWB_ERR_ARG, "NULL argument");
    *n = s->n;
    return XWB_OK;
}

int xwb_get_env_state(xwb_sim *s, int32_t env, void *stream, xwb_env_state *o) {
    if (!s || !o) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    hipStream_t st = as_stream(stream);
    memset(o, 0, sizeof *o);
    uint8_t done = 0, succ = 0;
    int32_t steps = 0, act = -1;
    HIP_TRY(hipMemcpyAsync(&o->reward, s->d_reward + env, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&done, s->d_done + env, 1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&succ, s->d_success + env, 1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&steps, s->d_num_steps + env, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&act, s->d_actions + env, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&o->episode, s->d_episode + env, 4, hipMemcpyDeviceToHost, st));
    int32_t axy = 0, ts = 0, tsteps = 0, ts2 = 0, tsteps2 = 0;
    if (s->cfg.game == XWB_SIMPLE_GAME) {
        HIP_TRY(hipMemcpyAsync(&o->sg_pos, s->d_pos + env, 4, hipMemcpyDeviceToHost, st));
    } else if (s->cfg.game == XWB_SIMPLE_RACE) {
        HIP_TRY(hipMemcpyAsync(&o->race_x, s->d_x + env, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&o->race_y, s->d_y + env, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&o->race_angle, s->d_angle + env, 4, hipMemcpyDeviceToHost, st));
    } else {
        HIP_TRY(hipMemcpyAsync(&axy, s->d_agent + env, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&ts, s->d_task_state + env, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&tsteps, s->d_task_steps + env, 4, hipMemcpyDeviceToHost, st));
        if (s->d_task_state2) {
            HIP_TRY(hipMemcpyAsync(&ts2, s->d_task_state2 + env, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(&tsteps2, s->d_task_steps2 + env, 4, hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    o->game_over = done;
    o->num_steps = steps;
    o->last_action = act;
    o->last_action_success = succ;
    // get_lives: SimpleGame cpp:137 / XWorldSimulator :506 -> game_over ? 0 : 1 ; SimpleRace cpp:503 -> 1
    o->lives = s->cfg.game == XWB_SIMPLE_RACE ? 1 : (done ? 0 : 1);
    if (s->cfg.game == XWB_XWORLD2D) {
        o->xw_agent_x = axy & 0xffff; o->xw_agent_y = axy >> 16;
        o->xw_task = (ts >> 24) & 0xf;
        o->xw_target = (int16_t)(ts & 0xffff);
        o->xw_target_name = o->xw_task == XWB_TASK_TARGET ? o->xw_target : -1;
        o->xw_stage = (ts >> 16) & 0xf;
        o->xw_event = (ts >> 20) & 0xf;
        o->xw_steps_in_task = tsteps;
        if (s->d_task_state2) {
            o->xw_task2 = (ts2 >> 24) & 0xf; o->xw_target2 = (int16_t)(ts2 & 0xffff); o->xw_stage2 = (ts2 >> 16) & 0xf;
            o->xw_event2 = (ts2 >> 20) & 0xf; o->xw_steps_in_task2 = tsteps2;
        }
        uint8_t dir = 1;
        HIP_TRY(hipMemcpy(&dir, s->d_agent_dir + env, 1, hipMemcpyDeviceToHost));
        o->xw_agent_dir = dir;
        o->xw_level = 0; o->xw_check_counter = 0;
        if (s->d_cur_level) {
            uint8_t lv = 0;
            HIP_TRY(hipMemcpy(&lv, s->d_cur_level + env, 1, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&o->xw_check_counter, s->d_cur_counter + env, 4, hipMemcpyDeviceToHost));
            o->xw_level = lv;
        }
        uint32_t sn = 0xffffffffu;
        HIP_TRY(hipMemcpy(&sn, s->d_sent_names + env, 4, hipMemcpyDeviceToHost));
        o->xw_sentence_names = sn;
        o->xw_group_first = o->xw_group_ran = -1;
        if (s->d_grp_order) {
            uint8_t go = 0;
            HIP_TRY(hipMemcpy(&go, s->d_grp_order + env, 1, hipMemcpyDeviceToHost));
            o->xw_group_first = go & 1; o->xw_group_ran = (go >> 1) & 1;
        }
    }
    return XWB_OK;
}

namespace {
int copy_out(xwb_sim *s, void *dst, const void *src, size_t bytes, void *stream) {
    if (!s || !dst) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
    hipPointerAttribute_t attr;
    const bool device = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (!device) { (void)hipGetLastError(); HIP_TRY(hipStreamSynchronize(st)); }
    return XWB_OK;
}
}  // namespace

int xwb_get_obs(xwb_sim *s, void *dst, size_t bytes, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (bytes != (size_t)s->n * s->obs_bytes_per_env) return fail(XWB_ERR_ARG, "bytes must be num_envs * bytes_per_env");
    return copy_out(s, dst, s->d_obs, bytes, stream);
}
int xwb_get_reward(xwb_sim *s, float *dst, void *stream) { return s ? copy_out(s, dst, s->d_reward, (size_t)s->n * 4, stream) : fail(XWB_ERR_ARG, "sim is NULL"); }
int xwb_get_done(xwb_sim *s, uint8_t *dst, void *stream) { return s ? copy_out(s, dst, s->d_done, (size_t)s->n, stream) : fail(XWB_ERR_ARG, "sim is NULL"); }

int xwb_get_env_obs(xwb_sim *s, int32_t env, void *stream, void *out_host, size_t bytes) {
    if (!s || !out_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    if (bytes != s->obs_bytes_per_env) return fail(XWB_ERR_ARG, "bytes must equal bytes_per_env");
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(out_host, static_cast<uint8_t *>(s->d_obs) + (size_t)env * bytes, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return XWB_OK;
}

int xwb_get_env_grid(xwb_sim *s, int32_t env, void *stream, uint16_t *out_host) {
    if (!s || !out_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    hipStream_t st = as_stream(stream);
    const size_t cells = (size_t)s->cfg.max_dim * s->cfg.max_dim;
    HIP_TRY(hipMemcpyAsync(out_host, s->d_grid + (size_t)env * cells, cells * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return XWB_OK;
}

int xwb_xw_load_map_task(xwb_sim *s, int32_t env, const uint16_t *grid_host, int32_t agent_x, int32_t agent_y,
                         int32_t dim, int32_t task, int32_t target) {
    if (!s || !grid_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    if (task < XWB_TASK_TARGET || task > XWB_TASK2D_BETWEEN) return fail(XWB_ERR_ARG, "unknown task id");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (s->cfg.n_tasks2 > 0) return fail(XWB_ERR_STATE, "map replay is for batches with one task group");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    if (s->d_cur_level) {
        if (dim < 3 || dim > 8) return fail(XWB_ERR_ARG, "dim is not one of the curriculum's levels (3..8)");
    } else if (dim != s->cfg.dim) return fail(XWB_ERR_ARG, "dim differs from the batch's dim");
    const int D = s->cfg.max_dim;
    if (agent_x < 0 || agent_y < 0 || agent_x >= D || agent_y >= D) return fail(XWB_ERR_ARG, "agent outside the map");
    {   // the goal-slot table (XW_MAX_GOALS cells per env) is what the step kernel's "bumped into a goal" test reads
        int goals = 0;
        for (int c = 0; c < D * D; ++c) {
            const int icon = (int)(grid_host[c] & XWB_CELL_ICON_MASK) - 1;
            if (icon >= s->cfg.n_icons) return fail(XWB_ERR_ARG, "cell code beyond the palette");
            if (icon >= 0 && s->icon_type_h[icon] == XWB_ICON_GOAL) goals++;
        }
        if (goals > XW_MAX_GOALS) return fail(XWB_ERR_ARG, "a map holds at most 16 goals");
    }
    HIP_TRY(hipDeviceSynchronize());
    s->shadow_ok = false; s->regen_pending = false; s->regen_deferred = false; s->snap_ok = false;
    const size_t cells = (size_t)D * D;
    int32_t axy = agent_x | (agent_y << 16);
    const bool is2d = task >= XWB_TASK2D_TARGET;
    if (is2d != (s->xw.group2d != 0)) return fail(XWB_ERR_ARG, "task is not of this batch's task family");
    if (s->d_cur_level) {                                                // the level whose dims the map has
        const uint8_t lv = (uint8_t)(dim - 3);
        HIP_TRY(hipMemcpy(s->d_cur_level + env, &lv, 1, hipMemcpyHostToDevice));
    }
    // stage NAV, no event (xw_device.h); a 2-D-native task without a target stays in its idle stage
    const int stage = is2d && target < 0 ? 0 : 1;
    int32_t ts = (target & 0xffff) | (stage << 16) | (task << 24);
    if (is2d) {
        // the per-episode candidate tables of the step-time idle stages: goal slots in row-major order; reachable =
        // same component as the agent with the blocks as the only obstacles (xworld_task.py:347-357)
        std::vector<uint8_t> gc(XW_MAX_GOALS, 0xff), seen(cells, 0);
        std::vector<int> queue{agent_y * D + agent_x};
        seen[queue[0]] = 1;
        const int lo = (D - dim) / 2, hi = lo + dim;                     // XWorldEnv.set_dims offsets
        for (size_t h = 0; h < queue.size(); ++h) {
            const int c = queue[h], cx = c % D, cy = c / D;
            const int nb[4][2] = {{cx - 1, cy}, {cx + 1, cy}, {cx, cy - 1}, {cx, cy + 1}};
            for (auto &q : nb) {
                if (q[0] < lo || q[1] < lo || q[0] >= hi || q[1] >= hi) continue;
                const int nc = q[1] * D + q[0];
                const int icon = (int)(grid_host[nc] & XWB_CELL_ICON_MASK) - 1;
                if (icon >= s->cfg.n_icons) return fail(XWB_ERR_ARG, "cell code beyond the palette");
                if (seen[nc] || (icon >= 0 && s->icon_type_h[icon] == XWB_ICON_BLOCK)) continue;
                seen[nc] = 1;
                queue.push_back(nc);
            }
        }
        uint32_t cand = 0;
        int slot = 0;
        for (size_t c = 0; c < cells && slot < XW_MAX_GOALS; ++c) {
            const int icon = (int)(grid_host[c] & XWB_CELL_ICON_MASK) - 1;
            if (icon < 0 || icon >= s->cfg.n_icons || s->icon_type_h[icon] != XWB_ICON_GOAL) continue;
            gc[slot] = (uint8_t)c;
            if (seen[c]) cand |= (1u << slot) | (s->icon_colored_h[icon] ? (1u << (16 + slot)) : 0u);
            slot++;
        }
        HIP_TRY(hipMemcpy(s->d_goal_cells + (size_t)env * XW_MAX_GOALS, gc.data(), XW_MAX_GOALS, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_cand2d + env, &cand, 4, hipMemcpyHostToDevice));
    } else {
        // goal slot -> cell table (slots in row-major order): the egocentric render finds a goal's pose through it
        std::vector<uint8_t> gc(XW_MAX_GOALS, 0xff);
        int slot = 0;
        for (size_t c = 0; c < cells && slot < XW_MAX_GOALS; ++c) {
            const int icon = (int)(grid_host[c] & XWB_CELL_ICON_MASK) - 1;
            if (icon >= 0 && icon < s->cfg.n_icons && s->icon_type_h[icon] == XWB_ICON_GOAL) gc[slot++] = (uint8_t)c;
        }
        HIP_TRY(hipMemcpy(s->d_goal_cells + (size_t)env * XW_MAX_GOALS, gc.data(), XW_MAX_GOALS, hipMemcpyHostToDevice));
        if (s->d_goal_warp) {                               // default pose: yaw 1.5707963, scale 1, offset 0 = the identity warp
            const double ident[6] = {1, 0, 0, 0, 1, 0};
            for (int i = 0; i < XW_MAX_GOALS; ++i)
                HIP_TRY(hipMemcpy(s->d_goal_warp + ((size_t)env * XW_MAX_GOALS + i) * 6, ident, sizeof ident, hipMemcpyHostToDevice));
        }
    }
    int32_t zero = 0;
    uint8_t z8 = 0, one = 2;
    HIP_TRY(hipMemcpy(s->d_grid + (size_t)env * cells, grid_host, cells * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_agent + env, &axy, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_task_state + env, &ts, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_task_steps + env, &zero, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_num_steps + env, &zero, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_done + env, &z8, 1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_fresh + env, &one, 1, hipMemcpyHostToDevice));
    // init_screen of that env: render the one-entry list
    XwParams p = xw_params(s);
    int32_t cnt = 1;
    HIP_TRY(hipMemcpy(p.done_list, &env, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.done_count, &cnt, 4, hipMemcpyHostToDevice));
    if (p.visible_radius) HIP_TRY(launch_xw_warp_goals(p, true, nullptr));
    HIP_TRY(launch_xw_render(p, 1, nullptr));
    s->frame_src = 0; s->draws_since_pack += 2;            // (xwb_xw_pack_grids: one env redrawn out of turn -- a context ring elsewhere
                                                           //  cannot follow that: context > 1 must re-synchronise with the screens)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(p.done_count, 0, 4));
    s->list_valid = false;
    return XWB_OK;
}

int xwb_xw_load_map(xwb_sim *s, int32_t env, const uint16_t *grid_host, int32_t agent_x, int32_t agent_y,
                    int32_t target_name, int32_t dim) {
    if (!s || !grid_host) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    // XWorld3DNavTarget: every goal named target_name is a target
    const size_t cells = (size_t)s->cfg.max_dim * s->cfg.max_dim;
    std::vector<uint16_t> g(grid_host, grid_host + cells);
    for (auto &c : g) {
        const int icon = (int)(c & XWB_CELL_ICON_MASK) - 1;
        c &= XWB_CELL_ICON_MASK;
        if (icon >= s->cfg.n_icons) return fail(XWB_ERR_ARG, "cell code beyond the palette");
        if (icon >= 0 && s->icon_type_h[icon] == XWB_ICON_GOAL && s->icon_name_h[icon] == target_name) c |= XWB_CELL_TARGET;
    }
    return xwb_xw_load_map_task(s, env, g.data(), agent_x, agent_y, dim, XWB_TASK_TARGET, target_name);
}

int xwb_xw_set_agent_dir(xwb_sim *s, int32_t env, int32_t dir) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    if (s->cfg.game != XWB_XWORLD2D || s->cfg.visible_radius == 0) return fail(XWB_ERR_STATE, "not an egocentric xworld batch");
    if (env < 0 || env >= s->n || dir < 0 || dir > 3) return fail(XWB_ERR_ARG, "env or dir out of range");
    HIP_TRY(hipDeviceSynchronize());
    const uint8_t d = (uint8_t)dir;
    HIP_TRY(hipMemcpy(s->d_agent_dir + env, &d, 1, hipMemcpyHostToDevice));
    return XWB_OK;
}

int xwb_xw_set_goal_pose(xwb_sim *s, int32_t env, int32_t cell_x, int32_t cell_y, double yaw, double scale, double offset) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    if (s->cfg.game != XWB_XWORLD2D || s->cfg.visible_radius == 0) return fail(XWB_ERR_STATE, "not an egocentric xworld batch");
    const int D = s->cfg.max_dim;
    if (env < 0 || env >= s->n || cell_x < 0 || cell_y < 0 || cell_x >= D || cell_y >= D) return fail(XWB_ERR_ARG, "env or cell out of range");
    HIP_TRY(hipDeviceSynchronize());
    uint8_t gc[XW_MAX_GOALS];
    HIP_TRY(hipMemcpy(gc, s->d_goal_cells + (size_t)env * XW_MAX_GOALS, XW_MAX_GOALS, hipMemcpyDeviceToHost));
    int slot = -1;
    for (int i = 0; i < XW_MAX_GOALS; ++i) if (gc[i] == cell_y * D + cell_x) slot = i;
    if (slot < 0) return fail(XWB_ERR_ARG, "no goal at that cell");
    // XItem::get_item_image (xitem.cpp:46-60) + the inversion cv::warpAffine performs
    const double angle = (90 - yaw * 180 / 3.14159265358979323846) * 3.1415926535897932384626433832795 / 180;
    double sn, cs;                                  // include/xwb_trig.h: the reset kernel's arithmetic, bit for bit
    xwb_sincos(angle, &sn, &cs);
    const double alpha = cs * scale, beta = sn * scale;
    double M[6] = {alpha, beta, (1 - alpha) * 32.0 - beta * 32.0, -beta, alpha, beta * 32.0 + (1 - alpha) * 32.0};
    M[2] += (offset + scale / 2 - 0.5) * 64;
    M[5] += (offset + scale / 2 - 0.5) * 64;
    double Dt = M[0] * M[4] - M[1] * M[3];
    Dt = Dt != 0 ? 1. / Dt : 0;
    const double A11 = M[4] * Dt, A22 = M[0] * Dt;
    M[0] = A11; M[1] *= -Dt; M[3] *= -Dt; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    HIP_TRY(hipMemcpy(s->d_goal_warp + ((size_t)env * XW_MAX_GOALS + slot) * 6, M, sizeof M, hipMemcpyHostToDevice));
    return XWB_OK;
}

int xwb_xw_refresh_obs(xwb_sim *s, int32_t env) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    HIP_TRY(hipDeviceSynchronize());
    XwParams p = xw_params(s);
    const int32_t cnt = 1;
    const uint8_t two = 2;
    HIP_TRY(hipMemcpy(p.done_list, &env, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p.done_count, &cnt, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_fresh + env, &two, 1, hipMemcpyHostToDevice));
    if (p.visible_radius) HIP_TRY(launch_xw_warp_goals(p, true, nullptr));
    HIP_TRY(launch_xw_render(p, 1, nullptr));
    s->frame_src = 0; s->draws_since_pack += 2;            // (xwb_xw_pack_grids: one env redrawn out of turn -- a context ring elsewhere
                                                           //  cannot follow that: context > 1 must re-synchronise with the screens)
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(p.done_count, 0, 4));
    s->list_valid = false;
    return XWB_OK;
}

int xwb_race_set_car(xwb_sim *s, int32_t env, float x, float y, float angle) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    if (s->cfg.game != XWB_SIMPLE_RACE) return fail(XWB_ERR_STATE, "not a simple_race batch");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(s->d_x + env, &x, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_y + env, &y, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_angle + env, &angle, 4, hipMemcpyHostToDevice));
    return XWB_OK;
}

int xwb_get_extra_info(xwb_sim *s, int32_t env, void *stream, char *out, size_t cap) {
    if (!s || !out || cap == 0) return fail(XWB_ERR_ARG, "NULL argument");
    out[0] = 0;
    if (s->cfg.game != XWB_XWORLD2D) return XWB_OK;
    xwb_env_state st;
    int rc = xwb_get_env_state(s, env, stream, &st);
    if (rc) return rc;
    static const char *tasks[] = {"XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection",
                                  "XWorld3DNavTargetAvoid", "XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"};
    static const char *events[] = {"", "correct_goal", "wrong_goal", "time_up"};
    const char *task = st.xw_task >= 0 && st.xw_task < 9 ? tasks[st.xw_task] : "";
    const char *event = st.xw_event >= 0 && st.xw_event < 4 ? events[st.xw_event] : "";
    // xworld_.actual_height() / actual_width() (xworld.h:59,70): the level's dims under FLAGS_curriculum
    const int dim = s->d_cur_level ? 3 + st.xw_level : s->cfg.dim;
    snprintf(out, cap, "%d|task:%s,event:%s,height:%d,width:%d", (int)getpid(), task, event, dim, dim);
    return XWB_OK;
}

// BatchedSimulator.sentence / _group_sentence (xworld_amd/batched.py), on this side of the ABI
static int group_sentence(xwb_sim *s, int32_t env, void *stream, const xwb_env_state &st, int task, int stage, int event, int target,
                          int steps_in_task, std::string *out) {
    out->clear();
    const uint32_t gid = s->cfg.env_gid0 + (uint32_t)env;
    if (task == 5 || task == 7) {
        // 2-D-native Target / ColorTarget: they speak on the teach() call that picked the target, and "Time up ." on the
        // one_channel step that runs out of time (xworld_task.py:205-211): back to idle with the target still recorded
        if (stage == 0 && event == 0 && target >= 0 && st.num_steps > 0 && s->cfg.task_mode == XWB_TASKMODE_ONE_CHANNEL) {
            *out = xwb::lang::sentence_2d_timeup(task);
            return XWB_OK;
        }
        if (stage != 1 || steps_in_task != 0 || target < 0) return XWB_OK;
        uint16_t code = 0;
        const int cells = s->cfg.max_dim * s->cfg.max_dim;
        if (target >= cells) return XWB_OK;
        HIP_TRY(hipMemcpyAsync(&code, s->d_grid + (size_t)env * cells + target, 2, hipMemcpyDeviceToHost, as_stream(stream)));
        HIP_TRY(hipStreamSynchronize(as_stream(stream)));
        const int icon = (int)(code & 0x7fffu) - 1;           // (xw_device.h CELL_ICON_MASK: bit 15 marks target goals)
        if (icon < 0 || icon >= (int)s->icon_names.size()) return XWB_OK;   // (two groups: the 3-D stage may have moved the goal away since)
        *out = xwb::lang::sentence_2d(task, s->icon_names[icon], s->icon_colors[icon], s->cfg.seed, gid, st.episode, (uint32_t)st.num_steps);
        return XWB_OK;
    }
    const uint32_t sn = st.xw_sentence_names;
    *out = xwb::lang::sentence(task, stage, event, s->goal_names, sn & 0xffffu, sn >> 16, task == 3 && target >= 0 ? (target >> 8) & 7 : 0,
                               s->cfg.seed, gid, st.episode);
    return XWB_OK;
}

static int env_sentence(xwb_sim *s, int32_t env, void *stream, std::string *out) {
    xwb_env_state st;
    int rc = xwb_get_env_state(s, env, stream, &st);
    if (rc) return rc;
    if (st.xw_group_ran == 1)      // exclusive scheduling: only the group the last teach() ran can have spoken
        return group_sentence(s, env, stream, st, st.xw_task2, st.xw_stage2, st.xw_event2, st.xw_target2, st.xw_steps_in_task2, out);
    rc = group_sentence(s, env, stream, st, st.xw_task, st.xw_stage, st.xw_event, st.xw_target, st.xw_steps_in_task, out);
    if (rc || st.xw_group_ran == 0) return rc;
    // two task groups run side by side: the first one (conf order) that speaks wins -- Task::teacher_speak only records into
    // an empty buffer (teaching_task.cpp:118-127)
    if (out->empty() && s->cfg.n_tasks2 > 0)
        rc = group_sentence(s, env, stream, st, st.xw_task2, st.xw_stage2, st.xw_event2, st.xw_target2, st.xw_steps_in_task2, out);
    return rc;
}

int xwb_set_names(xwb_sim *s, const char *const *goal_names, int32_t n_goal_names, const char *const *icon_names,
                  const char *const *icon_colors, int32_t n_icons) {
    if (!s || !goal_names || !icon_names || !icon_colors) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (n_icons != s->cfg.n_icons || n_goal_names < 0) return fail(XWB_ERR_ARG, "one name and one colour per icon of the palette");
    for (int i = 0; i < n_icons; ++i) {
        if (!icon_names[i] || !icon_colors[i]) return fail(XWB_ERR_ARG, "NULL name");
        if (s->icon_type_h[i] == 0 && (s->icon_name_h[i] < 0 || s->icon_name_h[i] >= n_goal_names))
            return fail(XWB_ERR_ARG, "a goal icon's name id has no string");
    }
    for (int i = 0; i < n_goal_names; ++i) if (!goal_names[i]) return fail(XWB_ERR_ARG, "NULL name");
    s->goal_names.assign(goal_names, goal_names + n_goal_names);
    s->icon_names.assign(icon_names, icon_names + n_icons);
    s->icon_colors.assign(icon_colors, icon_colors + n_icons);
    s->have_names = true;
    return XWB_OK;
}

int xwb_sentence(xwb_sim *s, int32_t env, void *stream, char *out, size_t cap, size_t *need) {
    if (!s || !need) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (!s->have_names) return fail(XWB_ERR_STATE, "xwb_set_names has not been called: the library only has name ids");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    XWB_ON_DEVICE(s);
    std::string str;
    const int rc = env_sentence(s, env, stream, &str);
    if (rc) return rc;
    *need = str.size() + 1;
    if (out && cap >= str.size() + 1) memcpy(out, str.c_str(), str.size() + 1);
    return XWB_OK;
}

static int copy_out(const std::string &str, char *out, size_t cap, size_t *need) {
    *need = str.size() + 1;
    if (out && cap >= str.size() + 1) memcpy(out, str.c_str(), str.size() + 1);
    return XWB_OK;
}

int xwb_language_sentence(int32_t task, int32_t stage, int32_t event, const char *const *goal_names, int32_t n_goal_names,
                          uint32_t name_a, uint32_t name_b, int32_t direction, uint32_t seed, uint32_t gid, uint32_t episode,
                          char *out, size_t cap, size_t *need) {
    if (!need || (n_goal_names > 0 && !goal_names) || n_goal_names < 0) return fail(XWB_ERR_ARG, "NULL argument");
    std::vector<std::string> names;
    for (int i = 0; i < n_goal_names; ++i) { if (!goal_names[i]) return fail(XWB_ERR_ARG, "NULL name"); names.push_back(goal_names[i]); }
    return copy_out(xwb::lang::sentence(task, stage, event, names, name_a, name_b, direction, seed, gid, episode), out, cap, need);
}

int xwb_language_sentence_2d(int32_t task, int32_t timeup, const char *goal_name, const char *color, uint32_t seed, uint32_t gid,
                             uint32_t episode, uint32_t num_steps, char *out, size_t cap, size_t *need) {
    if (!need) return fail(XWB_ERR_ARG, "NULL argument");
    if (timeup) return copy_out(xwb::lang::sentence_2d_timeup(task), out, cap, need);
    if (!goal_name || !color) return fail(XWB_ERR_ARG, "NULL argument");
    return copy_out(xwb::lang::sentence_2d(task, goal_name, color, seed, gid, episode, num_steps), out, cap, need);
}

int xwb_get_state_packet(xwb_sim *s, int32_t env, float reward, void *stream, uint8_t *out_host, size_t cap,
                         size_t *need) {
    if (!s || !need) return fail(XWB_ERR_ARG, "NULL argument");
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    const bool xw = s->cfg.game == XWB_XWORLD2D;
    // float32 frames (SimpleRace; XWorld2D with XWB_OBS_F32) travel as a reals buffer, uint8 frames as pixels
    const bool is_float = s->cfg.game == XWB_SIMPLE_RACE || (xw && s->cfg.obs_format == XWB_OBS_F32);
    const size_t n_screen = is_float ? s->obs_bytes_per_env / 4 : s->obs_bytes_per_env;
    // sizes first
    size_t total = 8;
    total += 8 + 7 + 1 + 8 + 4;                                  // "reward": flags reals, 1 float
    total += 8 + 7 + 1 + 8 + s->obs_bytes_per_env;               // "screen"
    // XWorldSimulator::define_state_specs (:486-493): the teacher's sentence, "-" when it is silent (or when the strings behind
    // the name ids were never handed over: xwb_set_names)
    std::string sent = "-";
    if (xw && s->have_names) {
        XWB_ON_DEVICE(s);
        std::string str;
        const int rcs = env_sentence(s, env, stream, &str);
        if (rcs) return rcs;
        if (!str.empty()) sent = str;
    }
    if (xw) total += 8 + 9 + 1 + 8 + sent.size() + 1;           // "sentence": str
    *need = total;
    if (!out_host || cap < total) return XWB_OK;
    std::vector<uint8_t> screen(s->obs_bytes_per_env);
    int rc = xwb_get_env_obs(s, env, stream, screen.data(), screen.size());
    if (rc) return rc;
    Writer w{out_host, cap, 0};
    w.u64(xw ? 3 : 2);
    w.str("reward");
    uint8_t f = 1; w.put(&f, 1); w.u64(1); w.put(&reward, 4);
    w.str("screen");
    f = is_float ? 1 : 2; w.put(&f, 1); w.u64(n_screen); w.put(screen.data(), screen.size());
    if (xw) {
        w.str("sentence");
        f = 8; w.put(&f, 1); w.str(sent.c_str());
    }
    return XWB_OK;
}

static const char *const TASK_CLASS[9] = {"XWorld3DNavTarget", "XWorld3DNavTargetNear", "XWorld3DNavTargetBetween", "XWorld3DNavTargetDirection",
                                          "XWorld3DNavTargetAvoid", "XWorldNavTarget", "XWorldNavNear", "XWorldNavColorTarget", "XWorldNavBetween"};

int xwb_get_task_performance(xwb_sim *s, void *stream, xwb_task_performance out[9], int64_t *resets) {
    if (!s || !out) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not a teaching environment");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    unsigned long long h[40];
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(h, s->d_perf, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 9; ++k) {
        out[k].successes = (int64_t)h[k * 4]; out[k].failures = (int64_t)h[k * 4 + 1];
        out[k].success_steps = (int64_t)h[k * 4 + 2]; out[k].time_ups = (int64_t)h[k * 4 + 3];
    }
    if (resets) *resets = (int64_t)h[36];
    return XWB_OK;
}

int xwb_task_performance_report(xwb_sim *s, void *stream, char *out, size_t cap, size_t *need) {
    if (!s || !need) return fail(XWB_ERR_ARG, "NULL argument");
    xwb_task_performance perf[9];
    const int rc = xwb_get_task_performance(s, stream, perf, nullptr);
    if (rc) return rc;
    // Teacher::report_task_performance, teacher.cpp:175-200 (an unordered_map there: the order of the blocks is unspecified;
    // here: task id).  Tasks of the batch's groups only; a task that did not occur prints its name line alone.
    std::string text;
    auto add_group = [&](const int32_t *tasks, int n) {
        for (int i = 0; i < n; ++i) {
            const int k = tasks[i];
            if (k < 0 || k >= 9) continue;
            text += std::string("=== ") + TASK_CLASS[k] + " ===\n";
            const long long succ = perf[k].successes, failed = perf[k].failures;
            if (succ + failed == 0) continue;                        // "skip task that did not occur"
            const double per = succ > 0 ? (double)perf[k].success_steps / (double)succ : -1.0;
            char line[160];
            snprintf(line, sizeof line, "=== %lld(S)/%lld(F) -> %g@%g\n", succ, failed, (double)succ / (double)(succ + failed), per);
            text += line;
        }
    };
    static const int32_t only_target[1] = {XWB_TASK_TARGET};
    if (s->cfg.n_tasks > 0) add_group(s->cfg.tasks, s->cfg.n_tasks); else add_group(only_target, 1);
    add_group(s->cfg.tasks2, s->cfg.n_tasks2);
    *need = text.size() + 1;
    if (out && cap >= text.size() + 1) memcpy(out, text.c_str(), text.size() + 1);
    return XWB_OK;
}

int xwb_decode_game_over_code(int32_t code, char *out, size_t cap) {
    if (!out || cap == 0) return fail(XWB_ERR_ARG, "NULL argument");
    std::string sres;
    if (code == 0) sres = "alive";
    else {
        if (code & XWB_MAX_STEP) sres += "max_step|";
        if (code & XWB_DEAD) sres += "dead|";
        if (code & XWB_SUCCESS) sres += "success|";
        if (code & XWB_LOST_LIFE) sres += "lost_life|";
        if (sres.empty()) return fail(XWB_ERR_ARG, "unknown game over code");     // CHECK(!code_str.empty())
        sres.pop_back();
    }
    if (sres.size() + 1 > cap) return fail(XWB_ERR_ARG, "buffer too small");
    memcpy(out, sres.c_str(), sres.size() + 1);
    return XWB_OK;
}

int xwb_xw_get_tile_table(const xwb_sim *s, uint8_t *out_host, size_t cap, size_t *need) {
    if (!s || !need) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    *need = s->tile_table.size();
    if (out_host && cap >= s->tile_table.size()) memcpy(out_host, s->tile_table.data(), s->tile_table.size());
    return XWB_OK;
}

}  // extern "C"
