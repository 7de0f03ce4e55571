// xwb_checkpoint.hip -- host side of libxwb.so, part 4: xwb_state_bytes / xwb_save_state / xwb_load_state.
#include "xwb_sim.h"

using namespace xwb;
using namespace xwb::host;

extern "C" {

// ---- checkpoint ----
extern "C++" {
namespace {
struct StateArray { void *ptr; size_t bytes; };

std::vector<StateArray> state_arrays(xwb_sim *s, bool include_obs) {
    const size_t n = (size_t)s->n;
    std::vector<StateArray> a;
    auto add = [&](void *p, size_t bytes) { if (p) a.push_back(StateArray{p, bytes}); };
    add(s->d_actions, n * 4); add(s->d_num_steps, n * 4); add(s->d_episode, n * 4); add(s->d_reward, n * 4);
    add(s->d_done, n); add(s->d_success, n); add(s->d_err, 4); add(s->d_reset_partial, ((n + 255) / 256) * 4);
    add(s->d_pos, n * 4); add(s->d_flags, n);
    add(s->d_x, n * 4); add(s->d_y, n * 4); add(s->d_angle, n * 4);
    add(s->d_minstd, n * 4);
    if (s->cfg.game == XWB_XWORLD2D) {
        const size_t cells = (size_t)s->cfg.max_dim * s->cfg.max_dim;
        add(s->d_grid, n * cells * 2); add(s->d_agent, n * 4); add(s->d_task_steps, n * 4); add(s->d_task_state, n * 4);
        add(s->d_task_steps2, n * 4); add(s->d_task_state2, n * 4); add(s->d_grp_order, n);
        // (the done list and its counter rotate through two / three buffers: the current ones are saved, a load rewinds the rotation)
        add(s->d_done_list + (size_t)s->list_sel * n, n * 4); add(s->d_done_count + s->count_sel, 4); add(s->d_fresh, n); add(s->d_perf, 40 * 8);
        add(s->d_goal_cells, n * XW_MAX_GOALS); add(s->d_cand2d, n * 4); add(s->d_agent_dir, n); add(s->d_sent_names, n * 4);
        add(s->d_goal_warp, n * XW_MAX_GOALS * 6 * sizeof(double));     // goal images are re-warped from these on load
        add(s->d_cur_level, n); add(s->d_cur_counter, n * 4); add(s->d_cur_usage, n * 9 * XW_USAGE_BYTES);
    }
    if (include_obs) add(s->d_obs, n * s->obs_bytes_per_env);
    return a;
}

// version 3 (round 4): per-workgroup reset counts (d_reset_partial, sized by num_envs) replaced the single counter, the
// exclusive schedule's group order and the task performance counters joined, count_sel lost its rc_sel bit
// version 4 (round 6): ONE done counter (the current one of the rotation) instead of the pair
constexpr uint32_t XWB_STATE_VERSION = 4;
struct StateHeader {
    char magic[8];
    uint32_t version, game, num_envs, include_obs, n_arrays, policy_step, count_sel, list_valid;
    uint64_t obs_bytes_per_env, cfg_hash;
};

uint64_t config_hash(const xwb_config &c) {            // everything that shapes the state; pointers excluded
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    const int32_t v[] = {c.game, c.num_envs, c.context, c.max_steps, c.array_size, c.track_type, c.race_full_manouver, c.random,
                         c.difficulty_hard, c.map_kind, c.max_dim, c.dim, c.num_goals, c.num_blocks, c.max_steps_factor, c.task_mode,
                         c.n_tasks, c.color, c.visible_radius, c.obs_format, c.n_icons};
    mix(v, sizeof v); mix(c.tasks, sizeof c.tasks);
    mix(&c.seed, 4); mix(&c.policy_seed, 4); mix(&c.env_gid0, 4);
    mix(&c.rng_mode, 4); mix(&c.simulator_seed, 4); mix(&c.thread_base, 4);
    mix(&c.n_tasks2, 4); mix(c.tasks2, sizeof c.tasks2); mix(&c.task_schedule2, 4); mix(c.task_weights2, sizeof c.task_weights2);
    mix(&c.task_groups_exclusive, 4); mix(&c.task_group_weight, 8); mix(&c.task_group_weight2, 8);
    mix(&c.curriculum, 8); mix(&c.start_level, 4); mix(&c.task_schedule, 4); mix(c.task_weights, sizeof c.task_weights); mix(&c.no_wall_shadow, 4);
    return h;
}
}  // namespace
}  // extern "C++"

int xwb_state_bytes(xwb_sim *s, int32_t include_obs, size_t *bytes) {
    if (!s || !bytes) return fail(XWB_ERR_ARG, "NULL argument");
    size_t total = sizeof(StateHeader);
    for (auto &a : state_arrays(s, include_obs != 0)) total += 8 + a.bytes;
    *bytes = total;
    return XWB_OK;
}

int xwb_save_state(xwb_sim *s, int32_t include_obs, uint8_t *out_host, size_t cap) {
    if (!s || !out_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    size_t need = 0;
    xwb_state_bytes(s, include_obs, &need);
    if (cap < need) return fail(XWB_ERR_ARG, "buffer smaller than xwb_state_bytes");
    HIP_TRY(hipDeviceSynchronize());
    const auto arrays = state_arrays(s, include_obs != 0);
    StateHeader h{};
    memcpy(h.magic, "XWBSTATE", 8);
    h.version = XWB_STATE_VERSION; h.game = (uint32_t)s->cfg.game; h.num_envs = (uint32_t)s->n; h.include_obs = include_obs ? 1u : 0u;
    h.n_arrays = (uint32_t)arrays.size(); h.policy_step = s->policy_step; h.count_sel = 0;
    h.list_valid = (s->list_valid ? 1u : 0u) | (s->autoreset_done ? 2u : 0u); h.obs_bytes_per_env = s->obs_bytes_per_env; h.cfg_hash = config_hash(s->cfg);
    uint8_t *w = out_host;
    memcpy(w, &h, sizeof h); w += sizeof h;
    for (auto &a : arrays) {
        const uint64_t b = a.bytes;
        memcpy(w, &b, 8); w += 8;
        HIP_TRY(hipMemcpy(w, a.ptr, a.bytes, hipMemcpyDeviceToHost));
        w += a.bytes;
    }
    return XWB_OK;
}

int xwb_load_state(xwb_sim *s, const uint8_t *in_host, size_t bytes) {
    if (!s || !in_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (bytes < sizeof(StateHeader)) return fail(XWB_ERR_ARG, "not a state blob");
    StateHeader h;
    memcpy(&h, in_host, sizeof h);
    if (memcmp(h.magic, "XWBSTATE", 8) != 0) return fail(XWB_ERR_ARG, "not a state blob");
    if (h.version != XWB_STATE_VERSION)
        return fail(XWB_ERR_ARG, "state blob version " + std::to_string(h.version) + ", this library reads version " + std::to_string(XWB_STATE_VERSION) +
                                 " (the array layout changed: save again with this library)");
    if (h.game != (uint32_t)s->cfg.game || h.num_envs != (uint32_t)s->n || h.obs_bytes_per_env != s->obs_bytes_per_env ||
        h.cfg_hash != config_hash(s->cfg))
        return fail(XWB_ERR_ARG, "state blob was saved from a batch with another configuration");
    HIP_TRY(hipDeviceSynchronize());
    s->count_sel = 0; s->list_sel = 0;                  // the saved list and counter become the rotation's current ones
    const auto arrays = state_arrays(s, h.include_obs != 0);
    if (arrays.size() != h.n_arrays) return fail(XWB_ERR_ARG, "state blob layout mismatch");
    if (s->d_done_count) HIP_TRY(hipMemset(s->d_done_count, 0, 3 * sizeof(int32_t)));
    const uint8_t *r = in_host + sizeof h, *end = in_host + bytes;
    for (auto &a : arrays) {
        uint64_t b;
        if (r + 8 > end) return fail(XWB_ERR_ARG, "truncated state blob");
        memcpy(&b, r, 8); r += 8;
        if (b != a.bytes || r + b > end) return fail(XWB_ERR_ARG, "state blob layout mismatch");
        HIP_TRY(hipMemcpy(a.ptr, r, a.bytes, hipMemcpyHostToDevice));
        r += b;
    }
    s->shadow_ok = false; s->regen_pending = false; s->step_lazy = false; s->regen_deferred = false; s->snap_ok = false; s->step_fused = false;
    s->frame_src = 0; s->draws_since_pack = 0;
    s->policy_step = h.policy_step; s->list_valid = (h.list_valid & 1u) != 0; s->autoreset_done = (h.list_valid & 2u) != 0;
    if (s->cfg.game == XWB_XWORLD2D) {
        XwParams p = xw_params(s);
        if (p.visible_radius) HIP_TRY(launch_xw_warp_goals(p, false, nullptr));
        if (!h.include_obs) {                           // frames from the state; older context frames start black
            HIP_TRY(hipMemset(s->d_fresh, 2, (size_t)s->n));
            HIP_TRY(launch_xw_render(p, 0, nullptr));
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return XWB_OK;
}

}  // extern "C"
