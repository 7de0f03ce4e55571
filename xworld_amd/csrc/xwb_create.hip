// xwb_create.hip -- host side of libxwb.so, part 1: configuration, set-up of a batch, xwb_create / xwb_destroy
// (SimulatorInterface::SimulatorInterface, simulator_interface.cpp:37-85).
#include "xwb_sim.h"
#include "../../include/xwb_minstd.h"

using namespace xwb;
using namespace xwb::host;

namespace xwb {
namespace host {
namespace { thread_local std::string g_err; }
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
const char *const POISON_MSG = "a device-side queue hand-off was not released within its watchdog (kernels of the batch's two queues did "
                               "not run concurrently, or the device is wedged): the batch is poisoned -- results since the last "
                               "successful xwb_check_errors are void, destroy it (XWB_QUEUE_SYNC=events / xwb_config.queue_sync avoid epochs)";
}  // namespace host
}  // namespace xwb

namespace {

// XWB_DEBUG (include/xwb.h, xwb_config "Debug configuration"): parsed once per process, OR-ed into every batch created
struct DebugEnv { int32_t flags = 0, ego_per = 0, ego_pad = 0, render_shape = 0, ego_miss_blocks = 0; };
const DebugEnv &debug_env() {
    static const DebugEnv d = [] {
        DebugEnv e;
        const char *v = getenv("XWB_DEBUG");
        if (!v) return e;
        std::string all(v);
        size_t pos = 0;
        while (pos <= all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string t = all.substr(pos, end - pos);
            if (t == "no_pregen") e.flags |= XWB_DEBUG_NO_PREGEN;
            else if (t == "no_lazy") e.flags |= XWB_DEBUG_NO_LAZY;
            else if (t == "ego_no_cache") e.flags |= XWB_DEBUG_EGO_NO_CACHE;
            else if (t == "ego_no_span") e.flags |= XWB_DEBUG_EGO_NO_SPAN;
            else if (t == "ego_no_flat") e.flags |= XWB_DEBUG_EGO_NO_FLAT;
            else if (t == "no_fused") e.flags |= XWB_DEBUG_NO_FUSED;
            else if (t.compare(0, 8, "ego_per=") == 0) e.ego_per = atoi(t.c_str() + 8);
            else if (t.compare(0, 8, "ego_pad=") == 0) e.ego_pad = atoi(t.c_str() + 8) + 1;
            else if (t.compare(0, 10, "ego_fused=") == 0) { fprintf(stderr, "xwb: XWB_DEBUG ego_fused: the fused lab kernel was removed (see profiles/NOTES.md)\n"); }
            else if (t.compare(0, 16, "ego_miss_blocks=") == 0) { const int v = atoi(t.c_str() + 16); e.ego_miss_blocks = v >= 4 && v <= 65536 ? (v & ~3) : 0; }
            else if (t == "render_shape=64x2") e.render_shape = 1;
            else if (t == "render_shape=256x2") e.render_shape = 2;
            else if (!t.empty()) fprintf(stderr, "libxwb: XWB_DEBUG: unknown entry '%s' ignored\n", t.c_str());
            pos = end + 1;
        }
        return e;
    }();
    return d;
}

// ---- host restatement of the SimpleRace constructors (float/double conversion points matter) ----
void race_setup(const xwb_config &c, RaceParams &r) {
    const double PI = 3.1415926;                       // simple_race_simulator.h:39
    r.track_type = c.track_type;
    r.random = c.random;
    r.difficulty_hard = c.difficulty_hard;
    r.reward_scale = c.reward_scale;
    r.delta_ang = (float)(PI / 10);                    // RaceEngine ctor, cpp:257-261
    r.delta_fwd = 1;
    if (c.race_full_manouver) { r.n_legal = 9; for (int i = 0; i < 9; ++i) r.legal[i] = i; }
    else { r.n_legal = 2; r.legal[0] = 4; r.legal[1] = 7; }         // get_action_set, cpp:432-440
    const float cx = (float)(480 / 2), cy = (float)(720 / 2);       // WINDOW_WIDTH/HEIGHT, cpp:34-35,446
    if (c.track_type == 1) {                           // CircleTrack ctor, cpp:55-59
        float r_in = (float)c.track_radius, width = (float)c.track_width;
        r.center_x = cx; r.center_y = cy;
        r.inner_radius = r_in;
        r.width = width;
        r.outer_radius = r_in + r.width;
        r.length = 0; r.mid_x = r.mid_y = r.start_x = r.start_y = r.end_x = r.end_y = 0;
    } else {                                           // StraightTrack ctor, cpp:105-110
        float length = (float)c.track_length, width = (float)c.track_width;
        r.mid_x = cx; r.mid_y = cy;
        r.length = length;
        r.width = width;
        float d0 = (float)(0.4 * (double)r.length), d1 = (float)(0.6 * (double)r.length);
        r.start_x = r.mid_x - 0.0f; r.start_y = r.mid_y - d0;
        r.end_x = r.mid_x + 0.0f;   r.end_y = r.mid_y + d1;
        r.center_x = r.center_y = r.inner_radius = r.outer_radius = 0;
    }
}

int round_half_even(float v) { return (int)lrintf(v); }          // cvRound

}  // namespace

namespace xwb {

// The 12x12 tile of one icon = what cv::resize(INTER_LINEAR) makes of that icon's cell when the
// 64 px/cell canvas is shrunk to 12 px/cell (xworld_simulator.cpp:521-522).  The ratio is 16/3 in
// both axes for every map size, so output pixel k of a cell takes source pixels s_k, s_k+1 of the
// *same* cell with 11-bit weights; OpenCV 3.2 fixed-point arithmetic (imgwarp.cpp): horizontal pass
// in int32, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.  Gray: BGR2GRAY
// (B*1868 + G*9617 + R*4899 + 8192) >> 14 applied to the resized BGR tile (cvtColor after resize).
void build_tile_table(const uint8_t *icons64, int n_icons, int channels, uint8_t *out) {
    int tap[12];
    short w0[12], w1[12];
    const double scale = 1.0 / (12.0 / 64.0);
    for (int k = 0; k < 12; ++k) {
        float f = (float)((k + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        tap[k] = s;
        w0[k] = (short)round_half_even((1.f - f) * 2048);
        w1[k] = (short)round_half_even(f * 2048);
    }
    for (int ic = 0; ic < n_icons; ++ic) {
        const uint8_t *src = icons64 + (size_t)ic * 64 * 64 * 3;
        uint8_t bgr[12][12][3];
        for (int py = 0; py < 12; ++py)
            for (int px = 0; px < 12; ++px)
                for (int c = 0; c < 3; ++c) {
                    const uint8_t *r0 = src + (size_t)tap[py] * 64 * 3, *r1 = r0 + 64 * 3;
                    int h0 = r0[tap[px] * 3 + c] * w0[px] + r0[(tap[px] + 1) * 3 + c] * w1[px];
                    int h1 = r1[tap[px] * 3 + c] * w0[px] + r1[(tap[px] + 1) * 3 + c] * w1[px];
                    bgr[py][px][c] = (uint8_t)((((w0[py] * (h0 >> 4)) >> 16) + ((w1[py] * (h1 >> 4)) >> 16) + 2) >> 2);
                }
        uint8_t *dst = out + (size_t)ic * channels * 144;
        for (int py = 0; py < 12; ++py)
            for (int px = 0; px < 12; ++px) {
                if (channels == 3) {
                    for (int c = 0; c < 3; ++c) dst[c * 144 + py * 12 + px] = bgr[py][px][c];
                } else {
                    dst[py * 12 + px] = (uint8_t)((bgr[py][px][0] * 1868 + bgr[py][px][1] * 9617 +
                                                   bgr[py][px][2] * 4899 + (1 << 13)) >> 14);
                }
            }
    }
}

}  // namespace xwb

namespace {

bool curriculum_cfg(const xwb_config &c) { return c.curriculum != 0 && c.map_kind == XWB_MAP_NAV; }

int xw_setup(xwb_sim *s) {
    const xwb_config &c = s->cfg;
    if (c.max_dim < 1 || c.max_dim > XW_MAX_DIM || c.dim < 1 || c.dim > c.max_dim)
        return fail(XWB_ERR_ARG, "xworld: need 1 <= dim <= max_dim <= 16");
    if (c.num_goals < 1 || c.num_goals > XW_MAX_GOALS) return fail(XWB_ERR_ARG, "xworld: need 1 <= num_goals <= 16");
    if (c.task_schedule != XWB_SCHEDULE_RANDOM && c.task_schedule != XWB_SCHEDULE_WEIGHTED) return fail(XWB_ERR_ARG, "xworld: unknown task_schedule");
    if (c.task_schedule == XWB_SCHEDULE_WEIGHTED) {
        if (c.n_tasks < 1) return fail(XWB_ERR_ARG, "xworld: the weighted schedule needs the task list");
        for (int i = 0; i < c.n_tasks; ++i)
            if (!(c.task_weights[i] > 0)) return fail(XWB_ERR_ARG, "A task must have a positive weight");   // teaching_task.cpp:148
    }
    if (c.curriculum != 0 && c.map_kind == XWB_MAP_NAV) {
        // XWorldNav.py:27-30: six levels, dims 3 .. max_h -- the class asserts n_levels == 6, i.e. its 8x8 world
        if (c.max_dim != 8) return fail(XWB_ERR_ARG, "xworld: curriculum != 0 needs XWorldNav's 8x8 world (max_dim 8)");
        if (c.start_level < 0 || c.start_level > 5) return fail(XWB_ERR_ARG, "xworld: start_level must be in 0..5");
    }
    if (c.n_icons < 1 || !c.icons64 || !c.icon_type || !c.icon_name)
        return fail(XWB_ERR_ARG, "xworld: icons64 / icon_type / icon_name are required (the reference loads item_path images)");
    if (c.n_icons > 4000) return fail(XWB_ERR_ARG, "xworld: too many icons");
    if (c.n_tasks < 0 || c.n_tasks > 8) return fail(XWB_ERR_ARG, "xworld: need 0 <= n_tasks <= 8");
    for (int i = 0; i < c.n_tasks; ++i)
        if (c.tasks[i] < XWB_TASK_TARGET || c.tasks[i] > XWB_TASK2D_BETWEEN) return fail(XWB_ERR_ARG, "xworld: unknown task id");
    for (int i = 1; i < c.n_tasks; ++i)
        if ((c.tasks[i] >= XWB_TASK2D_TARGET) != (c.tasks[0] >= XWB_TASK2D_TARGET))
            return fail(XWB_ERR_ARG, "xworld: a task group holds XWorld3DNav* tasks or 2-D-native XWorldNav* tasks, not both");
    if (c.n_tasks2 < 0 || c.n_tasks2 > 8) return fail(XWB_ERR_ARG, "xworld: need 0 <= n_tasks2 <= 8");
    if (c.n_tasks2 > 0) {
        if (c.n_tasks < 1) return fail(XWB_ERR_ARG, "xworld: a second task group needs a first one");
        for (int i = 0; i < c.n_tasks2; ++i) {
            if (c.tasks2[i] < XWB_TASK_TARGET || c.tasks2[i] > XWB_TASK2D_BETWEEN) return fail(XWB_ERR_ARG, "xworld: unknown task id");
            if ((c.tasks2[i] >= XWB_TASK2D_TARGET) != (c.tasks2[0] >= XWB_TASK2D_TARGET))
                return fail(XWB_ERR_ARG, "xworld: a task group holds XWorld3DNav* tasks or 2-D-native XWorldNav* tasks, not both");
        }
        if ((c.tasks2[0] >= XWB_TASK2D_TARGET) == (c.tasks[0] >= XWB_TASK2D_TARGET))
            return fail(XWB_ERR_ARG, "xworld: two task groups: one must hold XWorld3DNav* tasks, the other the 2-D-native ones");
        if (c.task_schedule2 != XWB_SCHEDULE_RANDOM && c.task_schedule2 != XWB_SCHEDULE_WEIGHTED) return fail(XWB_ERR_ARG, "xworld: unknown task_schedule2");
        if (c.task_schedule2 == XWB_SCHEDULE_WEIGHTED)
            for (int i = 0; i < c.n_tasks2; ++i)
                if (!(c.task_weights2[i] > 0)) return fail(XWB_ERR_ARG, "A task must have a positive weight");
    }
    if (!(c.task_group_weight >= 0) || !(c.task_group_weight2 >= 0)) return fail(XWB_ERR_ARG, "xworld: task group weights must be >= 0");
    const int n = s->n, cells = c.max_dim * c.max_dim, ch = c.color ? 3 : 1;
    const bool group2d_cfg = (c.n_tasks > 0 && c.tasks[0] >= XWB_TASK2D_TARGET) || (c.n_tasks2 > 0 && c.tasks2[0] >= XWB_TASK2D_TARGET);
    // goal_cells holds one byte per goal slot with 0xff = "no goal": cell 255 only exists on a 16x16 map
    if (c.max_dim > 15 && (c.visible_radius > 0 || group2d_cfg))
        return fail(XWB_ERR_ARG, "xworld: max_dim 16 is not available with visible_radius > 0 or the 2-D-native task group (<= 15)");
    // name tables (xworld_env.py:247-255): per type, names -> icon variants (icon order = path order)
    int n_names[3] = {0, 0, 0};
    for (int i = 0; i < c.n_icons; ++i) {
        int t = c.icon_type[i];
        if (t < 0 || t > 2 || c.icon_name[i] < 0) return fail(XWB_ERR_ARG, "xworld: bad icon_type / icon_name");
        if (c.icon_name[i] + 1 > n_names[t]) n_names[t] = c.icon_name[i] + 1;
    }
    if (n_names[1] < 1 || n_names[2] < 1 || n_names[0] < 1)
        return fail(XWB_ERR_ARG, "xworld: palette needs at least one goal, one block and one agent icon");
    if (c.map_kind == XWB_MAP_NAV && c.num_goals > n_names[0])
        return fail(XWB_ERR_ARG, "xworld: XWorldNav needs num_goals distinct goal names");
    std::vector<int16_t> first, variants;
    int off[3];
    for (int t = 0; t < 3; ++t) {
        off[t] = (int)first.size();
        for (int nm = 0; nm < n_names[t]; ++nm) {
            first.push_back((int16_t)variants.size());
            int cnt = 0;
            for (int i = 0; i < c.n_icons; ++i)
                if (c.icon_type[i] == t && c.icon_name[i] == nm) { variants.push_back((int16_t)i); cnt++; }
            if (cnt == 0) return fail(XWB_ERR_ARG, "xworld: name ids of a type must be dense");
        }
        first.push_back((int16_t)variants.size());
    }
    // free cells / block capacity checks the reference leaves to Python asserts
    if (c.map_kind == XWB_MAP_NAV) {
        int X = c.dim % 2 == 0 ? c.dim - 1 : c.dim;
        int nodes = ((X + 1) / 2) * ((X + 1) / 2);
        int hashes = X * X - nodes - (nodes - 1) + (c.dim % 2 == 0 ? (X / 2) + (c.dim / 2) : 0);
        if (c.num_blocks > hashes) return fail(XWB_ERR_ARG, "xworld: too many blocks for a valid maze");
        int free_cells = c.dim * c.dim - hashes;
        if (c.num_goals + 1 > free_cells) return fail(XWB_ERR_ARG, "xworld: not enough free cells");
        if (nodes > 64) return fail(XWB_ERR_ARG, "xworld: maze node lattice larger than 8x8");
    } else {
        int walls = std::min(c.num_blocks, c.dim) + std::min(std::max(c.num_blocks - c.dim, 0), c.dim - 1);
        if (c.num_goals + 1 + walls > c.dim * c.dim) return fail(XWB_ERR_ARG, "xworld: not enough free cells");
    }
    // tile table: entry 0 = empty cell (canvas fill 255, xmap.cpp:129-132), entry i+1 = icon i
    s->tile_table.assign((size_t)c.n_icons * ch * 144, 0);
    build_tile_table(c.icons64, c.n_icons, ch, s->tile_table.data());
    std::vector<uint8_t> atlas((size_t)(c.n_icons + 1) * ch * 144, 255);
    memcpy(atlas.data() + (size_t)ch * 144, s->tile_table.data(), s->tile_table.size());
    std::vector<uint8_t> types(c.n_icons);
    std::vector<int16_t> names(c.n_icons);
    for (int i = 0; i < c.n_icons; ++i) { types[i] = (uint8_t)c.icon_type[i]; names[i] = (int16_t)c.icon_name[i]; }
    s->icon_type_h.assign(c.icon_type, c.icon_type + c.n_icons);
    s->icon_name_h.assign(c.icon_name, c.icon_name + c.n_icons);
    s->icon_colored_h.assign(c.n_icons, 0);
    if (c.icon_colored) s->icon_colored_h.assign(c.icon_colored, c.icon_colored + c.n_icons);

    int rc;
    if ((rc = dev_alloc(s, &s->d_grid, (size_t)n * cells))) return rc;
    if ((rc = dev_alloc(s, &s->d_agent, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_task_steps, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_task_state, n))) return rc;
    if (c.n_tasks2 > 0) {
        if ((rc = dev_alloc(s, &s->d_task_state2, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_task_steps2, n))) return rc;
    }
    // simulator_interface.cpp:46-48: lang_acquisition runs the groups non-exclusively whatever the flag says
    const bool exclusive = c.task_groups_exclusive && c.task_mode != XWB_TASKMODE_LANG_ACQ;
    if (exclusive && c.n_tasks2 > 0) {
        if ((rc = dev_alloc(s, &s->d_grp_order, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_idle_list, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_idle_count, 3))) return rc;
    }
    if ((rc = dev_alloc(s, &s->d_perf, 40))) return rc;
    // Pre-generated next episodes: possible where an env's next episode is a pure function of (seed, global id, episode + 1)
    // and the render reads nothing but the grid -- full observation, no curriculum (the level depends on the results so far),
    // no per-env reference engine (its state depends on the draws so far), no exclusive group order carried across resets.
    // (float32 frames stay on the classic paths: their plain whole-batch render variant measured 5-8 % slower than the
    // variants the classic paths use -- 416 vs 385 / 394 us on the C4-sized batch)
    s->pregen = c.visible_radius == 0 && !curriculum_cfg(c) && c.rng_mode != XWB_RNG_MINSTD && !(exclusive && c.n_tasks2 > 0) &&
                c.obs_format == XWB_OBS_U8 && !(c.debug_flags & XWB_DEBUG_NO_PREGEN);
    if (s->pregen) {
        if ((rc = dev_alloc(s, &s->d_sh_ep, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_grid, (size_t)2 * n * cells))) return rc;           // two slots per env
        if ((rc = dev_alloc(s, &s->d_sh_agent, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_task_state, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_task_state2, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_sent_names, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_cand2d, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_goal_cells, (size_t)2 * n * XW_MAX_GOALS))) return rc;
        // the default loop's xwb_step as ONE launch (XwParams::snap_*): frames without a context ring only
        if (c.context == 1 && !(c.debug_flags & XWB_DEBUG_NO_FUSED))
            for (int k = 0; k < 2; ++k) {
                if ((rc = dev_alloc(s, &s->d_snap_grid[k], (size_t)n * cells))) return rc;
            }
    }
    if ((rc = dev_alloc(s, &s->d_done_list, (size_t)2 * n))) return rc;       // two lists, three counters: xwb_sim.h count_sel
    if ((rc = dev_alloc(s, &s->d_done_ep, (size_t)2 * n))) return rc;
    if ((rc = dev_alloc(s, &s->d_done_count, 3))) return rc;
    if ((rc = dev_alloc(s, &s->d_fresh, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_icon_type, ((size_t)c.n_icons + 3) & ~(size_t)3))) return rc;   // the step kernel stages it dword-wise
    if ((rc = dev_alloc(s, &s->d_icon_colored, c.n_icons))) return rc;
    if ((rc = dev_alloc(s, &s->d_goal_cells, (size_t)n * XW_MAX_GOALS, 0xff))) return rc;
    if ((rc = dev_alloc(s, &s->d_cand2d, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_sent_names, n, 0xff))) return rc;
    const bool curriculum = c.curriculum != 0 && c.map_kind == XWB_MAP_NAV;       // XWorldWalls never reads the flag
    // under the curriculum the levels place 2 or 4 goals whatever cfg.num_goals says (XWorldNav.py:27-34): the per-env
    // goal-image cache and every kernel that indexes it use the levels' maximum
    const int img_goals = curriculum ? 4 : c.num_goals;
    if (curriculum) {
        if ((rc = dev_alloc(s, &s->d_cur_level, n, c.start_level))) return rc;
        if ((rc = dev_alloc(s, &s->d_cur_counter, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_cur_usage, (size_t)n * 9 * XW_USAGE_BYTES))) return rc;
    }
    if ((rc = dev_alloc(s, &s->d_sync, 16))) return rc;
    if ((rc = dev_alloc(s, &s->d_term_grid, (size_t)n * cells))) return rc;
    if ((rc = dev_alloc(s, &s->d_term_flag, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_agent_dir, n, 1))) return rc;                 // heading "down": yaw 1.5707963
    if (c.visible_radius > 0) {
        if ((rc = dev_alloc(s, &s->d_goal_warp, (size_t)n * XW_MAX_GOALS * 6))) return rc;
        if ((rc = dev_alloc(s, &s->d_goal_img, (size_t)n * img_goals * 4096))) return rc;
        const size_t npx = (size_t)c.n_icons * 64 * 64;
        std::vector<uint8_t> a4((npx + 2) * 4, 0);
        for (size_t i = 0; i < npx; ++i) for (int k = 0; k < 3; ++k) a4[i * 4 + k] = c.icons64[i * 3 + k];
        for (int k = 0; k < 3; ++k) a4[npx * 4 + k] = 255;            // white pixel, then a black one
        // XItem::get_item_image turns the agent's icon by 90 - yaw degrees about (32, 32) with a white border: the three
        // quarter turns are exact integer maps (source index 64 falls outside): heading right, left, up
        std::vector<uint32_t> rot_off(c.n_icons, 0);
        for (int ic = 0; ic < c.n_icons; ++ic) {
            if (c.icon_type[ic] != XWB_ICON_AGENT) continue;
            rot_off[ic] = (uint32_t)(a4.size() / 4);
            for (int h = 0; h < 3; ++h)
                for (int py = 0; py < 64; ++py)
                    for (int px = 0; px < 64; ++px) {
                        const int ix = h == 0 ? 64 - py : (h == 1 ? py : 64 - px), iy = h == 0 ? px : (h == 1 ? 64 - px : 64 - py);
                        const bool in = ix >= 0 && ix < 64 && iy >= 0 && iy < 64;
                        for (int k = 0; k < 3; ++k) a4.push_back(in ? c.icons64[(((size_t)ic * 64 + iy) * 64 + ix) * 3 + k] : 255);
                        a4.push_back(0);
                    }
        }
        if ((rc = dev_alloc(s, &s->d_atlas64, a4.size()))) return rc;
        HIP_TRY(hipMemcpy(s->d_atlas64, a4.data(), a4.size(), hipMemcpyHostToDevice));
        if ((rc = dev_alloc(s, &s->d_agent_rot, (size_t)c.n_icons))) return rc;
        HIP_TRY(hipMemcpy(s->d_agent_rot, rot_off.data(), rot_off.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(xw_ego_tables(c.visible_radius, c.max_dim, s->out_h, &s->d_ego_taps, &s->xw.ego_fast, &s->ego_cell_edge, &s->xw.ego_span));
        s->allocs.push_back(s->d_ego_taps);
    }
    if ((rc = dev_alloc(s, &s->d_icon_name, c.n_icons))) return rc;
    if ((rc = dev_alloc(s, &s->d_name_first, first.size()))) return rc;
    if ((rc = dev_alloc(s, &s->d_name_variants, variants.size()))) return rc;
    const bool f32 = c.obs_format == XWB_OBS_F32;
    if ((rc = dev_alloc(s, &s->d_atlas, f32 ? atlas.size() : atlas.size() / 4))) return rc;
    HIP_TRY(hipMemcpy(s->d_icon_type, types.data(), types.size(), hipMemcpyHostToDevice));
    if (c.icon_colored) {
        std::vector<uint8_t> col(c.n_icons);
        for (int i = 0; i < c.n_icons; ++i) col[i] = c.icon_colored[i] ? 1 : 0;
        HIP_TRY(hipMemcpy(s->d_icon_colored, col.data(), col.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(s->d_icon_name, names.data(), names.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_name_first, first.data(), first.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_name_variants, variants.data(), variants.size() * 2, hipMemcpyHostToDevice));
    if (f32) {
        // py_simulator.cpp:262-272: `float scale = 1 / 255.0` then pixel * scale, a float32 product
        std::vector<float> af(atlas.size());
        const float scale = (float)(1 / 255.0);
        for (size_t i = 0; i < atlas.size(); ++i) af[i] = (float)atlas[i] * scale;
        HIP_TRY(hipMemcpy(s->d_atlas, af.data(), af.size() * 4, hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemcpy(s->d_atlas, atlas.data(), atlas.size(), hipMemcpyHostToDevice));
    }
    // (a high-priority side queue was tried: no gain beside the renders, and batches created after another one in the same
    // process then failed their resume tests -- left at the default priority)
    HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_step, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_reset, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_term, hipEventDisableTiming | hipEventDisableSystemFence));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_cells, hipEventDisableTiming | hipEventDisableSystemFence));

    XwParams &p = s->xw;
    p.n = n; p.context = c.context; p.max_steps = c.max_steps; p.act_rep = 1; p.auto_reset = 0;
    p.map_kind = c.map_kind; p.max_dim = c.max_dim; p.dim = c.dim; p.num_goals = img_goals;
    p.num_blocks = c.num_blocks; p.max_steps_factor = c.max_steps_factor; p.task_mode = c.task_mode;
    p.channels = ch; p.n_icons = c.n_icons;
    p.obs_f32 = f32 ? 1 : 0;
    p.dbg_ego_per = c.debug_ego_per; p.dbg_ego_pad = c.debug_ego_pad; p.dbg_render_shape = c.debug_render_shape;
    p.dbg_ego_miss_blocks = debug_env().ego_miss_blocks;
    p.n_tasks = c.n_tasks;
    p.group2d = c.n_tasks > 0 && c.tasks[0] >= XWB_TASK2D_TARGET;
    p.curriculum = curriculum ? c.curriculum : 0.0; p.cur_level = s->d_cur_level; p.cur_counter = s->d_cur_counter; p.cur_usage = s->d_cur_usage;
    p.sync = s->d_sync; p.sig_epoch = 0; p.wait_epoch = 0;
    {   // the watchdog's host-visible word (read at the top of every verb, no sync)
        void *hp = nullptr, *dp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, 64, hipHostMallocMapped));
        memset(hp, 0, 64);
        s->h_poison = static_cast<uint32_t *>(hp);
        HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
        p.poison_host = static_cast<uint32_t *>(dp);
    }
    p.minstd = s->d_minstd;
    p.sent_names = s->d_sent_names; p.term_grid = s->d_term_grid; p.term_flag = s->d_term_flag;
    p.goal_cells = s->d_goal_cells; p.cand2d = s->d_cand2d; p.icon_colored = s->d_icon_colored;
    p.visible_radius = c.visible_radius; p.out_dim = s->out_h; p.no_wall_shadow = c.no_wall_shadow;
    p.agent_dir = s->d_agent_dir; p.goal_warp = s->d_goal_warp; p.atlas64 = s->d_atlas64; p.ego_taps = s->d_ego_taps; p.goal_img = s->d_goal_img; p.ego_agent_rot = s->d_agent_rot;
    for (int i = 0; i < 8; ++i) p.tasks[i] = i < c.n_tasks ? c.tasks[i] : 0;
    p.task_weighted = c.task_schedule == XWB_SCHEDULE_WEIGHTED;
    p.n_tasks2 = c.n_tasks2;
    p.group2d_2 = c.n_tasks2 > 0 && c.tasks2[0] >= XWB_TASK2D_TARGET;
    p.task_weighted2 = c.task_schedule2 == XWB_SCHEDULE_WEIGHTED;
    for (int i = 0; i < 8; ++i) p.tasks2[i] = i < c.n_tasks2 ? c.tasks2[i] : 0;
    for (int i = 0; i < 8; ++i) p.task_acc2[i] = (i ? p.task_acc2[i - 1] : 0.0) + (i < c.n_tasks2 && p.task_weighted2 ? c.task_weights2[i] : 0.0);
    p.task_state2 = s->d_task_state2; p.task_steps2 = s->d_task_steps2;
    p.perf = s->d_perf;
    p.shadow = 0; p.swap_shadow = 0; p.list_swap = 0; p.regen_wait = 0; p.wait_slot = 3; p.sh_ep = s->d_sh_ep;
    p.sh_grid = s->d_sh_grid; p.sh_agent_xy = s->d_sh_agent; p.sh_task_state = s->d_sh_task_state; p.sh_task_state2 = s->d_sh_task_state2;
    p.sh_sent_names = s->d_sh_sent_names; p.sh_cand2d = s->d_sh_cand2d; p.sh_goal_cells = s->d_sh_goal_cells;
    p.exclusive = exclusive ? 1 : 0;
    p.group_weight[0] = c.task_group_weight; p.group_weight[1] = c.task_group_weight2;
    p.grp_order = s->d_grp_order; p.idle_list = s->d_idle_list; p.idle_count = s->d_idle_count; p.idle_count_next = nullptr;
    for (int i = 0; i < 8; ++i) p.task_acc[i] = (i ? p.task_acc[i - 1] : 0.0) + (i < c.n_tasks && p.task_weighted ? c.task_weights[i] : 0.0);
    p.policy_seed = c.policy_seed; p.env_gid0 = c.env_gid0; p.policy_step = 0; p.seed = c.seed;
    p.icon_type = s->d_icon_type; p.icon_name = s->d_icon_name;
    p.name_first = s->d_name_first; p.name_variants = s->d_name_variants;
    for (int t = 0; t < 3; ++t) { p.n_names[t] = n_names[t]; p.name_first_off[t] = off[t]; }
    p.name_first_len = (int)first.size(); p.name_variants_len = (int)variants.size();
    p.atlas = s->d_atlas;
    p.actions = nullptr; p.mask = nullptr; p.actions_out = s->d_actions;
    p.grid = s->d_grid; p.agent_xy = s->d_agent; p.task_steps = s->d_task_steps; p.task_state = s->d_task_state;
    p.num_steps = s->d_num_steps; p.episode = s->d_episode; p.success = s->d_success; p.fresh = s->d_fresh;
    p.reward = s->d_reward; p.done = s->d_done; p.obs = static_cast<uint8_t *>(s->d_obs);
    p.packed = nullptr;                     // set per call (xw_params)
    p.done_list = s->d_done_list; p.done_ep = s->d_done_ep; p.done_count = s->d_done_count; p.done_count_next = s->d_done_count + 1;
    p.err_count = s->d_err;
    if (c.visible_radius > 0) {
        if ((rc = dev_alloc(s, &s->d_ego_tab, xw_ego_tab_bytes(p)))) return rc;
        p.ego_tab = s->d_ego_tab;
        p.ego_cache = nullptr; p.ego_cache_valid = nullptr; p.ego_cache_entry = 0; p.ego_cache_words = 0;
        p.ego_cellinfo = nullptr; p.ego_miss = nullptr; p.ego_miss_count = nullptr; p.ego_xtab = nullptr; p.ego_clsimg = nullptr; p.ego_tab3 = nullptr; p.ego_cellsrc = nullptr;
        p.ego_cellsrc_list = nullptr; p.ego_miss_list = nullptr; p.ego_miss_count_list = nullptr;
        if (p.ego_fast && !(c.debug_flags & XWB_DEBUG_EGO_NO_CACHE)) {
            // rendered goal cells, [env][goal slot][view cell][heading]: ~340 KB per env at r = 3 (11 GB for a C4-sized batch;
            // the GPU has 288 GB).  Taken only if it leaves at least half of the free memory to the caller.
            size_t entry = xw_ego_cache_entry_bytes(p, s->ego_cell_edge);
            if (p.ego_span && xw_ego_square_entry_bytes(p) > entry) entry = xw_ego_square_entry_bytes(p);   // (the span path's layout)
            const size_t per_env = (size_t)p.num_goals * c.visible_radius * c.visible_radius * 4;
            const size_t bytes = (size_t)n * per_env * entry;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes < free_b / 2 && (per_env + 31) / 32 <= 64) {
                void *q = nullptr;
                if (hipMalloc(&q, bytes) == hipSuccess) {
                    s->allocs.push_back(q);
                    s->d_ego_cache = static_cast<uint8_t *>(q);
                    const size_t words = (per_env + 31) / 32;
                    if ((rc = dev_alloc(s, &s->d_ego_cache_valid, (size_t)n * words))) return rc;
                    p.ego_cache = s->d_ego_cache; p.ego_cache_valid = s->d_ego_cache_valid;
                    p.ego_cache_entry = (uint32_t)entry; p.ego_cache_words = (uint32_t)words;
                    // span path: classes of the images every env shares (everything but goals)
                    std::vector<uint8_t> cls((size_t)c.n_icons + 2, 0xff);
                    std::vector<uint16_t> cls_icon;
                    for (int i = 0; i < c.n_icons + 2; ++i)
                        if (i >= c.n_icons || c.icon_type[i] != 0) { cls[i] = (uint8_t)(cls_icon.size() < 255 ? cls_icon.size() : 0); cls_icon.push_back((uint16_t)i); }
                    if (p.ego_span && !(c.debug_flags & XWB_DEBUG_EGO_NO_SPAN) && p.n_icons < 8000 && cls_icon.size() <= 16 &&
                        (p.ego_ncls = (int)cls_icon.size(), xw_ego_square_tab_bytes(p) <= ((size_t)1 << 27))) {   // (its offsets are 23 bits of 16-byte units)
                        const int rr = c.visible_radius * c.visible_radius;
                        if ((rc = dev_alloc(s, &s->d_ego_cellinfo, (size_t)n * rr))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_cellsrc, (size_t)n * rr))) return rc;
                        p.ego_cellsrc = s->d_ego_cellsrc;
                        if ((rc = dev_alloc(s, &s->d_ego_cellsrc_list, (size_t)n * rr))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_miss_list, (size_t)n * (p.num_goals < rr ? p.num_goals : rr)))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_miss_count_list, 4))) return rc;
                        p.ego_cellsrc_list = s->d_ego_cellsrc_list; p.ego_miss_list = s->d_ego_miss_list; p.ego_miss_count_list = s->d_ego_miss_count_list;
                        if ((rc = dev_alloc(s, &s->d_ego_miss, (size_t)n * (p.num_goals < rr ? p.num_goals : rr)))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_miss_count, 4))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_cls, cls.size()))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_cls_icon, cls_icon.size()))) return rc;
                        HIP_TRY(hipMemcpy(s->d_ego_cls, cls.data(), cls.size(), hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(s->d_ego_cls_icon, cls_icon.data(), cls_icon.size() * 2, hipMemcpyHostToDevice));
                        p.ego_cls = s->d_ego_cls; p.ego_cls_icon = s->d_ego_cls_icon; p.ego_ncls = (int)cls_icon.size();
                        if ((rc = dev_alloc(s, &s->d_ego_tab3, xw_ego_square_tab_bytes(p) + 16))) return rc;
                        p.ego_tab3 = s->d_ego_tab3;
                        if ((rc = dev_alloc(s, &s->d_ego_xtab, xw_ego_xtab_bytes(p) / sizeof(uint32_t)))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_clsimg, (size_t)4 * 16))) return rc;
                        p.ego_xtab = s->d_ego_xtab; p.ego_clsimg = s->d_ego_clsimg;
                        p.ego_cellinfo = s->d_ego_cellinfo; p.ego_miss = s->d_ego_miss; p.ego_miss_count = s->d_ego_miss_count;
                    }
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        HIP_TRY(launch_xw_ego_build_tab(p, nullptr));
        if (p.ego_cellinfo) HIP_TRY(launch_xw_ego_build_squares(p, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (p.ego_cellinfo) {
            // which squares of the table are one flat colour (empty cells: 255; outside the map / shadow: 0): found by looking
            // at the table itself, so the shortcut the gather takes for them (XwParams::ego_flat) cannot change a byte
            const int r = c.visible_radius, U = 84 / r, UP = 4 * ((U / 4 + 3) & ~3), RR = r * r, nc = p.ego_ncls;
            const size_t CBP = (size_t)U * UP, PBP = (size_t)RR * CBP, keys = (size_t)4 * nc * nc * nc;
            std::vector<uint8_t> tab(xw_ego_square_tab_bytes(p)), flat(keys * RR, 0);
            HIP_TRY(hipMemcpy(tab.data(), s->d_ego_tab3, tab.size(), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < keys; ++k)
                for (int f = 0; f < RR; ++f) {
                    const uint8_t v0 = tab[k * p.channels * PBP + (size_t)f * CBP];
                    bool same = v0 == 0 || v0 == 255;
                    for (int chn = 0; chn < p.channels && same; ++chn)
                        for (int y = 0; y < U && same; ++y) {
                            const uint8_t *row = tab.data() + (k * p.channels + chn) * PBP + (size_t)f * CBP + (size_t)y * UP;
                            for (int x = 0; x < U; ++x) if (row[x] != v0) { same = false; break; }
                        }
                    flat[k * RR + f] = same && !(c.debug_flags & XWB_DEBUG_EGO_NO_FLAT) ? (v0 == 255 ? 1 : 2) : 0;
                }
            if ((rc = dev_alloc(s, &s->d_ego_flat, flat.size()))) return rc;
            HIP_TRY(hipMemcpy(s->d_ego_flat, flat.data(), flat.size(), hipMemcpyHostToDevice));
            if ((rc = dev_alloc(s, &s->d_ego_constline, 256))) return rc;
            HIP_TRY(hipMemset(s->d_ego_constline, 0xff, 128));
            p.ego_flat = s->d_ego_flat; p.ego_constline = s->d_ego_constline;
        }
    }
    return XWB_OK;
}

}  // namespace

// (xwb_comm.hip reports its errors through the same per-thread message)
extern "C" __attribute__((visibility("hidden"))) int xwb_internal_fail(int code, const char *msg) { return fail(code, msg); }

// =============================================================== C ABI =====
extern "C" {

const char *xwb_last_error(void) { return g_err.c_str(); }
const char *xwb_version(void) { return "xwb 0.1 (gfx950)"; }

int xwb_default_config(int32_t game, xwb_config *c) {
    if (!c) return fail(XWB_ERR_ARG, "cfg is NULL");
    memset(c, 0, sizeof *c);
    c->abi_version = XWB_ABI_VERSION;
    c->game = game;
    c->num_envs = 1;
    c->seed = 0xC0FFEEu;
    c->policy_seed = 0x5EEDu;
    c->context = 1;                 // simulator.cpp:21
    c->max_steps = 0;               // simulator.cpp:22
    c->array_size = 6;              // simple_game_simulator.cpp:19
    c->track_type = 0;              // simple_race_simulator.cpp:17
    c->track_width = 20.0f; c->track_length = 100.0f; c->track_radius = 30.0f;   // :18-20
    c->reward_scale = 1.0;          // :26
    c->map_kind = XWB_MAP_NAV; c->max_dim = 8; c->dim = 8; c->num_goals = 4; c->num_blocks = 16;  // XWorldNav.py:8-13,27-39
    c->max_steps_factor = 10;       // simulator.cpp:23
    c->task_mode = XWB_TASKMODE_LANG_ACQ;   // xworld_simulator.cpp:33-37
    c->color = 0;                   // simulator.cpp:25
    if (game < 0 || game > 2) return fail(XWB_ERR_ARG, "unknown game");
    return XWB_OK;
}

int xwb_create(const xwb_config *cfg, xwb_sim **out) {
    if (!cfg || !out) return fail(XWB_ERR_ARG, "NULL argument");
    if (cfg->abi_version != XWB_ABI_VERSION) return fail(XWB_ERR_ARG, "abi_version mismatch");
    if (cfg->num_envs < 1) return fail(XWB_ERR_ARG, "num_envs must be >= 1");
    if (cfg->context < 1) return fail(XWB_ERR_ARG, "context must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(XWB_ERR_HIP, "no HIP device: libxwb.so has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(XWB_ERR_ARG, "bad device ordinal");
    DeviceGuard _device_guard(cfg->device);           // the caller's current device is restored on return
    if (cfg->debug_flags & ~63) return fail(XWB_ERR_ARG, "unknown debug_flags bit");
    if ((cfg->debug_ego_per != 0 && cfg->debug_ego_per != 2 && cfg->debug_ego_per != 4 && cfg->debug_ego_per != 8) || cfg->debug_ego_pad < 0 ||
        cfg->debug_ego_pad > 65536 || cfg->debug_render_shape < 0 || cfg->debug_render_shape > 2)
        return fail(XWB_ERR_ARG, "debug_ego_per must be 0 | 2 | 4 | 8, debug_ego_pad 0 .. 65536, debug_render_shape 0 .. 2");
    xwb_sim *s = new xwb_sim();
    s->cfg = *cfg;
    {   // the process-wide override (XWB_DEBUG), see xwb.h
        const DebugEnv &d = debug_env();
        s->cfg.debug_flags |= d.flags;
        // (the same range checks as the configuration's own fields: an entry outside them is ignored, with a warning)
        if (d.ego_per == 2 || d.ego_per == 4 || d.ego_per == 8) s->cfg.debug_ego_per = d.ego_per;
        else if (d.ego_per) fprintf(stderr, "libxwb: XWB_DEBUG: ego_per=%d ignored (2 | 4 | 8)\n", d.ego_per);
        if (d.ego_pad > 0 && d.ego_pad <= 65536) s->cfg.debug_ego_pad = d.ego_pad;
        else if (d.ego_pad) fprintf(stderr, "libxwb: XWB_DEBUG: ego_pad=%d ignored (0 .. 65535)\n", d.ego_pad - 1);
        if (d.render_shape) s->cfg.debug_render_shape = d.render_shape;
    }
    s->device = cfg->device;
    s->n = cfg->num_envs;
    const int n = s->n;
    int rc = XWB_OK;
    auto bail = [&](int code) { xwb_destroy(s); return code; };
    switch (cfg->game) {
        case XWB_SIMPLE_GAME:
            if (cfg->array_size < 1) return bail(fail(XWB_ERR_ARG, "array_size must be >= 1"));
            s->out_h = 1; s->out_w = cfg->array_size; s->out_c = 1;       // simple_game_simulator.cpp:118-124
            s->obs_bytes_per_env = (size_t)cfg->context * cfg->array_size;
            s->num_actions = 2;
            break;
        case XWB_SIMPLE_RACE:
            if (cfg->track_type != 0 && cfg->track_type != 1) return bail(fail(XWB_ERR_ARG, "track_type must be 0 or 1"));
            s->out_h = 1; s->out_w = 4; s->out_c = 1;                     // simple_race_simulator.cpp:492-501
            s->obs_bytes_per_env = (size_t)cfg->context * 4 * sizeof(float);
            race_setup(*cfg, s->race);
            s->num_actions = s->race.n_legal;
            break;
        case XWB_XWORLD2D:
            s->out_h = cfg->max_dim * 12; s->out_w = cfg->max_dim * 12; s->out_c = cfg->color ? 3 : 1;   // xworld_simulator.cpp:53-61,106-112
            if (cfg->obs_format != XWB_OBS_U8 && cfg->obs_format != XWB_OBS_F32) return bail(fail(XWB_ERR_ARG, "xworld: unknown obs_format"));
            s->num_actions = 4;                                           // xitem.cpp:82-83
            if (cfg->visible_radius < 0) return bail(fail(XWB_ERR_ARG, "xworld: visible_radius must be >= 0"));
            if (cfg->visible_radius > 0) {
                // xworld_simulator.cpp:62-68: clamp to the map, frame edge r * (84 / r); xmap.cpp:277: r must be odd
                if (s->cfg.visible_radius > cfg->max_dim) s->cfg.visible_radius = cfg->max_dim;
                const int r = s->cfg.visible_radius;
                if (r % 2 != 1) return bail(fail(XWB_ERR_ARG, "xworld: visible_radius must be an odd int (xmap.cpp:277)"));
                if (cfg->map_kind != XWB_MAP_NAV)
                    return bail(fail(XWB_ERR_ARG, "xworld: visible_radius > 0 needs a maze map (XWorldNav): without maze "
                                                  "generation the reference's set_property rejects the agent's default yaw "
                                                  "(xworld_env.py:208-210, py_util.py:27-29)"));
                s->out_h = s->out_w = r * (84 / r);
                s->num_actions = 6;                                       // xitem.cpp:84-86
            }
            s->obs_bytes_per_env = (size_t)cfg->context * s->out_c * s->out_h * s->out_w * (cfg->obs_format == XWB_OBS_F32 ? 4 : 1);
            break;
        default:
            return bail(fail(XWB_ERR_ARG, "Unrecognized game type"));     // simulator_interface.cpp:82
    }
    if (cfg->rng_mode != XWB_RNG_PHILOX && cfg->rng_mode != XWB_RNG_MINSTD) return bail(fail(XWB_ERR_ARG, "unknown rng_mode"));
    if (cfg->queue_sync < XWB_QUEUE_SYNC_AUTO || cfg->queue_sync > XWB_QUEUE_SYNC_EPOCHS) return bail(fail(XWB_ERR_ARG, "unknown queue_sync"));
    if (cfg->rng_mode == XWB_RNG_MINSTD) {
        // the reference seeds an engine per thread only when FLAGS_simulator_seed != 0 (simulator_util.cpp:44-52); with 0
        // its engines start from hash(thread id), which nobody can replay
        if (cfg->simulator_seed == 0) return bail(fail(XWB_ERR_ARG, "rng_mode minstd needs simulator_seed != 0"));
        if (cfg->thread_base < 0) return bail(fail(XWB_ERR_ARG, "thread_base must be >= 0"));
        std::vector<uint32_t> st((size_t)n);
        for (int e = 0; e < n; ++e)
            st[(size_t)e] = xwb_minstd_seed_thread(cfg->simulator_seed, cfg->thread_base + (int32_t)(cfg->env_gid0 + (uint32_t)e) + 1);
        if ((rc = dev_alloc(s, &s->d_minstd, n))) return bail(rc);
        HIP_TRY(hipMemcpy(s->d_minstd, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    }
    if ((rc = dev_alloc(s, &s->d_actions, n, 0xff))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_actions_in, n, 0xff))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_mask, n))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_num_steps, n))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_err, 1))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_reset_partial, (size_t)(n + 255) / 256))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_episode, n, 0xff))) return bail(rc);      // first reset -> episode 0
    if ((rc = dev_alloc(s, &s->d_reward, n))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_done, n))) return bail(rc);
    if ((rc = dev_alloc(s, &s->d_success, n, 1))) return bail(rc);         // last_action_success_(true), simulator.cpp:33-34
    {
        uint8_t *obs = nullptr;
        if ((rc = dev_alloc(s, &obs, (size_t)n * s->obs_bytes_per_env))) return bail(rc);
        s->d_obs = s->d_obs_owned = obs;
    }
    if (cfg->game == XWB_SIMPLE_GAME) {
        if ((rc = dev_alloc(s, &s->d_pos, n))) return bail(rc);
        if ((rc = dev_alloc(s, &s->d_flags, n))) return bail(rc);
    } else if (cfg->game == XWB_SIMPLE_RACE) {
        if ((rc = dev_alloc(s, &s->d_x, n))) return bail(rc);
        if ((rc = dev_alloc(s, &s->d_y, n))) return bail(rc);
        if ((rc = dev_alloc(s, &s->d_angle, n))) return bail(rc);
    } else {
        if ((rc = xw_setup(s))) return bail(rc);
    }
    // the reference constructors leave a reset game behind (SimpleGame ctor cpp:82-85, SimpleRaceGame
    // ctor cpp:457, XWorld ctor xworld.cpp:106); screens_ stays empty until reset_game -> init_screen.
    s->cfg.icons64 = nullptr; s->cfg.icon_type = nullptr; s->cfg.icon_name = nullptr;   // not owned
    s->cfg.icon_colored = nullptr;
    rc = xwb_reset(s, nullptr);
    if (rc) return bail(rc);
    HIP_TRY(hipDeviceSynchronize());
    if (s->d_sync) {
        // the default stream is probed now (other streams: xwb_queue_sync_mode); the probe also re-selects the internal stream
        // when it shares the caller's hardware queue (side_beside, xwb_verbs.hip).  Forced modes and tools skip the probe in
        // use_epochs: the internal stream is still chosen, unless a tool serialises kernels (nothing would pass).
        int why = 0, reason = 0;
        const bool tool = queue_sync_env(&why) == 0 && why == XWB_SYNC_REASON_TOOL;
        (void)use_epochs(s, nullptr, true);
        const int verdict = s->sync_reason;
        const bool probed = verdict == XWB_SYNC_REASON_PROBE_OK || verdict == XWB_SYNC_REASON_PROBE_FAILED || verdict == XWB_SYNC_REASON_PROBE_ERROR;
        if (!probed && !tool) (void)side_beside(s, nullptr, &reason);
        s->sync_reason = verdict;
    }
    *out = s;
    return XWB_OK;
}

int xwb_destroy(xwb_sim *s) {
    if (!s) return XWB_OK;
    XWB_ON_DEVICE(s);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->h_poison) (void)hipHostFree(s->h_poison);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->ev_step) (void)hipEventDestroy(s->ev_step);
    if (s->ev_reset) (void)hipEventDestroy(s->ev_reset);
    if (s->ev_term) (void)hipEventDestroy(s->ev_term);
    if (s->ev_cells) (void)hipEventDestroy(s->ev_cells);
    if (s->ev_results) (void)hipEventDestroy(s->ev_results);
    for (KernelTimer *t : {&s->t_render, &s->t_step, &s->t_reset, &s->t_list})
        for (auto &ep : t->pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    delete s;
    return XWB_OK;
}

}  // extern "C"
