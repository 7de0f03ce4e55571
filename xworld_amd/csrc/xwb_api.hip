// xwb_api.hip -- host side of libxwb.so: the C ABI of include/xwb.h.
//
// A xwb_sim is the batched counterpart of simulator::SimulatorInterface
// (simulator_interface.h:40-89): it owns the SoA state of num_envs environments in
// HBM and sequences the kernels in the reference's call order
// (simulator_interface.cpp:95-143).  No CPU fallback exists: without a usable
// gfx950 device xwb_create fails.
#include "../../include/xwb.h"
#include "xwb_common.h"
#include "xwb_language.h"
#include "../../include/xwb_trig.h"
#include "../../include/xwb_minstd.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

using namespace xwb;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(XWB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

struct EventPair { hipEvent_t a, b; };

// Every entry point that touches the device runs with the batch's device current and restores the caller's
// device on return: two batches on different GPUs of one process, or a caller whose current device is not the
// batch's, launch on the right device.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define XWB_ON_DEVICE(s) DeviceGuard _device_guard((s)->device)

struct KernelTimer {
    std::vector<EventPair> pool;
    size_t used = 0;
};

}  // namespace

struct xwb_sim {
    xwb_config cfg;
    int device = 0;
    int n = 0;
    size_t obs_bytes_per_env = 0;
    int out_h = 0, out_w = 0, out_c = 0;
    int num_actions = 0;
    uint32_t policy_step = 0;
    bool list_valid = false;
    // xwb_xw_pack_grids: what the last frame-drawing verb read (xw_pack_grids_kernel's src) and how many such verbs ran since
    // the last pack (a context ring can only be replayed elsewhere one draw at a time)
    int frame_src = 0, draws_since_pack = 0;
    bool autoreset_done = false;           // the last step call already reset the envs whose codes are still set
    int count_sel = 0;
    bool profiling = false;
    KernelTimer t_render, t_step, t_reset, t_list;   // t_list: the list render (first frames of the envs a reset started)
    int last_path = XWB_PATH_NONE;           // xwb_step_path: which kernel sequence the last step call ran
    hipStream_t side = nullptr;            // reset of finished envs runs here, beside render_all
    uint32_t *d_minstd = nullptr;          // XWB_RNG_MINSTD: one engine state per env
    uint32_t *d_sync = nullptr;            // device-side epochs of the step / reset kernels (XwParams::sync)
    uint32_t epoch_step = 0, epoch_reset = 0;
    // queue hand-off mode (include/xwb.h xwb_queue_sync_mode): decided per caller stream by a one-time probe
    struct StreamProbe { hipStream_t st; bool ok; int reason; };
    std::vector<StreamProbe> probes;
    int sync_reason = XWB_SYNC_REASON_NOT_USED;
    bool step_epochs = false;              // the last step call's hand-overs were epochs (a following reset_done follows suit:
                                           // its waiters wait for what that step's kernels publish)
    uint32_t probe_token = 0;
    uint32_t *h_poison = nullptr;          // pinned host word: a watchdog expired (XwParams::poison_host points at it)
    bool poisoned = false;
    hipEvent_t ev_step = nullptr, ev_reset = nullptr, ev_term = nullptr, ev_cells = nullptr;
    bool span_epochs = false;              // ... and handed over through epochs (d_sync[5..7]) rather than those events
    bool span_step = false;                // the last step drew its frames on the egocentric span path (ev_cells / ev_step / ev_term are its)
    // common device buffers
    int32_t *d_actions_in = nullptr;       // staging for xwb_step_host
    uint8_t *d_mask = nullptr;             // staging for xwb_reset_env
    int32_t *d_actions = nullptr, *d_num_steps = nullptr, *d_err = nullptr, *d_reset_partial = nullptr;   // (SgParams::reset_partial)
    uint32_t *d_episode = nullptr;
    float *d_reward = nullptr;
    uint8_t *d_done = nullptr, *d_success = nullptr;
    void *d_obs = nullptr, *d_obs_owned = nullptr;
    float2 *d_packed = nullptr;            // caller-owned (xwb_bind_results): slot 0 of the ring
    int64_t packed_slots = 1, packed_pos = 0;   // xwb_bind_results_ring: step call k writes slot k % slots
    // simple_game
    int32_t *d_pos = nullptr;
    uint8_t *d_flags = nullptr;
    // simple_race
    float *d_x = nullptr, *d_y = nullptr, *d_angle = nullptr;
    RaceParams race{};
    // xworld
    uint16_t *d_grid = nullptr;
    int32_t *d_task_steps2 = nullptr, *d_task_state2 = nullptr;
    uint8_t *d_grp_order = nullptr;        // exclusive group scheduling (XwParams::grp_order)
    int32_t *d_idle_list = nullptr, *d_idle_count = nullptr;
    unsigned long long *d_perf = nullptr;  // XwParams::perf
    // pre-generated next episodes (XwParams::shadow / swap_shadow): xwb_step_autoreset's fast path
    bool pregen = false, shadow_ok = false, regen_pending = false, regen_by_epoch = false;
    bool step_lazy = false;                // the last plain step kept no terminal snapshot: its reset_done installs shadows
    int shadow_breaks = 0;                 // times another verb made the shadows stale (the lazy default path gives up after a few)
    uint32_t epoch_regen = 0;
    uint32_t *d_sh_ep = nullptr;
    uint8_t *d_sh_goal_cells = nullptr;
    uint16_t *d_sh_grid = nullptr;
    int32_t *d_sh_agent = nullptr, *d_sh_task_state = nullptr, *d_sh_task_state2 = nullptr;
    uint32_t *d_sh_sent_names = nullptr, *d_sh_cand2d = nullptr;
    int32_t *d_agent = nullptr, *d_task_steps = nullptr, *d_task_state = nullptr, *d_done_list = nullptr,
            *d_done_count = nullptr;
    uint8_t *d_fresh = nullptr, *d_icon_type = nullptr, *d_icon_colored = nullptr, *d_goal_cells = nullptr;
    uint32_t *d_cand2d = nullptr, *d_sent_names = nullptr;
    uint8_t *d_cur_level = nullptr, *d_cur_usage = nullptr;
    int32_t *d_cur_counter = nullptr;
    uint16_t *d_term_grid = nullptr;
    uint8_t *d_term_flag = nullptr;
    uint8_t *d_agent_dir = nullptr, *d_atlas64 = nullptr;
    uint32_t *d_goal_img = nullptr, *d_agent_rot = nullptr;
    EgoTap *d_ego_taps = nullptr;
    uint8_t *d_ego_tab = nullptr;
    int ego_cell_edge = 1;
    uint8_t *d_ego_cache = nullptr;        // lazily filled cache of rendered goal cells (XwParams::ego_cache)
    uint32_t *d_ego_cache_valid = nullptr;
    uint32_t *d_ego_cellsrc = nullptr, *d_ego_cellsrc_list = nullptr;
    uint2 *d_ego_miss_list = nullptr;
    int32_t *d_ego_miss_count_list = nullptr;
    uint32_t *d_ego_cellinfo = nullptr;    // span path of the egocentric render (XwParams::ego_span)
    uint2 *d_ego_miss = nullptr;
    int32_t *d_ego_miss_count = nullptr;
    uint8_t *d_ego_border = nullptr, *d_ego_cls = nullptr, *d_ego_tab3 = nullptr, *d_ego_flat = nullptr, *d_ego_constline = nullptr;
    uint16_t *d_ego_cls_icon = nullptr;
    double *d_goal_warp = nullptr;
    int16_t *d_icon_name = nullptr, *d_name_first = nullptr, *d_name_variants = nullptr;
    uint32_t *d_atlas = nullptr;
    std::vector<uint8_t> tile_table;   // host copy, n_icons x c x 12 x 12
    std::vector<int32_t> icon_type_h, icon_name_h, icon_colored_h;
    // xwb_set_names: the strings behind the name ids (the teacher's sentences are built from them)
    std::vector<std::string> goal_names, icon_names, icon_colors;
    bool have_names = false;
    XwParams xw{};
    std::vector<void *> allocs;
};

namespace {

template <typename T>
int dev_alloc(xwb_sim *s, T **p, size_t count, int fill = 0) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemset(q, fill, bytes));
    s->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return XWB_OK;
}

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// The step loop's two queues hand over through epochs in device memory (XwParams::sync) instead of event / barrier packets
// (3-6 us of idle GPU each).  A waiter polls until a kernel of the other queue has run.  Three things keep that safe
// (include/xwb.h, xwb_queue_sync_mode): publishers are enqueued before their waiters everywhere below; epochs are only used on
// a caller stream that passed a concurrency probe against s->side (epoch_probe); a watchdog poisons the batch.
// Overrides of the AUTO mode: XWB_QUEUE_SYNC=events|epochs, and tools that serialise kernel execution (rocprofv3's counter
// collection: ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS; AMD_SERIALIZE_KERNEL; HIP_LAUNCH_BLOCKING) -> events.
// returns -1: no override, 0: events, 1: epochs; *reason = XWB_SYNC_REASON_ENV | _TOOL
int queue_sync_env(int *reason) {
    static int mode = -2, why = 0;
    if (mode == -2) {
        auto on = [](const char *name) { const char *v = getenv(name); return v && *v && strcmp(v, "0") != 0; };
        mode = -1;
        if (on("ROCPROF_COUNTER_COLLECTION") || getenv("ROCPROF_COUNTERS") || on("AMD_SERIALIZE_KERNEL") || on("HIP_LAUNCH_BLOCKING") ||
            on("CUDA_LAUNCH_BLOCKING")) { mode = 0; why = XWB_SYNC_REASON_TOOL; }
        if (const char *v = getenv("XWB_QUEUE_SYNC")) {
            if (strcmp(v, "events") == 0) { mode = 0; why = XWB_SYNC_REASON_ENV; }
            else if (strcmp(v, "epochs") == 0) { mode = 1; why = XWB_SYNC_REASON_ENV; }
        }
    }
    *reason = why;
    return mode;
}

// One-time probe of (caller stream, s->side): do kernels of the two really run concurrently?  A waiter with a 2 ms watchdog
// is enqueued FIRST on one stream, its publisher on the other, in both directions; on streams that share a hardware queue
// (or under a tool that serialises kernels) the waiter runs alone, expires and raises the probe's own flag (d_sync[2], not
// the batch's poison word).  Both streams are drained before and after, so work of the caller that is still queued cannot
// make the probe fail (or be delayed by it) -- the cost is one synchronisation the first time a stream is seen.
bool epoch_probe(xwb_sim *s, hipStream_t st, int *reason) {
    auto bad = [&](int why) { (void)hipGetLastError(); *reason = why; return false; };
    if (hipStreamSynchronize(st) != hipSuccess || hipStreamSynchronize(s->side) != hipSuccess) return bad(XWB_SYNC_REASON_PROBE_ERROR);
    if (hipMemsetAsync(s->d_sync + 2, 0, sizeof(uint32_t), s->side) != hipSuccess || hipStreamSynchronize(s->side) != hipSuccess)
        return bad(XWB_SYNC_REASON_PROBE_ERROR);
    for (int dir = 0; dir < 2; ++dir) {
        hipStream_t waiter = dir ? st : s->side, publisher = dir ? s->side : st;
        if (++s->probe_token == 0) s->probe_token = 1;
        if (launch_xw_wait(s->d_sync + 0, s->probe_token, s->d_sync + 2, nullptr, waiter, 200000ull) != hipSuccess)   // 2 ms
            return bad(XWB_SYNC_REASON_PROBE_ERROR);
        if (launch_xw_signal(s->d_sync + 0, s->probe_token, publisher) != hipSuccess) return bad(XWB_SYNC_REASON_PROBE_ERROR);
        if (hipStreamSynchronize(waiter) != hipSuccess || hipStreamSynchronize(publisher) != hipSuccess) return bad(XWB_SYNC_REASON_PROBE_ERROR);
    }
    uint32_t expired = 1;
    if (hipMemcpy(&expired, s->d_sync + 2, sizeof expired, hipMemcpyDeviceToHost) != hipSuccess) return bad(XWB_SYNC_REASON_PROBE_ERROR);
    if (expired) {
        (void)hipMemset(s->d_sync + 2, 0, sizeof(uint32_t));
        *reason = XWB_SYNC_REASON_PROBE_FAILED;
        return false;
    }
    *reason = XWB_SYNC_REASON_PROBE_OK;
    return true;
}

// may calls on stream `st` hand over through epochs?  (xworld batches only: the other games have no internal stream)
// may_probe: only xwb_create (the default stream) and xwb_queue_sync_mode (any stream, an explicit call) run the probe -- it
// synchronises both streams and the host; the step verbs never do: a stream nobody probed hands over through events.
bool use_epochs(xwb_sim *s, hipStream_t st, bool may_probe) {
    if (!s->d_sync || !s->side) { s->sync_reason = XWB_SYNC_REASON_NOT_USED; return false; }
    if (s->cfg.queue_sync == XWB_QUEUE_SYNC_EVENTS) { s->sync_reason = XWB_SYNC_REASON_CONFIG; return false; }
    if (s->cfg.queue_sync == XWB_QUEUE_SYNC_EPOCHS) { s->sync_reason = XWB_SYNC_REASON_CONFIG; return true; }
    int why = 0;
    const int env = queue_sync_env(&why);
    if (env >= 0) { s->sync_reason = why; return env == 1; }
    for (auto &pr : s->probes) if (pr.st == st) { s->sync_reason = pr.reason; return pr.ok; }
    if (!may_probe) { s->sync_reason = XWB_SYNC_REASON_NOT_PROBED; return false; }
    {   // a stream under graph capture cannot be synchronised (the probe would invalidate the capture): events, nothing cached
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap != hipStreamCaptureStatusNone) { s->sync_reason = XWB_SYNC_REASON_NOT_PROBED; return false; }
    }
    int reason = 0;
    const bool ok = epoch_probe(s, st, &reason);
    if (s->probes.size() >= 16) s->probes.erase(s->probes.begin());
    s->probes.push_back(xwb_sim::StreamProbe{st, ok, reason});
    s->sync_reason = reason;
    return ok;
}

const char *POISON_MSG = "a device-side queue hand-off was not released within its watchdog (kernels of the batch's two queues did "
                         "not run concurrently, or the device is wedged): the batch is poisoned -- results since the last "
                         "successful xwb_check_errors are void, destroy it (XWB_QUEUE_SYNC=events / xwb_config.queue_sync avoid epochs)";
bool is_poisoned(xwb_sim *s) {
    if (!s->poisoned && s->h_poison && *(volatile uint32_t *)s->h_poison) s->poisoned = true;
    return s->poisoned;
}
#define XWB_LIVE(s) do { if (is_poisoned(s)) return fail(XWB_ERR_STATE, POISON_MSG); } while (0)

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
        if ((rc = dev_alloc(s, &s->d_idle_count, 2))) return rc;
    }
    if ((rc = dev_alloc(s, &s->d_perf, 40))) return rc;
    // Pre-generated next episodes: possible where an env's next episode is a pure function of (seed, global id, episode + 1)
    // and the render reads nothing but the grid -- full observation, no curriculum (the level depends on the results so far),
    // no per-env reference engine (its state depends on the draws so far), no exclusive group order carried across resets.
    // (float32 frames stay on the classic paths: their plain whole-batch render variant measured 5-8 % slower than the
    // variants the classic paths use -- 416 vs 385 / 394 us on the C4-sized batch)
    s->pregen = c.visible_radius == 0 && !curriculum_cfg(c) && c.rng_mode != XWB_RNG_MINSTD && !(exclusive && c.n_tasks2 > 0) &&
                c.obs_format == XWB_OBS_U8 && !getenv("XWB_NO_PREGEN");
    if (s->pregen) {
        if ((rc = dev_alloc(s, &s->d_sh_ep, n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_grid, (size_t)2 * n * cells))) return rc;           // two slots per env
        if ((rc = dev_alloc(s, &s->d_sh_agent, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_task_state, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_task_state2, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_sent_names, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_cand2d, (size_t)2 * n))) return rc;
        if ((rc = dev_alloc(s, &s->d_sh_goal_cells, (size_t)2 * n * XW_MAX_GOALS))) return rc;
    }
    if ((rc = dev_alloc(s, &s->d_done_list, n))) return rc;
    if ((rc = dev_alloc(s, &s->d_done_count, 2))) return rc;
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
    p.done_list = s->d_done_list; p.done_count = s->d_done_count; p.done_count_next = s->d_done_count + 1;
    p.err_count = s->d_err;
    if (c.visible_radius > 0) {
        if ((rc = dev_alloc(s, &s->d_ego_tab, xw_ego_tab_bytes(p)))) return rc;
        p.ego_tab = s->d_ego_tab;
        p.ego_cache = nullptr; p.ego_cache_valid = nullptr; p.ego_cache_entry = 0; p.ego_cache_words = 0;
        p.ego_cellinfo = nullptr; p.ego_miss = nullptr; p.ego_miss_count = nullptr; p.ego_border = nullptr; p.ego_tab3 = nullptr; p.ego_cellsrc = nullptr;
        p.ego_cellsrc_list = nullptr; p.ego_miss_list = nullptr; p.ego_miss_count_list = nullptr;
        if (p.ego_fast && !getenv("XWB_EGO_NO_CACHE")) {
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
                    if (p.ego_span && !getenv("XWB_EGO_NO_SPAN") && p.n_icons < 8000 && cls_icon.size() <= 16 &&
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
                        if ((rc = dev_alloc(s, &s->d_ego_border, (size_t)n * 2 * (c.visible_radius - 1) * p.channels * p.out_dim + 16))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_cls, cls.size()))) return rc;
                        if ((rc = dev_alloc(s, &s->d_ego_cls_icon, cls_icon.size()))) return rc;
                        HIP_TRY(hipMemcpy(s->d_ego_cls, cls.data(), cls.size(), hipMemcpyHostToDevice));
                        HIP_TRY(hipMemcpy(s->d_ego_cls_icon, cls_icon.data(), cls_icon.size() * 2, hipMemcpyHostToDevice));
                        p.ego_cls = s->d_ego_cls; p.ego_cls_icon = s->d_ego_cls_icon; p.ego_ncls = (int)cls_icon.size();
                        if ((rc = dev_alloc(s, &s->d_ego_tab3, xw_ego_square_tab_bytes(p) + 16))) return rc;
                        p.ego_tab3 = s->d_ego_tab3;
                        p.ego_border = s->d_ego_border;
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
                    flat[k * RR + f] = same && !getenv("XWB_EGO_NO_FLAT") ? (v0 == 255 ? 1 : 2) : 0;      // (A/B hook)
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

void timer_begin(xwb_sim *s, KernelTimer &t, hipStream_t st) {
    if (!s->profiling) return;
    if (t.used == t.pool.size()) {
        EventPair ep;
        if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return;
        t.pool.push_back(ep);
    }
    (void)hipEventRecord(t.pool[t.used].a, st);
}

void timer_end(xwb_sim *s, KernelTimer &t, hipStream_t st) {
    if (!s->profiling || t.used >= t.pool.size()) return;
    (void)hipEventRecord(t.pool[t.used].b, st);
    t.used++;
}

// the results slot of the step call being queued (xwb_bind_results_ring)
float2 *packed_slot(xwb_sim *s) {
    return s->d_packed ? s->d_packed + (size_t)(s->packed_pos % s->packed_slots) * (size_t)s->n : nullptr;
}

SgParams sg_params(xwb_sim *s) {
    SgParams p{};
    const xwb_config &c = s->cfg;
    p.n = s->n; p.array_size = c.array_size; p.context = c.context; p.max_steps = c.max_steps;
    p.act_rep = 1; p.mode = MODE_STEP; p.auto_reset = 0;
    p.policy_seed = c.policy_seed; p.env_gid0 = c.env_gid0; p.policy_step = s->policy_step;
    p.actions = nullptr; p.mask = nullptr; p.actions_out = s->d_actions;
    p.pos = s->d_pos; p.flags = s->d_flags; p.num_steps = s->d_num_steps; p.episode = s->d_episode;
    p.reward = s->d_reward; p.done = s->d_done; p.obs = static_cast<uint8_t *>(s->d_obs);
    p.packed = packed_slot(s);
    p.n_steps = 1;
    p.err_count = s->d_err; p.reset_partial = nullptr;
    return p;
}

RaceParams race_params(xwb_sim *s) {
    RaceParams p = s->race;
    const xwb_config &c = s->cfg;
    p.n = s->n; p.context = c.context; p.max_steps = c.max_steps; p.act_rep = 1; p.mode = MODE_STEP;
    p.auto_reset = 0;
    p.policy_seed = c.policy_seed; p.env_gid0 = c.env_gid0; p.policy_step = s->policy_step; p.seed = c.seed;
    p.actions = nullptr; p.mask = nullptr; p.actions_out = s->d_actions;
    p.x = s->d_x; p.y = s->d_y; p.angle = s->d_angle; p.num_steps = s->d_num_steps; p.episode = s->d_episode;
    p.reward = s->d_reward; p.done = s->d_done; p.obs = static_cast<float *>(s->d_obs);
    p.packed = packed_slot(s);
    p.n_steps = 1;
    p.err_count = s->d_err; p.reset_partial = nullptr;
    p.minstd = s->d_minstd;
    return p;
}

// a launch that may reset envs writes its per-workgroup counts (xwb_done_count reports the last such launch)
template <typename P>
void take_reset_counter(xwb_sim *s, P &p) { p.reset_partial = s->d_reset_partial; }

// reset for the simple games: one launch, mode selects the envs
int simple_reset(xwb_sim *s, int mode, const uint8_t *mask, hipStream_t st) {
    timer_begin(s, s->t_reset, st);
    if (s->cfg.game == XWB_SIMPLE_GAME) {
        SgParams p = sg_params(s);
        p.mode = mode; p.mask = mask;
        take_reset_counter(s, p);
        HIP_TRY(launch_simple_game(p, st));
    } else {
        RaceParams p = race_params(s);
        p.mode = mode; p.mask = mask;
        take_reset_counter(s, p);
        HIP_TRY(launch_simple_race(p, st));
    }
    timer_end(s, s->t_reset, st);
    return XWB_OK;
}

XwParams xw_params(xwb_sim *s) {
    XwParams p = s->xw;
    p.obs = static_cast<uint8_t *>(s->d_obs);
    p.packed = packed_slot(s);
    p.policy_step = s->policy_step;
    p.list_flag = 2;
    p.done_count = s->d_done_count + s->count_sel;
    p.done_count_next = s->d_done_count + (1 - s->count_sel);
    if (s->d_idle_count) { p.idle_count = s->d_idle_count + s->count_sel; p.idle_count_next = s->d_idle_count + (1 - s->count_sel); }
    return p;
}

// the reset kernel's parameters for a pre-generation pass: episode[e] + 1 of the listed envs into the shadow arrays
XwParams shadow_params(xwb_sim *s) {
    XwParams q = xw_params(s);
    q.shadow = 1; q.auto_reset = 1; q.sig_epoch = 0; q.wait_epoch = 0; q.packed = nullptr;
    q.grid = s->d_sh_grid; q.agent_xy = s->d_sh_agent; q.task_state = s->d_sh_task_state; q.task_state2 = s->d_sh_task_state2;
    q.sent_names = s->d_sh_sent_names; q.cand2d = s->d_sh_cand2d; q.goal_cells = s->d_sh_goal_cells;
    return q;
}

// A regeneration pass of xwb_step_autoreset may still be reading the done list and the episode counters on the side queue:
// every other verb that touches them orders `st` behind it first (the next xwb_step_autoreset waits inside its step kernel).
int join_regen(xwb_sim *s, hipStream_t st) {
    if (!s->regen_pending) return XWB_OK;
    if (s->regen_by_epoch) HIP_TRY(launch_xw_wait(s->d_sync + 8, s->epoch_regen, s->d_sync + 4, s->xw.poison_host, st));
    else HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));
    s->regen_pending = false;
    return XWB_OK;
}

// xworld: reset the compacted list (or all), then re-render those envs.
// `beside_render`: the list comes from the step kernel that was just launched on `st` followed by render_all;
// the (latency-bound, two-wavefront) reset kernel then runs on the side stream as soon as the step kernel is
// done, i.e. *beside* render_all.  render_all may read grid rows of finished envs while they are being
// regenerated; those envs' frames are rewritten in full by render(list) below, which waits for both.
int xw_reset_list(xwb_sim *s, int mode, bool keep_done, bool render, hipStream_t st, bool beside_render = false) {
    { const int rcj = join_regen(s, st); if (rcj) return rcj; }
    if (render) { s->frame_src = mode == MODE_RESET_ALL ? 0 : 2; s->draws_since_pack += 1; }
    if (s->shadow_ok) s->shadow_breaks += 1;
    s->shadow_ok = false;                  // the episodes these envs start now are the ones their shadows held
    XwParams p = xw_params(s);
    // 0: the reset kernel clears the done codes; 1: they are kept (step_autoreset); 2: the reset runs on the side stream
    // beside work already queued on `st` that may still read this step's codes -> the list render, which is ordered
    // on `st` after that work, clears them instead
    p.auto_reset = keep_done ? 1 : (beside_render && render ? 2 : 0);
    hipStream_t rs = beside_render ? s->side : st;
    const bool span_sync = beside_render && s->span_step && xw_ego_span(p);
    // full observation: the two queues hand over through epochs in device memory (XwParams::sync) -- the side queue's
    // kernel waits for the step kernel's epoch, the list render for the reset kernel's; no event / barrier packets.
    // The mode is the one the step call chose (s->step_epochs): its kernels are the publishers, already enqueued.
    const bool by_epoch = s->step_epochs && beside_render && render && !p.visible_radius && mode != MODE_RESET_ALL;
    if (by_epoch) {
        HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, p.poison_host, s->side));
        if (++s->epoch_reset == 0) s->epoch_reset = 1;
    } else if (beside_render) {
        // (span path: the map generator only has to wait for the kernel that reads the grids; the goal images are redrawn
        // once the kernels that evaluate pixels from them are through)
        if (span_sync && s->span_epochs) HIP_TRY(launch_xw_wait(s->d_sync + 5, s->epoch_step, s->d_sync + 4, p.poison_host, s->side));
        else HIP_TRY(hipStreamWaitEvent(s->side, span_sync ? s->ev_cells : s->ev_step, 0));
    }
    // (span path with a list render to follow: the goal images are redrawn by that render's first launch, beside the cell tables)
    const bool split = span_sync && render && mode != MODE_RESET_ALL;
    timer_begin(s, s->t_reset, rs);
    if (split) HIP_TRY(launch_xw_reset(p, mode, rs, nullptr, nullptr, 0, 1));
    else if (span_sync && s->span_epochs) HIP_TRY(launch_xw_reset(p, mode, rs, nullptr, s->d_sync + 6, s->epoch_step));
    else HIP_TRY(launch_xw_reset(p, mode, rs, span_sync ? s->ev_step : nullptr));
    timer_end(s, s->t_reset, rs);
    if (by_epoch) {
        HIP_TRY(launch_xw_signal(s->d_sync + 3, s->epoch_reset, s->side));     // queued behind the reset kernel
        p.wait_epoch = s->epoch_reset;
        timer_begin(s, s->t_list, st);
        HIP_TRY(launch_xw_render(p, 1, st));
        timer_end(s, s->t_list, st);
        return XWB_OK;
    }
    if (split) {
        // egocentric span path: the map generator and the front kernels of the new episodes' first frames run on the side
        // queue, beside the big gather (they write nothing the caller reads).  They follow the step's term gather -- it shares
        // their buffers -- which also puts them behind its evaluation kernel, the last reader of the old goal images.
        // Only the short gather that stores those frames runs on the CALLER's stream: it overwrites the terminal frames, which
        // work queued there before this call may still read (xwb.h xwb_reset_done).  (auto_reset == 2: that gather clears the codes.)
        if (s->span_epochs) HIP_TRY(launch_xw_wait(s->d_sync + 7, s->epoch_step, s->d_sync + 4, p.poison_host, rs));
        else HIP_TRY(hipStreamWaitEvent(rs, s->ev_term, 0));
        HIP_TRY(launch_xw_render(p, 7, rs));                 // goal images + cell tables in one launch, then the evaluation
        if (s->span_epochs) {
            if (++s->epoch_reset == 0) s->epoch_reset = 1;
            HIP_TRY(launch_xw_signal(s->d_sync + 3, s->epoch_reset, rs));
            HIP_TRY(launch_xw_wait(s->d_sync + 3, s->epoch_reset, s->d_sync + 4, p.poison_host, st));
        } else {
            HIP_TRY(hipEventRecord(s->ev_reset, s->side));
            HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));
        }
        timer_begin(s, s->t_list, st);
        HIP_TRY(launch_xw_render(p, 6, st));
        timer_end(s, s->t_list, st);
        return XWB_OK;
    }
    if (beside_render) {
        HIP_TRY(hipEventRecord(s->ev_reset, s->side));
        HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));
    }
    if (render) {
        if (mode == MODE_RESET_ALL) {
            timer_begin(s, s->t_render, st);
            HIP_TRY(launch_xw_render(p, 0, st));
            timer_end(s, s->t_render, st);
        } else {
            timer_begin(s, s->t_list, st);
            HIP_TRY(launch_xw_render(p, 1, st));
            timer_end(s, s->t_list, st);
        }
    }
    return XWB_OK;
}

int do_step(xwb_sim *s, const int32_t *actions_dev, int32_t act_rep, bool autoreset, hipStream_t st) {
    if (act_rep < 1) return fail(XWB_ERR_ARG, "act_rep must be >= 1");
    if (s->cfg.game == XWB_SIMPLE_GAME) {
        SgParams p = sg_params(s);
        p.actions = actions_dev; p.act_rep = act_rep; p.auto_reset = autoreset ? 1 : 0;
        if (autoreset) take_reset_counter(s, p);
        timer_begin(s, s->t_step, st);
        HIP_TRY(launch_simple_game(p, st));
        timer_end(s, s->t_step, st);
    } else if (s->cfg.game == XWB_SIMPLE_RACE) {
        RaceParams p = race_params(s);
        p.actions = actions_dev; p.act_rep = act_rep; p.auto_reset = autoreset ? 1 : 0;
        if (autoreset) take_reset_counter(s, p);
        timer_begin(s, s->t_step, st);
        HIP_TRY(launch_simple_race(p, st));
        timer_end(s, s->t_step, st);
    } else {
        // hand-over mode of this call: what xwb_create / xwb_queue_sync_mode found out about `st`; events for a stream
        // nobody probed (no verb synchronises the host by itself)
        const bool epochs = use_epochs(s, st, false);
        s->step_epochs = epochs;
        // xwb_step_autoreset with pre-generated episodes (XwParams::swap_shadow): the step kernel starts the next episode of
        // the envs it finishes, ONE render draws every env, the side queue regenerates the consumed shadows beside it
        const bool pregen = autoreset && s->pregen;
        // ... and a plain step whose xwb_reset_done installs them (XwParams::list_swap): no terminal snapshot, the render reads
        // the live grid.  Only while the caller's verbs leave the shadows alone (a loop of masked / single resets would pay a
        // whole-batch regeneration per call: after a few such breaks the batch stays on the classic path).
        static const bool no_lazy = getenv("XWB_NO_LAZY") != nullptr;
        const bool lazy = !autoreset && s->pregen && s->shadow_breaks < 3 && !no_lazy;
        if (pregen || lazy) {
            if (!s->shadow_ok) {               // first use, or another verb reset envs since: make every env's next episode
                { const int rcj = join_regen(s, st); if (rcj) return rcj; }
                HIP_TRY(launch_xw_reset(shadow_params(s), MODE_RESET_ALL, st));
                s->shadow_ok = true;
            } else if (s->regen_pending && !s->regen_by_epoch) {
                HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));      // (events: the step kernel cannot wait for itself)
                s->regen_pending = false;
            }
        } else {
            const int rcj = join_regen(s, st);
            if (rcj) return rcj;
        }
        s->step_lazy = lazy;
        s->count_sel ^= 1;                     // this step appends to the counter the previous one zeroed
        XwParams p = xw_params(s);
        p.actions = actions_dev; p.act_rep = act_rep;
        if (pregen || lazy) { p.swap_shadow = pregen ? 1 : 2; p.regen_wait = s->regen_pending ? s->epoch_regen : 0; }
        if (++s->epoch_step == 0) s->epoch_step = 1;
        p.sig_epoch = epochs ? s->epoch_step : 0;   // published by the render kernel queued behind the step kernel
        timer_begin(s, s->t_step, st);
        HIP_TRY(launch_xw_step(p, st));
        // exclusive scheduling of two groups: idle XWorld3DNav* groups the step picked run their idle stage now
        if (p.idle_list) HIP_TRY(launch_xw_idle3d(p, st));
        timer_end(s, s->t_step, st);
        s->list_valid = true;
        XwParams pr = xw_params(s);
        pr.sig_epoch = 0;
        const bool span = xw_ego_span(p);
        if (pregen) {
            if (!epochs) { p.sig_epoch = 0; HIP_TRY(hipEventRecord(s->ev_step, st)); }
            timer_begin(s, s->t_render, st);
            HIP_TRY(launch_xw_render(p, 0, st));                     // every env from its live grid; publishes the step epoch
            timer_end(s, s->t_render, st);
            XwParams q = shadow_params(s);
            if (epochs) HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, p.poison_host, s->side));
            else HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
            timer_begin(s, s->t_reset, s->side);
            HIP_TRY(launch_xw_reset(q, MODE_RESET_DONE, s->side));
            timer_end(s, s->t_reset, s->side);
            if (epochs) {
                if (++s->epoch_regen == 0) s->epoch_regen = 1;
                HIP_TRY(launch_xw_signal(s->d_sync + 8, s->epoch_regen, s->side));
            } else {
                HIP_TRY(hipEventRecord(s->ev_reset, s->side));
            }
            s->regen_pending = true; s->regen_by_epoch = epochs;
            s->list_valid = false;
        } else if (autoreset) {
            // Finished envs: reset + first frame of the new episode on the side stream, beside the render of everyone else;
            // their terminal frames are not materialised.
            // Epochs (full observation, and the egocentric span path, whose cells kernel publishes the step epoch): the side
            // queue's first kernel waits for "step kernel complete", which the FIRST kernel of the render publishes; a
            // one-wavefront kernel at the end of this call waits for the side queue's.  The render is enqueued BEFORE the
            // side queue's waiter (publisher first: xw_device.h), and the side queue's signal before the final waiter.
            const bool auto_epochs = epochs && (!p.visible_radius || span);
            if (!auto_epochs) { p.sig_epoch = 0; HIP_TRY(hipEventRecord(s->ev_step, st)); }
            timer_begin(s, s->t_render, st);
            HIP_TRY(launch_xw_render(p, 2, st));
            timer_end(s, s->t_render, st);
            if (auto_epochs) {
                HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, p.poison_host, s->side));
                if (++s->epoch_reset == 0) s->epoch_reset = 1;
            } else {
                HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
            }
            pr.auto_reset = 1;
            // (span path: the goal images of the reset envs are redrawn in the list render's first launch, beside their cell tables;
            // nothing else reads them -- the big render's kernels skip the finished envs)
            timer_begin(s, s->t_reset, s->side);
            HIP_TRY(launch_xw_reset(pr, MODE_RESET_DONE, s->side, nullptr, nullptr, 0, span ? 1 : 0));
            timer_end(s, s->t_reset, s->side);
            timer_begin(s, s->t_list, s->side);
            HIP_TRY(launch_xw_render(pr, span ? 8 : 1, s->side));
            timer_end(s, s->t_list, s->side);
            if (auto_epochs) {
                HIP_TRY(launch_xw_signal(s->d_sync + 3, s->epoch_reset, s->side));      // queued behind the list render
                HIP_TRY(launch_xw_wait(s->d_sync + 3, s->epoch_reset, s->d_sync + 4, p.poison_host, st));
            } else {
                HIP_TRY(hipEventRecord(s->ev_reset, s->side));
                HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));
            }
            s->list_valid = false;
        } else {
            if (!p.visible_radius && !epochs) HIP_TRY(hipEventRecord(s->ev_step, st));
            // Finished envs keep a terminal snapshot of their grid (step kernel) from which the big render draws their
            // last frame, so a following xwb_reset_done can regenerate the live state beside that render right away.
            // The egocentric render reads more than the grid (heading, goal images): there the terminal frames are
            // rendered from the (short) list on the side stream, beside the big render, which skips those envs; a
            // following xwb_reset_done queues behind that list render.
            // On the span path (kernels_xworld_ego.hip) only the front kernels read the env state: ev_step is recorded
            // behind them, the terminal frames leave through a short list gather (ev_term) and the big gather skips them.
            if (p.visible_radius && !span) {
                HIP_TRY(hipEventRecord(s->ev_step, st));
                pr.list_flag = 1;
                pr.ego_list_beside = 1;
                HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
                HIP_TRY(launch_xw_render(pr, 1, s->side));
                HIP_TRY(hipEventRecord(s->ev_term, s->side));
            }
            timer_begin(s, s->t_render, st);
            if (span) {
                p.list_flag = 1;
                s->span_epochs = epochs;
                if (epochs) HIP_TRY(launch_xw_render(p, 4, st));          // (p.sig_epoch = this step's epoch: d_sync[5..7])
                else HIP_TRY(launch_xw_render(p, 4, st, s->ev_step, s->ev_term, s->ev_cells));
            } else {
                HIP_TRY(launch_xw_render(p, p.visible_radius ? 2 : (lazy ? 0 : 3), st));     // (lazy: nothing rewrites the live grid beside it)
            }
            timer_end(s, s->t_render, st);
            if (p.visible_radius && !span) HIP_TRY(hipStreamWaitEvent(st, s->ev_term, 0));
        }
    }
    s->span_step = !autoreset && xw_ego_span(s->xw);
    s->last_path = s->cfg.game != XWB_XWORLD2D ? XWB_PATH_NONE :
                   (s->cfg.visible_radius ? (xw_ego_span(s->xw) ? XWB_PATH_EGO_SPAN : XWB_PATH_EGO_PER_ENV) :
                    (autoreset && s->pregen ? XWB_PATH_PREGEN : (s->step_lazy ? XWB_PATH_LAZY : XWB_PATH_CLASSIC)));
    if (s->cfg.game == XWB_XWORLD2D) {     // a plain step on the classic path drew the finished envs from their terminal snapshots
        s->frame_src = (!autoreset && !s->step_lazy && !s->cfg.visible_radius) ? 1 : 0;
        s->draws_since_pack += 1;
    }
    s->policy_step += 1;
    s->packed_pos += 1;
    s->autoreset_done = autoreset;
    return XWB_OK;
}

// ---- StatePacket wire writer (data_packet.h:313-319, data_packet.cpp:143-162, memory_util.h:307-333) ----
struct Writer {
    uint8_t *p; size_t cap, n;
    void put(const void *d, size_t len) { if (p && n + len <= cap) memcpy(p + n, d, len); n += len; }
    void u64(uint64_t v) { put(&v, 8); }
    void str(const char *s) { size_t len = strlen(s); u64(len); put(s, len + 1); }
};

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
    xwb_sim *s = new xwb_sim();
    s->cfg = *cfg;
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
    if (s->d_sync) (void)use_epochs(s, nullptr, true);   // probe the default stream now; other streams: xwb_queue_sync_mode
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
    for (KernelTimer *t : {&s->t_render, &s->t_step, &s->t_reset, &s->t_list})
        for (auto &ep : t->pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    delete s;
    return XWB_OK;
}

int xwb_reset(xwb_sim *s, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    s->autoreset_done = false;
    if (s->cfg.game != XWB_XWORLD2D) return simple_reset(s, MODE_RESET_ALL, nullptr, st);
    s->list_valid = false;
    return xw_reset_list(s, MODE_RESET_ALL, false, true, st);
}

int xwb_reset_done(xwb_sim *s, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    if (s->autoreset_done) {
        // xwb_step_autoreset / xwb_step_n already reset every env whose code is set (the codes are kept for the caller
        // to read): clearing them is all that is left -- resetting those envs again would skip an episode
        s->autoreset_done = false;
        HIP_TRY(hipMemsetAsync(s->d_done, 0, (size_t)s->n, st));
        return XWB_OK;
    }
    if (s->cfg.game != XWB_XWORLD2D) return simple_reset(s, MODE_RESET_DONE, nullptr, st);
    if (!s->list_valid) {                      // no step since the last reset: rebuild the list from done[]
        { const int rcj = join_regen(s, st); if (rcj) return rcj; }
        XwParams p = xw_params(s);
        HIP_TRY(hipMemsetAsync(p.done_count, 0, sizeof(int32_t), st));
        HIP_TRY(launch_xw_compact(p, MODE_RESET_DONE, st));
    }
    if (s->list_valid && s->step_lazy && s->shadow_ok) {
        // the step kept no terminal snapshot and every env's next episode is pre-generated: the list render installs the
        // shadows of the finished envs and draws their first frames (st); the side queue regenerates what was consumed, for
        // nobody in particular -- the next holder of the done list waits for it device-side
        s->list_valid = false;
        s->frame_src = 2; s->draws_since_pack += 1;
        XwParams p = xw_params(s);
        p.auto_reset = 2; p.list_swap = 1;
        const bool by_epoch = s->step_epochs;
        if (s->regen_pending && !s->regen_by_epoch) { HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0)); s->regen_pending = false; }
        p.wait_slot = 8;
        p.wait_epoch = s->regen_pending ? s->epoch_regen : 0;
        timer_begin(s, s->t_list, st);
        HIP_TRY(launch_xw_render(p, 1, st));
        timer_end(s, s->t_list, st);
        XwParams q = shadow_params(s);
        if (by_epoch) HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, p.poison_host, s->side));
        else HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
        timer_begin(s, s->t_reset, s->side);
        HIP_TRY(launch_xw_reset(q, MODE_RESET_DONE, s->side));
        timer_end(s, s->t_reset, s->side);
        if (by_epoch) {
            if (++s->epoch_regen == 0) s->epoch_regen = 1;
            HIP_TRY(launch_xw_signal(s->d_sync + 8, s->epoch_regen, s->side));
        } else {
            HIP_TRY(hipEventRecord(s->ev_reset, s->side));
        }
        s->regen_pending = true; s->regen_by_epoch = by_epoch;
        return XWB_OK;
    }
    // (a lazy step's render reads the live grid: the classic reset may not rewrite it beside that render)
    const bool beside = s->list_valid && !s->step_lazy;
    s->list_valid = false;
    return xw_reset_list(s, MODE_RESET_DONE, false, true, st, beside);
}

int xwb_reset_masked(xwb_sim *s, const uint8_t *mask_dev, void *stream) {
    if (!s || !mask_dev) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    if (s->cfg.game != XWB_XWORLD2D) return simple_reset(s, MODE_RESET_MASK, mask_dev, st);
    { const int rcj = join_regen(s, st); if (rcj) return rcj; }
    XwParams p = xw_params(s);
    p.mask = mask_dev;
    HIP_TRY(hipMemsetAsync(p.done_count, 0, sizeof(int32_t), st));
    HIP_TRY(launch_xw_compact(p, MODE_RESET_MASK, st));
    s->list_valid = false;
    return xw_reset_list(s, MODE_RESET_MASK, false, true, st);
}

int xwb_reset_env(xwb_sim *s, int32_t env, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (env < 0 || env >= s->n) return fail(XWB_ERR_ARG, "env out of range");
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemsetAsync(s->d_mask, 0, (size_t)s->n, st));
    HIP_TRY(hipMemsetAsync(s->d_mask + env, 1, 1, st));
    return xwb_reset_masked(s, s->d_mask, stream);
}

int xwb_step(xwb_sim *s, const int32_t *actions_dev, int32_t act_rep, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    return do_step(s, actions_dev, act_rep, false, as_stream(stream));
}

int xwb_step_host(xwb_sim *s, const int32_t *actions_host, int32_t act_rep, void *stream) {
    if (!s || !actions_host) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(s->d_actions_in, actions_host, sizeof(int32_t) * (size_t)s->n, hipMemcpyHostToDevice, st));
    return do_step(s, s->d_actions_in, act_rep, false, st);
}

int xwb_step_n(xwb_sim *s, int32_t n_steps, int32_t act_rep, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (n_steps < 1 || act_rep < 1) return fail(XWB_ERR_ARG, "n_steps and act_rep must be >= 1");
    hipStream_t st = as_stream(stream);
    if (s->cfg.game == XWB_XWORLD2D) {                      // one render per step is the work: nothing to fuse
        // one call = one slot of a results ring, as for the simple games: every step writes it, the last one stays
        const int64_t slot = s->packed_pos;
        for (int i = 0; i < n_steps; ++i) {
            s->packed_pos = slot;
            int rc = do_step(s, nullptr, act_rep, true, st);
            if (rc) return rc;
        }
        return XWB_OK;
    }
    timer_begin(s, s->t_step, st);
    if (s->cfg.game == XWB_SIMPLE_GAME) {
        SgParams p = sg_params(s);
        p.actions = nullptr; p.act_rep = act_rep; p.auto_reset = 1; p.n_steps = n_steps;
        take_reset_counter(s, p);
        HIP_TRY(launch_simple_game(p, st));
    } else {
        RaceParams p = race_params(s);
        p.actions = nullptr; p.act_rep = act_rep; p.auto_reset = 1; p.n_steps = n_steps;
        take_reset_counter(s, p);
        HIP_TRY(launch_simple_race(p, st));
    }
    timer_end(s, s->t_step, st);
    s->policy_step += (uint32_t)n_steps;
    s->packed_pos += 1;
    s->autoreset_done = true;
    return XWB_OK;
}

int xwb_step_autoreset(xwb_sim *s, const int32_t *actions_dev, int32_t act_rep, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    return do_step(s, actions_dev, act_rep, true, as_stream(stream));
}

int xwb_ego_render_path(xwb_sim *s, int32_t *path) {
    if (!s || !path) return fail(XWB_ERR_ARG, "NULL argument");
    if (s->cfg.game != XWB_XWORLD2D || s->cfg.visible_radius == 0) return fail(XWB_ERR_STATE, "not an egocentric xworld batch");
    *path = xw_ego_span(s->xw) ? 1 : 0;
    return XWB_OK;
}

int xwb_check_errors(xwb_sim *s, void *stream, int32_t *n_bad) {
    if (!s || !n_bad) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(n_bad, s->d_err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(s->d_err, 0, sizeof(int32_t), st));
    uint32_t timed_out = 0;
    if (s->d_sync) HIP_TRY(hipMemcpyAsync(&timed_out, s->d_sync + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (timed_out) s->poisoned = true;                 // sticky: the device word is never cleared
    XWB_LIVE(s);
    return XWB_OK;
}

int xwb_queue_sync_mode(xwb_sim *s, void *stream, int32_t *mode, int32_t *reason) {
    if (!s || !mode) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    const bool e = use_epochs(s, as_stream(stream), true);
    *mode = e ? XWB_QUEUE_SYNC_EPOCHS : XWB_QUEUE_SYNC_EVENTS;
    if (reason) *reason = s->sync_reason;
    return XWB_OK;
}

int xwb_step_path(xwb_sim *s, int32_t *path, int32_t *sync_mode, int32_t *shadow_breaks) {
    if (!s || !path) return fail(XWB_ERR_ARG, "NULL argument");
    *path = s->last_path;
    if (sync_mode) *sync_mode = s->cfg.game == XWB_XWORLD2D ? (s->step_epochs ? XWB_QUEUE_SYNC_EPOCHS : XWB_QUEUE_SYNC_EVENTS) : XWB_QUEUE_SYNC_AUTO;
    if (shadow_breaks) *shadow_breaks = s->shadow_breaks;
    return XWB_OK;
}

int xwb_queue_sync_forget(xwb_sim *s, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    hipStream_t st = as_stream(stream);
    for (size_t i = 0; i < s->probes.size();)
        if (s->probes[i].st == st) s->probes.erase(s->probes.begin() + (long)i); else ++i;
    return XWB_OK;
}

int xwb_debug_stall_handoff(xwb_sim *s, void *stream, int64_t budget_us) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (!s->d_sync) return fail(XWB_ERR_STATE, "this game has no queue hand-off");
    if (budget_us < 1 || budget_us > 10000000) return fail(XWB_ERR_ARG, "budget_us must be in 1..10 000 000");
    XWB_ON_DEVICE(s);
    // (slot 0 is the probe's; its tokens count up from 1, so this value is never reached)
    HIP_TRY(launch_xw_wait(s->d_sync + 0, 0x7fffffffu, s->d_sync + 4, s->xw.poison_host, as_stream(stream), (unsigned long long)budget_us * 100ull));
    return XWB_OK;
}

int xwb_obs_dev(xwb_sim *s, void **ptr, size_t *bytes_per_env) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (ptr) *ptr = s->d_obs;
    if (bytes_per_env) *bytes_per_env = s->obs_bytes_per_env;
    return XWB_OK;
}

int xwb_bind_results(xwb_sim *s, float *packed_dev) { return xwb_bind_results_ring(s, packed_dev, 1); }

int xwb_bind_results_ring(xwb_sim *s, float *packed_dev, int64_t slots) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (packed_dev && (reinterpret_cast<uintptr_t>(packed_dev) & 7u)) return fail(XWB_ERR_ARG, "results buffer must be 8-byte aligned");
    if (slots < 1) return fail(XWB_ERR_ARG, "slots must be >= 1");
    s->d_packed = reinterpret_cast<float2 *>(packed_dev);
    s->packed_slots = slots;
    s->packed_pos = 0;
    return XWB_OK;
}

int xwb_bind_obs(xwb_sim *s, void *obs_dev) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (obs_dev && (reinterpret_cast<uintptr_t>(obs_dev) & 15u)) return fail(XWB_ERR_ARG, "obs buffer must be 16-byte aligned");
    s->d_obs = obs_dev ? obs_dev : s->d_obs_owned;
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
    HIP_TRY(hipDeviceSynchronize());
    s->shadow_ok = false; s->regen_pending = false;
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
        add(s->d_done_list, n * 4); add(s->d_done_count, 8); add(s->d_fresh, n); add(s->d_perf, 40 * 8);
        add(s->d_goal_cells, n * XW_MAX_GOALS); add(s->d_cand2d, n * 4); add(s->d_agent_dir, n); add(s->d_sent_names, n * 4);
        add(s->d_goal_warp, n * XW_MAX_GOALS * 6 * sizeof(double));     // goal images are re-warped from these on load
        add(s->d_cur_level, n); add(s->d_cur_counter, n * 4); add(s->d_cur_usage, n * 9 * XW_USAGE_BYTES);
    }
    if (include_obs) add(s->d_obs, n * s->obs_bytes_per_env);
    return a;
}

// version 3 (round 4): per-workgroup reset counts (d_reset_partial, sized by num_envs) replaced the single counter, the
// exclusive schedule's group order and the task performance counters joined, count_sel lost its rc_sel bit
constexpr uint32_t XWB_STATE_VERSION = 3;
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
    h.n_arrays = (uint32_t)arrays.size(); h.policy_step = s->policy_step; h.count_sel = (uint32_t)s->count_sel;
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
    const auto arrays = state_arrays(s, h.include_obs != 0);
    if (arrays.size() != h.n_arrays) return fail(XWB_ERR_ARG, "state blob layout mismatch");
    HIP_TRY(hipDeviceSynchronize());
    const uint8_t *r = in_host + sizeof h, *end = in_host + bytes;
    for (auto &a : arrays) {
        uint64_t b;
        if (r + 8 > end) return fail(XWB_ERR_ARG, "truncated state blob");
        memcpy(&b, r, 8); r += 8;
        if (b != a.bytes || r + b > end) return fail(XWB_ERR_ARG, "state blob layout mismatch");
        HIP_TRY(hipMemcpy(a.ptr, r, a.bytes, hipMemcpyHostToDevice));
        r += b;
    }
    s->shadow_ok = false; s->regen_pending = false; s->step_lazy = false;
    s->frame_src = 0; s->draws_since_pack = 0;
    s->policy_step = h.policy_step; s->count_sel = (int)(h.count_sel & 1u); s->list_valid = (h.list_valid & 1u) != 0; s->autoreset_done = (h.list_valid & 2u) != 0;
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

int xwb_xw_pack_grids(xwb_sim *s, uint16_t *grids_dev, uint8_t *flags_dev, void *stream) {
    if (!s || !grids_dev) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (s->cfg.visible_radius) return fail(XWB_ERR_STATE, "egocentric frames are not a function of the cell codes alone (heading, goal poses, shadows): gather the screens");
    if (s->cfg.context > 1) {
        if (!flags_dev) return fail(XWB_ERR_ARG, "context > 1 needs the ring flags");
        if (s->draws_since_pack != 1)
            return fail(XWB_ERR_STATE, "context > 1: the draw state must be packed after EVERY verb that draws frames (a context ring is "
                                       "replayed one draw at a time); re-synchronise with the screens themselves");
    }
    // (everything a verb leaves behind on the side queue writes the pre-generated episodes, never the live state read here)
    HIP_TRY(launch_xw_pack_grids(xw_params(s), s->frame_src, grids_dev, flags_dev, as_stream(stream)));
    s->draws_since_pack = 0;
    return XWB_OK;
}

int xwb_xw_render_grids(xwb_sim *s, const uint16_t *grids_dev, const uint8_t *flags_dev, int32_t n_envs, void *obs_dev, void *stream) {
    if (!s || !grids_dev || !obs_dev) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch");
    if (s->cfg.visible_radius) return fail(XWB_ERR_STATE, "egocentric batches cannot render from cell codes alone");
    if (n_envs < 1) return fail(XWB_ERR_ARG, "n_envs must be >= 1");
    if (s->cfg.context > 1 && !flags_dev) return fail(XWB_ERR_ARG, "context > 1 needs the ring flags");
    if (reinterpret_cast<uintptr_t>(obs_dev) & 15u) return fail(XWB_ERR_ARG, "obs buffer must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(grids_dev) & 1u) return fail(XWB_ERR_ARG, "grids must be 2-byte aligned");
    XwParams q = xw_params(s);
    q.n = n_envs;
    q.grid = const_cast<uint16_t *>(grids_dev);
    q.fresh = const_cast<uint8_t *>(flags_dev);
    q.obs = static_cast<uint8_t *>(obs_dev);
    q.sig_epoch = 0; q.wait_epoch = 0; q.packed = nullptr;
    HIP_TRY(launch_xw_render(q, 0, as_stream(stream)));
    return XWB_OK;
}

int xwb_profile_begin(xwb_sim *s) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    s->profiling = true;
    s->t_render.used = s->t_step.used = s->t_reset.used = s->t_list.used = 0;
    return XWB_OK;
}

int xwb_profile_end(xwb_sim *s, void *stream, const char *kernel, double *avg_us, int64_t *launches) {
    if (!s || !kernel || !avg_us || !launches) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    KernelTimer *t = nullptr;
    if (!strcmp(kernel, "render")) t = &s->t_render;
    else if (!strcmp(kernel, "step")) t = &s->t_step;
    else if (!strcmp(kernel, "reset")) t = &s->t_reset;
    else if (!strcmp(kernel, "list")) t = &s->t_list;
    else return fail(XWB_ERR_ARG, "kernel must be render | step | reset | list");
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    double total_ms = 0;
    for (size_t i = 0; i < t->used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t->pool[i].a, t->pool[i].b));
        total_ms += ms;
    }
    *launches = (int64_t)t->used;
    *avg_us = t->used ? total_ms * 1000.0 / (double)t->used : 0.0;
    return XWB_OK;
}

int xwb_profile_stop(xwb_sim *s) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    s->profiling = false;
    return XWB_OK;
}

}  // extern "C"
