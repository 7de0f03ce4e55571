// xwb_sim.h -- internal to libxwb.so: the batch object behind the opaque xwb_sim of include/xwb.h, and what the host-side
// translation units share (xwb_create.hip: configuration, set-up, create / destroy; xwb_verbs.hip: reset / step and the queue
// hand-off; xwb_getters.hip: getters, per-env host access, packets, sentences; xwb_checkpoint.hip: save / load).
//
// A xwb_sim is the batched counterpart of simulator::SimulatorInterface (simulator_interface.h:40-89): it owns the SoA state
// of num_envs environments in HBM and sequences the kernels in the reference's call order (simulator_interface.cpp:95-143).
// No CPU fallback exists: without a usable gfx950 device xwb_create fails.
#pragma once
#include "../../include/xwb.h"
#include "../../include/xwb_testing.h"
#include "xwb_common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace xwb {
namespace host {

// sets xwb_last_error of the calling thread, returns `code` (xwb_create.hip)
int fail(int code, const std::string &msg);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::xwb::host::fail(XWB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
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

}  // namespace host
}  // namespace xwb

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
    bool draw_off = false;                 // xwb_xw_set_draw(sim, 0): frames are not drawn (their consumer draws them from xwb_xw_pack_grids)
    bool autoreset_done = false;           // the last step call already reset the envs whose codes are still set
    // the done list lives in two buffers and its counter in three, rotated by every xworld step call (list_sel, count_sel): step k
    // appends to list k & 1 / counter k % 3 and zeroes counter (k + 1) % 3, so the regeneration pass of step k - 1, which reads that
    // step's list and counter on the internal queue, is never in the way -- only the one of step k - 2 has to be through
    int count_sel = 0, list_sel = 0;
    uint64_t step_seq = 0;                 // xworld step calls so far
    bool profiling = false;
    xwb::host::KernelTimer t_render, t_step, t_reset, t_list;   // t_list: the list render (first frames of the envs a reset started)
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
    hipEvent_t ev_results = nullptr;       // xwb_gather_results_beside's hand-over when the last step did not run on epochs (made on first use)
    bool results_by_epoch = false;         // the last step call published "step kernel complete" as an epoch in sync[1]
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
    xwb::RaceParams race{};
    // xworld
    uint16_t *d_grid = nullptr;
    int32_t *d_task_steps2 = nullptr, *d_task_state2 = nullptr;
    uint8_t *d_grp_order = nullptr;        // exclusive group scheduling (XwParams::grp_order)
    int32_t *d_idle_list = nullptr, *d_idle_count = nullptr;
    unsigned long long *d_perf = nullptr;  // XwParams::perf
    // pre-generated next episodes (XwParams::shadow / swap_shadow): xwb_step_autoreset's fast path
    bool pregen = false, shadow_ok = false, regen_pending = false, regen_by_epoch = false;
    bool regen_deferred = false, regen_deferred_by_epoch = false;   // xwb_reset_done after a fused step: the pass is queued by the next verb
    bool step_lazy = false;                // the last plain step kept no terminal snapshot: its reset_done installs shadows
    int shadow_breaks = 0;                 // times another verb made the shadows stale (the lazy default path gives up after a few)
    uint32_t epoch_regen = 0, epoch_regen_prev = 0;   // epochs of the last two regeneration passes handed over by epoch ...
    uint64_t regen_seq = 0, regen_seq_prev = 0;       // ... and the step calls (step_seq) whose lists they read
    // look-ahead snapshots (XwParams::snap_grid_*): while snap_ok, set snap_sel holds the grids as the built-in policy's step
    // number snap_step with act_rep = snap_act_rep WILL leave them; every verb that writes the live state without patching the
    // snapshot clears snap_ok, and the next xwb_step then runs step -> render as two launches again
    uint16_t *d_snap_grid[2] = {nullptr, nullptr};
    int snap_sel = 0, snap_act_rep = 1;
    bool snap_ok = false, step_fused = false;
    bool step_pub_queued = false;          // a kernel that publishes the last step call's epoch is in the caller's queue (after a fused
                                           // launch that is xwb_reset_done's list render: until then, results are handed over by an event)
    uint32_t snap_step = 0;
    uint32_t *d_sh_ep = nullptr, *d_done_ep = nullptr;
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
    xwb::EgoTap *d_ego_taps = nullptr;
    uint8_t *d_ego_tab = nullptr;
    int ego_cell_edge = 1;
    uint8_t *d_ego_cache = nullptr;        // lazily filled cache of rendered goal cells (XwParams::ego_cache)
    uint32_t *d_ego_cache_valid = nullptr;
    uint2 *d_ego_cellsrc = nullptr, *d_ego_cellsrc_list = nullptr;
    uint2 *d_ego_miss_list = nullptr;
    int32_t *d_ego_miss_count_list = nullptr;
    uint32_t *d_ego_cellinfo = nullptr;    // span path of the egocentric render (XwParams::ego_span)
    uint2 *d_ego_miss = nullptr;
    int32_t *d_ego_miss_count = nullptr;
    uint32_t *d_ego_xtab = nullptr;
    uint2 *d_ego_clsimg = nullptr;
    uint8_t *d_ego_cls = nullptr, *d_ego_tab3 = nullptr, *d_ego_flat = nullptr, *d_ego_constline = nullptr;
    uint16_t *d_ego_cls_icon = nullptr;
    double *d_goal_warp = nullptr;
    int16_t *d_icon_name = nullptr, *d_name_first = nullptr, *d_name_variants = nullptr;
    uint32_t *d_atlas = nullptr;
    std::vector<uint8_t> tile_table;   // host copy, n_icons x c x 12 x 12
    std::vector<int32_t> icon_type_h, icon_name_h, icon_colored_h;
    // xwb_set_names: the strings behind the name ids (the teacher's sentences are built from them)
    std::vector<std::string> goal_names, icon_names, icon_colors;
    bool have_names = false;
    xwb::XwParams xw{};
    std::vector<void *> allocs;
};

namespace xwb {
namespace host {

template <typename T>
inline int dev_alloc(xwb_sim *s, T **p, size_t count, int fill = 0) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(&q, bytes));
    HIP_TRY(hipMemset(q, fill, bytes));
    s->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return XWB_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

extern const char *const POISON_MSG;
inline bool is_poisoned(xwb_sim *s) {
    if (!s->poisoned && s->h_poison && *(volatile uint32_t *)s->h_poison) s->poisoned = true;
    return s->poisoned;
}
#define XWB_LIVE(s) do { if (::xwb::host::is_poisoned(s)) return ::xwb::host::fail(XWB_ERR_STATE, ::xwb::host::POISON_MSG); } while (0)

// ---- xwb_verbs.hip ----
// may calls on stream `st` hand over through epochs?  may_probe: only xwb_create and xwb_queue_sync_mode run the probe
bool use_epochs(xwb_sim *s, hipStream_t st, bool may_probe);
// the concurrency probe itself: do kernels of `st` and of the batch's internal queue run side by side?  (synchronises both)
bool epoch_probe(xwb_sim *s, hipStream_t st, int *reason);
// the probe, re-selecting the internal stream when it shares `st`'s hardware queue
bool side_beside(xwb_sim *s, hipStream_t st, int *reason);
// -1: the environment / a tool does not override the hand-over mode, 0: events, 1: epochs
int queue_sync_env(int *reason);
void timer_begin(xwb_sim *s, KernelTimer &t, hipStream_t st);
void timer_end(xwb_sim *s, KernelTimer &t, hipStream_t st);
SgParams sg_params(xwb_sim *s);
RaceParams race_params(xwb_sim *s);
XwParams xw_params(xwb_sim *s);
int join_regen(xwb_sim *s, hipStream_t st);
int launch_regen(xwb_sim *s, bool by_epoch);
int flush_regen(xwb_sim *s);
int xw_reset_list(xwb_sim *s, int mode, bool keep_done, bool render, hipStream_t st, bool beside_render = false);

}  // namespace host
}  // namespace xwb
