// xwb_verbs.hip -- host side of libxwb.so, part 2: the verbs (reset / step / their variants) and the hand-off between the
// batch's two queues.  Kernel sequences follow the reference's call order (simulator_interface.cpp:95-143).
#include "xwb_sim.h"

using namespace xwb;
using namespace xwb::host;

namespace xwb {
namespace host {

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

// Make s->side a stream whose kernels run beside those of `st`.  HIP maps streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4) in creation order; an internal queue that shares the CALLER's hardware queue runs nothing beside the
// caller's kernels: the map generator then follows the render it was meant to hide behind (C4: 0.190 instead of 0.113 ms per
// step -- seen with the first batch created after an RCCL communicator, and with pool streams of the caller).  If the probe
// of (st, side) finds no concurrency, up to seven further streams are tried; a candidate must also still run beside every
// stream that passed its probe earlier.  The old stream is idle when it is replaced (the probe drains it).  Returns the
// verdict for `st`; nothing changes when no candidate passes.
bool side_beside(xwb_sim *s, hipStream_t st, int *reason) {
    if (epoch_probe(s, st, reason) || *reason != XWB_SYNC_REASON_PROBE_FAILED) return *reason == XWB_SYNC_REASON_PROBE_OK;
    // the streams that passed earlier are probed again against each candidate -- but only those that are still alive: a caller
    // may have destroyed one without xwb_queue_sync_forget.  hipStreamQuery validates the handle (hipSuccess / hipErrorNotReady
    // for a live stream); anything else means "forget this stream", not "candidate rejected".
    std::vector<hipStream_t> keep;
    for (size_t i = 0; i < s->probes.size();) {
        const hipStream_t q = s->probes[i].st;
        if (q != st && q != nullptr) {
            const hipError_t live = hipStreamQuery(q);
            if (live != hipSuccess && live != hipErrorNotReady) { (void)hipGetLastError(); s->probes.erase(s->probes.begin() + (long)i); continue; }
        }
        if (s->probes[i].ok && q != st) keep.push_back(q);
        ++i;
    }
    hipStream_t original = s->side;
    std::vector<hipStream_t> rejected;                              // kept alive until the choice is made: the next one maps elsewhere
    bool found = false;
    for (int attempt = 0; attempt < 7 && !found; ++attempt) {
        hipStream_t alt = nullptr;
        if (hipStreamCreateWithFlags(&alt, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        s->side = alt;
        int r = 0;
        found = epoch_probe(s, st, &r);
        for (size_t k = 0; found && k < keep.size(); ++k) found = epoch_probe(s, keep[k], &r);
        if (!found) rejected.push_back(alt);
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    if (!found) { s->side = original; *reason = XWB_SYNC_REASON_PROBE_FAILED; return false; }
    (void)hipStreamDestroy(original);
    for (size_t i = 0; i < s->probes.size();)                       // verdicts of "no concurrency" were about the old stream
        if (!s->probes[i].ok) s->probes.erase(s->probes.begin() + (long)i); else ++i;
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
    const bool ok = side_beside(s, st, &reason);
    if (s->probes.size() >= 16) s->probes.erase(s->probes.begin());
    s->probes.push_back(xwb_sim::StreamProbe{st, ok, reason});
    s->sync_reason = reason;
    return ok;
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
    p.no_draw = s->draw_off ? 1 : 0;
    p.list_flag = 2;
    p.done_list = s->d_done_list + (size_t)s->list_sel * (size_t)s->n;
    p.done_ep = s->d_done_ep + (size_t)s->list_sel * (size_t)s->n;
    p.done_count = s->d_done_count + s->count_sel;
    p.done_count_next = s->d_done_count + (s->count_sel + 1) % 3;
    if (s->d_idle_count) { p.idle_count = s->d_idle_count + s->count_sel; p.idle_count_next = s->d_idle_count + (s->count_sel + 1) % 3; }
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
    { const int rcf = flush_regen(s); if (rcf) return rcf; }
    if (!s->regen_pending) return XWB_OK;
    if (s->regen_by_epoch) HIP_TRY(launch_xw_wait(s->d_sync + 8, s->epoch_regen, s->d_sync + 4, s->xw.poison_host, st));
    else HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0));
    s->regen_pending = false;
    return XWB_OK;
}

// The regeneration pass of the lazy loop: the episodes after the ones the last step's finished envs are about to start (or have
// just started), into the free shadow slots, on the internal queue behind that step's kernel.
int launch_regen(xwb_sim *s, bool by_epoch) {
    XwParams q = shadow_params(s);
    if (by_epoch) HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, s->xw.poison_host, s->side));
    else HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
    timer_begin(s, s->t_reset, s->side);
    HIP_TRY(launch_xw_reset(q, MODE_RESET_DONE, s->side));
    timer_end(s, s->t_reset, s->side);
    if (by_epoch) {
        s->epoch_regen_prev = s->epoch_regen; s->regen_seq_prev = s->regen_seq; s->regen_seq = s->step_seq;
        if (++s->epoch_regen == 0) s->epoch_regen = 1;
        HIP_TRY(launch_xw_signal(s->d_sync + 8, s->epoch_regen, s->side));
    } else {
        HIP_TRY(hipEventRecord(s->ev_reset, s->side));
    }
    s->regen_pending = true; s->regen_by_epoch = by_epoch;
    return XWB_OK;
}

// ... which xwb_reset_done leaves to the next verb after a fused step (see there): every verb that steps, or that touches what
// the pass reads or writes (join_regen), queues it first -- with the list and counter of the step it belongs to still current
int flush_regen(xwb_sim *s) {
    if (!s->regen_deferred) return XWB_OK;
    s->regen_deferred = false;
    return launch_regen(s, s->regen_deferred_by_epoch);
}

// xworld: reset the compacted list (or all), then re-render those envs.
// `beside_render`: the list comes from the step kernel that was just launched on `st` followed by render_all;
// the (latency-bound, two-wavefront) reset kernel then runs on the side stream as soon as the step kernel is
// done, i.e. *beside* render_all.  render_all may read grid rows of finished envs while they are being
// regenerated; those envs' frames are rewritten in full by render(list) below, which waits for both.
int xw_reset_list(xwb_sim *s, int mode, bool keep_done, bool render, hipStream_t st, bool beside_render) {
    { const int rcj = join_regen(s, st); if (rcj) return rcj; }
    if (render) { s->frame_src = mode == MODE_RESET_ALL ? 0 : 2; s->draws_since_pack += 1; }
    if (s->shadow_ok) s->shadow_breaks += 1;
    s->shadow_ok = false;                  // the episodes these envs start now are the ones their shadows held
    s->snap_ok = false;                    // ... and the live grids are rewritten without the snapshot
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
        { const int rcf = flush_regen(s); if (rcf) return rcf; }
        const bool epochs = use_epochs(s, st, false);
        s->step_epochs = epochs;
        // xwb_step_autoreset with pre-generated episodes (XwParams::swap_shadow): the step kernel starts the next episode of
        // the envs it finishes, ONE render draws every env, the side queue regenerates the consumed shadows beside it
        const bool pregen = autoreset && s->pregen;
        // ... and a plain step whose xwb_reset_done installs them (XwParams::list_swap): no terminal snapshot, the render reads
        // the live grid.  Only while the caller's verbs leave the shadows alone (a loop of masked / single resets would pay a
        // whole-batch regeneration per call: after a few such breaks the batch stays on the classic path).
        const bool lazy = !autoreset && s->pregen && s->shadow_breaks < 3 && !(s->cfg.debug_flags & XWB_DEBUG_NO_LAZY);
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
        s->count_sel = (s->count_sel + 1) % 3; // this step appends to the counter the previous one zeroed ...
        s->list_sel ^= 1;                      // ... and to the list the one before it filled
        s->step_seq += 1;
        XwParams p = xw_params(s);
        p.actions = actions_dev; p.act_rep = act_rep;
        if (pregen) { p.swap_shadow = 1; p.regen_wait = s->regen_pending ? s->epoch_regen : 0; }   // (it rewrites shadows: the newest pass)
        if (lazy) {
            // the buffers this step writes were last read by the regeneration pass of the step call two back (xwb_sim.h count_sel)
            p.swap_shadow = 2;
            p.regen_wait = !s->regen_pending || !s->regen_by_epoch ? 0u : (s->regen_seq + 2 <= s->step_seq ? s->epoch_regen : s->epoch_regen_prev);
        }
        // A lazy step under the built-in policy also writes the grids as the NEXT step will leave them (XwParams::snap_grid_out);
        // when the previous verbs kept such a snapshot current for exactly this step, this call is ONE launch: render blocks that
        // draw from it, step blocks beside them (XWB_PATH_LAZY_FUSED).
        const bool snaps = lazy && s->d_snap_grid[0] != nullptr && actions_dev == nullptr;
        const bool fused = snaps && s->snap_ok && s->snap_step == s->policy_step && s->snap_act_rep == act_rep && !s->draw_off;
        if (snaps) {
            p.snap_grid_out = s->d_snap_grid[s->snap_sel ^ 1];
            if (fused) p.snap_grid_in = s->d_snap_grid[s->snap_sel];
        }
        if (++s->epoch_step == 0) s->epoch_step = 1;
        p.sig_epoch = epochs ? s->epoch_step : 0;   // published by the render kernel queued behind the step kernel
        if (!fused) {
            timer_begin(s, s->t_step, st);
            HIP_TRY(launch_xw_step(p, st));
            // exclusive scheduling of two groups: idle XWorld3DNav* groups the step picked run their idle stage now
            if (p.idle_list) HIP_TRY(launch_xw_idle3d(p, st));
            timer_end(s, s->t_step, st);
        }
        if (snaps) s->snap_sel ^= 1;
        s->snap_ok = snaps;
        s->snap_step = s->policy_step + 1u;
        s->snap_act_rep = act_rep;
        s->step_fused = fused;
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
                s->epoch_regen_prev = s->epoch_regen; s->regen_seq_prev = s->regen_seq; s->regen_seq = s->step_seq;
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
        } else if (fused) {
            // the step is complete when this kernel is: nothing publishes its epoch here -- xwb_reset_done's list render does
            p.sig_epoch = 0;
            timer_begin(s, s->t_render, st);
            HIP_TRY(launch_xw_step_render(p, st));
            timer_end(s, s->t_render, st);
            if (!epochs) HIP_TRY(hipEventRecord(s->ev_step, st));
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
    // sync[1] = this call's epoch once its step kernel is complete: published by render_all's first thread on every full-observation
    // path that hands over through epochs (the egocentric paths publish other slots, or record events)
    s->results_by_epoch = s->cfg.game == XWB_XWORLD2D && s->step_epochs && !s->cfg.visible_radius;
    s->step_pub_queued = !s->step_fused;
    s->last_path = s->cfg.game != XWB_XWORLD2D ? XWB_PATH_NONE :
                   (s->cfg.visible_radius ? (xw_ego_span(s->xw) ? XWB_PATH_EGO_SPAN : XWB_PATH_EGO_PER_ENV) :
                    (autoreset && s->pregen ? XWB_PATH_PREGEN : (s->step_lazy ? (s->step_fused ? XWB_PATH_LAZY_FUSED : XWB_PATH_LAZY) : XWB_PATH_CLASSIC)));
    if (s->cfg.game == XWB_XWORLD2D) {     // a plain step on the classic path drew the finished envs from their terminal snapshots
        s->frame_src = (!autoreset && !s->step_lazy && !s->cfg.visible_radius) ? 1 : 0;
        s->draws_since_pack += 1;
    }
    s->policy_step += 1;
    s->packed_pos += 1;
    s->autoreset_done = autoreset;
    return XWB_OK;
}

}  // namespace host
}  // namespace xwb

extern "C" {

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
        // (the installs go to the live state AND to the snapshot of it that the next fused step draws from; after a fused step
        // this render is the first kernel behind the step in the caller's queue: it publishes that step's epoch)
        if (s->snap_ok) { p.snap_grid_out = s->d_snap_grid[s->snap_sel]; p.snap_act_rep = s->snap_act_rep; }
        p.sig_epoch = s->step_fused && by_epoch ? s->epoch_step : 0;
        if (p.sig_epoch) s->step_pub_queued = true;
        if (s->regen_pending && !s->regen_by_epoch) { HIP_TRY(hipStreamWaitEvent(st, s->ev_reset, 0)); s->regen_pending = false; }
        p.wait_slot = 8;
        p.wait_epoch = s->regen_pending ? s->epoch_regen : 0;
        timer_begin(s, s->t_list, st);
        HIP_TRY(launch_xw_render(p, 1, st));
        timer_end(s, s->t_list, st);
        // Behind a fused step + render launch the regeneration cannot start before this list render does (it publishes the step's
        // epoch), and nothing needs it before the next step call: it is queued at the top of that call (flush_regen), where it
        // runs beside the render exactly as it would from here -- but a caller that synchronises the device after this verb
        // does not wait 70 us for pre-generated episodes nobody has asked for yet.
        if (s->step_fused) { s->regen_deferred = true; s->regen_deferred_by_epoch = by_epoch; return XWB_OK; }
        return launch_regen(s, by_epoch);
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
    // Pinned (page-locked, device-mapped) host memory: the step kernel reads the ids where they are -- 4 bytes per env over PCIe
    // inside the kernel's first round trip -- instead of behind a copy operation of its own (C4, 131 KB: the copy is ~20 us of
    // a 112 us step, the in-kernel read ~2).  The caller keeps the buffer unchanged until `stream` has passed the call, as for
    // any asynchronous copy from pinned memory.  Pageable memory: staged through the batch's device buffer as before.
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, actions_host) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
        return do_step(s, static_cast<const int32_t *>(attr.devicePointer), act_rep, false, st);
    (void)hipGetLastError();
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

int xwb_run(xwb_sim *s, int32_t iterations, int32_t act_rep, int32_t flags, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (iterations < 1 || act_rep < 1) return fail(XWB_ERR_ARG, "iterations and act_rep must be >= 1");
    if (flags & ~XWB_RUN_AUTORESET) return fail(XWB_ERR_ARG, "unknown flag");
    for (int32_t i = 0; i < iterations; ++i) {
        int rc = (flags & XWB_RUN_AUTORESET) ? xwb_step_autoreset(s, nullptr, act_rep, stream) : xwb_step(s, nullptr, act_rep, stream);
        if (rc == XWB_OK && !(flags & XWB_RUN_AUTORESET)) rc = xwb_reset_done(s, stream);
        if (rc) return rc;
    }
    return XWB_OK;
}

int xwb_step_autoreset(xwb_sim *s, const int32_t *actions_dev, int32_t act_rep, void *stream) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    return do_step(s, actions_dev, act_rep, true, as_stream(stream));
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

// xwb_comm.hip's way in (it only uses the public ABI otherwise): the rows the LAST step call wrote into the results ring, and
// `beside` (the communicator's stream) ordered behind that call's step kernel -- through the step's epoch when it published one
// (nothing is enqueued on the caller's stream then), else through one event recorded on `step_stream`.
extern "C" __attribute__((visibility("hidden"))) int xwb_internal_last_results(xwb_sim *s, void *beside, void *step_stream, const float **rows, int32_t *n) {
    if (!s || !rows || !n) return fail(XWB_ERR_ARG, "NULL argument");
    XWB_ON_DEVICE(s);
    XWB_LIVE(s);
    if (!s->d_packed) return fail(XWB_ERR_STATE, "no results ring is bound (xwb_bind_results / xwb_bind_results_ring)");
    if (s->packed_pos < 1) return fail(XWB_ERR_STATE, "no step has written the results ring yet");
    *rows = reinterpret_cast<const float *>(s->d_packed + (size_t)((s->packed_pos - 1) % s->packed_slots) * (size_t)s->n);
    *n = s->n;
    hipStream_t bs = reinterpret_cast<hipStream_t>(beside);
    const bool by_epoch = s->results_by_epoch && s->d_sync && s->step_pub_queued;   // (publisher first, waiter second: xw_device.h)
    if (by_epoch) {
        HIP_TRY(launch_xw_wait(s->d_sync + 1, s->epoch_step, s->d_sync + 4, s->xw.poison_host, bs));
    } else {
        if (!s->ev_results) HIP_TRY(hipEventCreateWithFlags(&s->ev_results, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(s->ev_results, as_stream(step_stream)));
        HIP_TRY(hipStreamWaitEvent(bs, s->ev_results, 0));
    }
    return by_epoch ? 1 : 0;                                    // (>= 0: fine; 1 = nothing was enqueued on the caller's stream)
}

int xwb_bind_obs(xwb_sim *s, void *obs_dev) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (obs_dev && (reinterpret_cast<uintptr_t>(obs_dev) & 15u)) return fail(XWB_ERR_ARG, "obs buffer must be 16-byte aligned");
    s->d_obs = obs_dev ? obs_dev : s->d_obs_owned;
    return XWB_OK;
}

int xwb_xw_set_draw(xwb_sim *s, int32_t on) {
    if (!s) return fail(XWB_ERR_ARG, "sim is NULL");
    if (s->cfg.game != XWB_XWORLD2D) return fail(XWB_ERR_STATE, "not an xworld batch (the other games write their observation inside the step kernel)");
    if (s->cfg.visible_radius && !on) return fail(XWB_ERR_STATE, "egocentric frames cannot be drawn elsewhere from the cell codes: they stay on");
    s->draw_off = !on;
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
    q.sig_epoch = 0; q.wait_epoch = 0; q.packed = nullptr; q.no_draw = 0;
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
