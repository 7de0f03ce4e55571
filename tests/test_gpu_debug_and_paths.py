"""Round-4 host-side behaviour of the C ABI: the per-batch debug configuration and its one process-wide override
(XWB_DEBUG), xwb_step_path, probing only on explicit calls (xwb_queue_sync_mode / xwb_queue_sync_forget), the checkpoint
blob's version check.  Reference behaviour being replaced: process-global gflags (simulator.cpp:21-27)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAV = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")
OPTS = {"xwd_conf_path": NAV, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_step_path_reports_the_kernel_sequence():
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", OPTS, num_envs=512, seed=3)
    assert sim.step_path()["path"] == "none"                       # no step yet
    sim.step(); sim.reset_done()
    assert sim.step_path() == {"path": "lazy", "queue_sync": sim.queue_sync_mode()[0], "shadow_breaks": 0}
    sim.step(); sim.reset_done()                                   # the second step finds a snapshot: one launch
    assert sim.step_path() == {"path": "lazy_fused", "queue_sync": sim.queue_sync_mode()[0], "shadow_breaks": 0}
    sim.step_autoreset()
    assert sim.step_path()["path"] == "pregen"
    mask = torch.zeros(512, dtype=torch.uint8, device="cuda")
    mask[3] = 1
    for k in range(3):                                             # three foreign resets: the default loop falls back for good
        sim.step(); sim.reset_masked(mask); sim.reset_done()
    sim.step()
    p = sim.step_path()
    assert p["path"] == "classic" and p["shadow_breaks"] >= 3
    sim.close()
    for opts in (dict(OPTS, debug=["no_fused"]), dict(OPTS, context=2)):      # no fused launch: switched off, or a context ring
        s = BatchedSimulator("xworld", opts, num_envs=256)
        s.step(); s.reset_done(); s.step()
        assert s.step_path()["path"] == "lazy", (opts, s.step_path())
        s.close()
    for opts, want in ((dict(OPTS, debug=["no_pregen"]), "classic"), (dict(OPTS, debug=["no_lazy"]), "classic"),
                       (dict(OPTS, obs_format="float32"), "classic"), (dict(OPTS, visible_radius=3), "ego_span"),
                       (dict(OPTS, visible_radius=3, debug=["ego_no_span"]), "ego_per_env")):
        s = BatchedSimulator("xworld", opts, num_envs=256)
        s.step()
        assert s.step_path()["path"] == want, (opts, s.step_path())
        s.close()
    sg = BatchedSimulator("simple_game", {"array_size": 8}, num_envs=64)
    sg.step()
    assert sg.step_path() == {"path": "none", "queue_sync": "auto", "shadow_breaks": 0}
    sg.close()
    with pytest.raises(Exception, match="unknown debug switch"):
        BatchedSimulator("xworld", dict(OPTS, debug=["nonsense"]), num_envs=8)


def test_debug_env_override_is_read_once_per_process():
    """XWB_DEBUG: the one process-wide override of xwb_config.debug_flags (for tools that cannot reach the configuration)."""
    _torch()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from xworld_amd.batched import BatchedSimulator\n"
            "s = BatchedSimulator('xworld', %r, num_envs=256)\n"
            "s.step(); print('PATH', s.step_path()['path'], s.ego_render_path)\n") % (ROOT, dict(OPTS, visible_radius=3))
    for env, want in (({}, "PATH ego_span span"), ({"XWB_DEBUG": "ego_no_span,bogus_entry"}, "PATH ego_per_env per_env")):
        e = dict(os.environ)
        e.pop("XWB_DEBUG", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0 and want in r.stdout, (r.stdout, r.stderr[-1500:])
        assert ("unknown entry 'bogus_entry'" in r.stderr) == bool(env)


def test_streams_are_probed_only_on_request():
    """No step verb of the C ABI synchronises the host: a stream nobody probed hands over through events; xwb_queue_sync_mode
    probes it.  (BatchedSimulator makes that explicit call itself the first time it sees a stream handle -- ADVICE round 4,
    tests below -- so this test drives the C ABI directly for the unprobed calls.)"""
    torch = _torch()
    from xworld_amd import lib
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", OPTS, num_envs=1024, seed=5)
    ref = BatchedSimulator("xworld", OPTS, num_envs=1024, seed=5)
    default_mode = sim.queue_sync_mode()                          # the default stream was probed by xwb_create
    assert default_mode[1] in ("probe_ok", "probe_failed", "tool", "env")
    mine = torch.cuda.Stream()
    mh = C.c_void_p(mine.cuda_stream)
    for t in range(5):
        lib.check(sim.L.xwb_step(sim.h, None, 1, mh)); lib.check(sim.L.xwb_reset_done(sim.h, mh))
        ref.step(); ref.reset_done()
    assert sim.step_path()["queue_sync"] == "events"              # never probed: events
    mode = sim.queue_sync_mode(mine)                              # the explicit probe
    assert mode[1] in ("probe_ok", "probe_failed", "tool", "env")
    for t in range(5):
        sim.step(stream=mine); sim.reset_done(stream=mine)
        ref.step(); ref.reset_done()
    assert sim.step_path()["queue_sync"] == mode[0]
    lib.check(sim.L.xwb_queue_sync_forget(sim.h, C.c_void_p(mine.cuda_stream)))
    lib.check(sim.L.xwb_step(sim.h, None, 1, mh)); ref.step()
    assert sim.step_path()["queue_sync"] == "events"              # forgotten: events again
    mine.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(sim.obs, ref.obs) and torch.equal(sim.reward, ref.reward)
    sim.close(); ref.close()


def test_internal_stream_is_chosen_to_run_beside_the_callers():
    """HIP maps streams onto a few hardware queues: whichever of many caller streams a batch is probed on, the batch ends up
    with an internal stream that runs beside it (xwb_queue_sync_mode re-selects the internal stream when the probe finds no
    concurrency), streams that passed earlier keep passing, and the results do not depend on any of it."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    if os.environ.get("XWB_QUEUE_SYNC"):
        pytest.skip("the hand-over mode is forced from the environment")
    streams = [torch.cuda.Stream() for _ in range(9)]             # more streams than hardware queues: some share one
    ref = BatchedSimulator("xworld", OPTS, num_envs=2048, seed=6)
    sims = [BatchedSimulator("xworld", OPTS, num_envs=2048, seed=6) for _ in streams]
    if ref.queue_sync_mode()[1] == "tool":
        pytest.skip("a tool serialises kernels")
    for sim, st in zip(sims, streams):
        assert sim.queue_sync_mode(st) == ("epochs", "probe_ok"), (st, sim.queue_sync_mode(st))
        assert sim.queue_sync_mode() == ("epochs", "probe_ok")     # the default stream, probed by xwb_create, still passes
    for t in range(12):
        ref.step(); ref.reset_done()
        for sim, st in zip(sims, streams):
            sim.step(stream=st); sim.reset_done(stream=st)
    torch.cuda.synchronize()
    for sim in sims:
        assert sim.step_path()["queue_sync"] == "epochs"
        assert torch.equal(sim.obs, ref.obs) and torch.equal(sim.reward, ref.reward)
        sim.close()
    ref.close()


def test_checkpoint_blob_of_another_version_is_named():
    _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("simple_game", {"array_size": 16}, num_envs=64)
    sim.step()
    blob = bytearray(sim.save_state())
    assert blob[:8] == b"XWBSTATE" and int(np.frombuffer(bytes(blob[8:12]), np.uint32)[0]) == 4
    blob[8:12] = np.uint32(2).tobytes()                            # what round 3 wrote (rounds 4-5: 3)
    with pytest.raises(Exception, match="version 2"):
        sim.load_state(np.frombuffer(bytes(blob), np.uint8))
    blob[:8] = b"NOTSTATE"
    with pytest.raises(Exception, match="not a state blob"):
        sim.load_state(np.frombuffer(bytes(blob), np.uint8))
    sim.close()


def test_step_host_pageable_and_pinned_actions():
    """xwb_step_host: pageable host memory is staged through the device, page-locked memory is read by the step kernel in
    place; both equal the device-pointer call, skipped envs (XWB_ACTION_SKIP) and bad ids included."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 2048
    sims = [BatchedSimulator("xworld", OPTS, num_envs=n, seed=9) for _ in range(3)]
    rng = np.random.default_rng(4)
    pinned = torch.empty(n, dtype=torch.int32).pin_memory()
    for t in range(40):
        a = rng.integers(-1, 4, n).astype(np.int32)
        if t == 7:
            a[11] = 9                                               # out of range: counted, env untouched
        sims[0].step(torch.from_numpy(a).cuda())
        sims[1].step_host(a)                                        # pageable numpy memory
        torch.cuda.synchronize()                                    # (the pinned buffer is rewritten below: the last call has read it)
        pinned.copy_(torch.from_numpy(a))
        sims[2].step_host(pinned)
        for s in sims:
            s.reset_done()
        torch.cuda.synchronize()
        for s in sims[1:]:
            assert torch.equal(s.obs, sims[0].obs) and torch.equal(s.reward, sims[0].reward), t
            assert torch.equal(s.game_over_codes, sims[0].game_over_codes) and torch.equal(s.actions, sims[0].actions), t
    assert [s.check_errors() for s in sims] == [1, 1, 1]
    # what the kernel would read as int32 pairs, or past the end, is refused before it gets there
    for bad in (np.zeros(n, dtype=np.int64), np.zeros(n - 1, dtype=np.int32), np.zeros((n, 2), dtype=np.int32)[:, 0],
                torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int32, device="cuda")):
        with pytest.raises(ValueError, match="step_host"):
            sims[1].step_host(bad)
    for s in sims:
        s.close()


def test_a_new_torch_stream_is_probed_before_its_first_verb():
    """A PyTorch user on a non-default stream must not end up on the event hand-over without being told: BatchedSimulator
    probes a stream handle the first time it sees it; xwb_step_path then reports that call's mode."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    sim = BatchedSimulator("xworld", OPTS, num_envs=512, seed=3)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sim.step(stream=st)
        sim.reset_done(stream=st)
    st.synchronize()
    mode, reason = sim.queue_sync_mode(st)
    assert reason != "not_probed", (mode, reason)
    assert sim.step_path()["queue_sync"] == mode
    # a stream that is being captured is not probed (the probe synchronises it) -- the check is on the handle passed, whatever
    # torch's current stream is -- and nothing is remembered about it: it is probed once the capture is over
    cap = torch.cuda.Stream()
    x = torch.ones(64, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        g.capture_begin()
        try:
            x.add_(1)
            m2, r2 = sim.queue_sync_mode(cap)
        finally:
            g.capture_end()
    assert (m2, r2) == ("events", "not_probed")
    cap.synchronize()
    assert sim.queue_sync_mode(cap)[1] != "not_probed"
    sim.close()
