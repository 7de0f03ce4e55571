"""The step loop's queue hand-off (include/xwb.h xwb_queue_sync_mode): epochs in device memory are only used where they
are safe, and fail loudly where they are not.

* the per-stream concurrency probe: reports its verdict; with ONE hardware queue (GPU_MAX_HW_QUEUES=1: the caller's stream
  and the batch's internal stream share it) it fails and the batch falls back to events -- and even with epochs FORCED on
  that shared queue nothing deadlocks, because every publisher is enqueued before its waiter;
* foreign streams with work in flight (16 torch streams, a live single-rank RCCL communicator): every mode gives the same
  rollout, frame for frame, also when the verbs are issued on a non-default stream;
* the watchdog: an unreleased wait poisons the batch -- every later verb fails with XWB_ERR_STATE.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _rollout_hash(queue_sync, extra=None, n=4096, steps=120, stream=None, between=None):
    """step + reset_done / step_autoreset mixed, frames + rewards + codes hashed every step"""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    opts = {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "color": True, "queue_sync": queue_sync}
    opts.update(extra or {})
    sim = BatchedSimulator("xworld", opts, num_envs=n, seed=3, policy_seed=4)
    mode = sim.queue_sync_mode(stream)
    h = hashlib.sha256()
    for t in range(steps):
        if between is not None:
            between(t)
        if t % 4 == 3:
            sim.step_autoreset(stream=stream)
        else:
            sim.step(stream=stream)
        if stream is not None:
            stream.synchronize()
        h.update(sim.obs.cpu().numpy().tobytes())
        h.update(sim.reward.cpu().numpy().tobytes())
        h.update(sim.game_over_codes.cpu().numpy().tobytes())
        if t % 4 != 3:
            sim.reset_done(stream=stream)
    if stream is not None:
        stream.synchronize()
    h.update(sim.obs.cpu().numpy().tobytes())
    assert sim.check_errors(stream) == 0
    sim.close()
    return h.hexdigest(), mode


_SUB = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from test_gpu_queue_sync import _rollout_hash\n"
        "h, mode = _rollout_hash(%r, %s, n=2048, steps=80)\n"
        "print('RESULT', h, mode[0], mode[1])\n")


def _sub(queue_sync, extra, env):
    e = dict(os.environ)
    e.pop("XWB_QUEUE_SYNC", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _SUB % (ROOT, os.path.join(ROOT, "tests"), queue_sync, repr(extra))],
                       capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    return line[1], line[2], line[3]


@pytest.mark.parametrize("extra", [pytest.param(None, id="full"), pytest.param({"visible_radius": 3}, id="ego", marks=pytest.mark.slow)])
def test_probe_falls_back_on_a_shared_hardware_queue(extra):
    ref, m0, r0 = _sub("events", extra, {})
    assert (m0, r0) == ("events", "config")
    # default environment: the probe decides; whatever it says, the rollout is the same
    h, m, r = _sub("auto", extra, {})
    assert h == ref and (m, r) in (("epochs", "probe_ok"), ("events", "probe_failed")), (m, r)
    # one hardware queue for every stream of the process: the probe's waiter expires, the batch uses events
    h, m, r = _sub("auto", extra, {"GPU_MAX_HW_QUEUES": "1"})
    assert h == ref and (m, r) == ("events", "probe_failed"), (m, r)
    # epochs forced on that shared queue: publisher-before-waiter ordering keeps it alive and exact
    h, m, r = _sub("epochs", extra, {"GPU_MAX_HW_QUEUES": "1"})
    assert h == ref and (m, r) == ("epochs", "config"), (m, r)
    if extra:                                  # (how the mode is chosen does not depend on the render path: once is enough)
        return
    # the environment override still works and is reported as such
    h, m, r = _sub("auto", extra, {"XWB_QUEUE_SYNC": "epochs"})
    assert h == ref and (m, r) == ("epochs", "env"), (m, r)
    h, m, r = _sub("auto", extra, {"AMD_SERIALIZE_KERNEL": "3"})
    assert h == ref and (m, r) == ("events", "tool"), (m, r)


@pytest.mark.parametrize("extra", [pytest.param(None, id="full"), pytest.param({"visible_radius": 3}, id="ego", marks=pytest.mark.slow)])
def test_foreign_streams_with_work_in_flight(extra):
    torch = _torch()
    ref, _ = _rollout_hash("events", extra, n=2048, steps=60)
    streams = [torch.cuda.Stream() for _ in range(16)]
    a = torch.randn(1024, 1024, device="cuda")
    outs = [torch.empty_like(a) for _ in streams]

    def churn(t):
        if t % 3 == 0:
            for s, o in zip(streams, outs):
                with torch.cuda.stream(s):
                    for _ in range(4):
                        torch.mm(a, a, out=o)
    churn(0)
    for mode in ("epochs", "auto", "events"):
        h, m = _rollout_hash(mode, extra, n=2048, steps=60, between=churn)
        assert h == ref, (mode, m)
    # the verbs issued on a caller stream that is not the default one (probed when first seen)
    mine = torch.cuda.Stream()
    for mode in ("auto", "epochs"):
        h, m = _rollout_hash(mode, extra, n=2048, steps=60, stream=mine, between=churn)
        assert h == ref, (mode, m)
    torch.cuda.synchronize()


def test_beside_a_live_rccl_communicator():
    """RCCL creates its own streams / hardware queues; a single-rank group is what one device allows."""
    torch = _torch()
    import torch.distributed as dist
    ref, _ = _rollout_hash("events", None, n=2048, steps=50)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", world_size=1, rank=0)
    try:
        x = torch.ones(1 << 20, device="cuda")

        def allreduce(t):
            if t % 5 == 0:
                dist.all_reduce(x)
        allreduce(0)
        torch.cuda.synchronize()
        for mode in ("auto", "epochs", "events"):
            h, m = _rollout_hash(mode, None, n=2048, steps=50, between=allreduce)
            assert h == ref, (mode, m)
    finally:
        dist.destroy_process_group()


def test_watchdog_poisons_the_batch():
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    from xworld_amd.lib import XwbError, check
    sim = BatchedSimulator("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7}, num_envs=256)
    for _ in range(5):
        sim.step()
        sim.reset_done()
    assert sim.check_errors() == 0
    check(sim.L.xwb_debug_stall_handoff(sim.h, None, 2000))          # a wait nobody releases, 2 ms watchdog
    torch.cuda.synchronize()
    for verb in (sim.step, sim.reset_done, sim.reset, sim.step_autoreset, sim.check_errors, sim.env_state, sim.queue_sync_mode,
                 lambda: sim.step_n(3), sim.save_state):
        with pytest.raises(XwbError, match="poisoned"):
            verb()
    sim.close()
    # the process and the device are fine: a new batch works
    sim = BatchedSimulator("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7}, num_envs=256)
    sim.step()
    assert sim.check_errors() == 0
    sim.close()
    # the simple games have no internal stream
    sg = BatchedSimulator("simple_game", {"array_size": 8}, num_envs=16)
    assert sg.queue_sync_mode() == ("events", "not_used")
    sg.close()


def test_step_n_writes_one_ring_slot_per_call():
    """xwb_bind_results_ring: one xwb_step_n call = one slot, for xworld as for the simple games (last step stays)."""
    torch = _torch()
    from xworld_amd.batched import BatchedSimulator
    for game, opts in (("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7}),
                       ("simple_game", {"array_size": 8}), ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0})):
        sim = BatchedSimulator(game, opts, num_envs=128, policy_seed=2)
        ring = torch.full((3, 128, 2), -5.0, device="cuda")
        sim.bind_results_ring(ring)
        sim.step_n(4)
        torch.cuda.synchronize()
        assert torch.equal(ring[0, :, 0], sim.reward) and bool((ring[1:] == -5.0).all()), game
        sim.step_n(2)
        torch.cuda.synchronize()
        assert torch.equal(ring[1, :, 0], sim.reward) and bool((ring[2] == -5.0).all()), game
        sim.close()
