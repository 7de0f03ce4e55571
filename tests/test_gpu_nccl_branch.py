"""The `nccl` (= RCCL) branch of the torch.distributed sharding path on ONE GPU, before an 8-GPU node runs it for the first time:
a process group with backend nccl and world size 1, through which sharding.py's device-tensor code -- batch_isend_irecv of
frame slabs (gather_shards: two shards on the one rank, the sends to the own rank and their receives in one batch), the
all_gather_into_tensor of results, the all-gather fallback -- moves real tensors of the real simulator.  Each case runs in
a subprocess (one process group per process).  Reference scale-out being replaced: simulator_interface.cpp:270-283."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %(root)r)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    from xworld_amd import sharding
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", world_size=1, rank=0,
                            **sharding.nccl_init_kwargs(torch.device("cuda", 0)))
    from xworld_amd.batched import BatchedSimulator
    info = sharding.backend_info()
    assert info["backend"] == "nccl" and info["world_size"] == 1 and info["version"], info
    dist.barrier()                                                            # the communicator (and its streams) exist now
    conf = os.path.join(%(root)r, "xworld_amd", "confs", "navigation2d.json")
    opts = {"xwd_conf_path": conf, "task_mode": "lang_acquisition", "max_dim": 7, "color": True}
    counts = [300, 212]
    whole = BatchedSimulator("xworld", opts, num_envs=512, seed=8, policy_seed=3)
    # the FIRST batch created after the communicator: its internal queue must still run beside the caller's stream (xwb_create
    # re-selects the stream when the concurrency probe fails) -- every rank of a multi-GPU run is in this position
    if "XWB_QUEUE_SYNC" not in os.environ:
        assert whole.queue_sync_mode() == ("epochs", "probe_ok"), whole.queue_sync_mode()
    shards = [BatchedSimulator("xworld", opts, num_envs=300, seed=8, policy_seed=3, env_gid0=0),
              BatchedSimulator("xworld", opts, num_envs=212, seed=8, policy_seed=3, env_gid0=300)]
    dev = torch.device("cuda", 0)
    full = torch.zeros((512,) + tuple(whole.obs.shape[1:]), dtype=torch.uint8, device=dev)
    rg = sharding.ResultGather([512], 0, dev, force_collective=True)          # the all_gather_into_tensor of a world of one
    rec = torch.zeros((512, 2), device=dev)
    whole.bind_results(rec)
    p2p = 0
    for t in range(25):
        for s in [whole] + shards:
            s.step()
        rg.start(packed=rec)
        r, c = rg.finish()
        assert torch.equal(r, whole.reward) and torch.equal(c, whole.game_over_codes), t
        for s in [whole] + shards:
            s.reset_done()
        # device tensors through batch_isend_irecv: shard 1 -> the root's slice, both halves posted by this one rank
        work = sharding.gather_shards({0: shards[0].obs, 1: shards[1].obs}, full, counts, [0, 0], 0, dst=0, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert torch.equal(full, whole.obs), t
        p2p += 1
    assert not sharding._P2P_BROKEN, "RCCL refused the batched self send / recv"
    # the all-gather fallback (what a backend that refuses batched point-to-point would run), world of one
    out = torch.zeros_like(full)
    sharding._allgather_fallback(whole.obs, out, [512], 0, 0, None).wait()
    # (one rank: the root keeps its own slab through gather_slabs' copy; the fallback itself moves only remote slabs)
    assert sharding.gather_slabs(whole.obs, out, [512], 0) is out and torch.equal(out, whole.obs)
    # the class the bench drives, world of one
    sg = sharding.ScreensGather(whole, [512], 0)
    for t in range(4):
        sg.bind_next(); whole.step(); whole.reset_done(); sg.start()
    assert torch.equal(sg.drain(), whole.obs)
    gg = sharding.GridsGather(whole, [512], 0)
    whole.step(); whole.reset_done()
    assert torch.equal(gg(), whole.obs)
    for s in [whole] + shards:
        s.close()
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_BRANCH_OK", p2p, info["version"])
''')


def test_device_tensor_exchanges_over_nccl_world_of_one():
    import torch
    assert torch.cuda.is_available()
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "port": _free_port()}], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "NCCL_BRANCH_OK 25" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


@pytest.mark.parametrize("exchange", ["auto", pytest.param("torch", marks=pytest.mark.slow), pytest.param("lib", marks=pytest.mark.slow)])
def test_bench_forced_exchange_reports_rccl(exchange):
    """torchrun --nproc-per-node 1 bench.py --gpus 1 --backend nccl --force-exchange: the line says which RCCL ran."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--repeats", "3", "--spin-seconds", "0.02", "--envs-per-gpu", "2048", "--backend", "nccl", "--force-exchange",
           "--exchange", exchange, "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl"]["backend"] == "nccl" and line["rccl"]["world_size"] == 1 and line["rccl"]["version"]
    assert "forced_exchange" in line and line["parity"]["mismatches"] == 0
    # auto / lib: the per-step results run beside the step loop through the library's communicator, ordered by the step's epoch
    want = "torch.distributed" if exchange == "torch" else "libxwb.so"
    assert line["rccl"]["results_exchange"] == want, line["rccl"]
    assert (want in line["config"]["exchange"]) and (exchange == "torch" or "ordered by the step's e" in line["config"]["exchange"])
    sg = line["screens_gather"]
    assert sg["mode"] == "screens" and sg["value"] > 0 and "error" not in sg
    assert sg["grids"]["value"] > 0 and "error" not in sg["grids"], sg["grids"]
    assert line["path"]["path"] == "lazy_fused" and line["roofline"]["write_ceiling_GBps"] > 0
    assert set(line["roofline"]["kernels_us"]) >= {"render", "list"}        # (the step runs inside the render's launch)
