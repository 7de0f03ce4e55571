"""The N > 1 path with the REAL simulator on one GPU: 2 (and 3, ragged) ranks on cuda:0 over gloo (RCCL refuses two ranks
on one device; sharding.py stages gloo's device tensors through host memory, everything else -- shard ranges, global env
ids, bind_results_ring -> ResultGather, bind_obs slices -> ScreensGather double buffering -- is the code the 8-GPU run
executes).  Rank 0 also holds the unsharded batch and compares step for step (reference scale-out being replaced:
examples/demo_interface.cpp:67-95, simulator_interface.cpp:270-283,361-383)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONF = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")

GAMES = {
    "xworld7": ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16, "color": True}),
    "xworld7_ego3": ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "max_dim": 7, "num_blocks": 16,
                                "color": True, "visible_radius": 3}),
    "simple_game": ("simple_game", {"array_size": 16}),
    "simple_race": ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0, "random": True}),
    "xworld8_ctx2": ("xworld", {"xwd_conf_path": CONF, "task_mode": "lang_acquisition", "context": 2}),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, game_key, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xworld_amd import sharding
        from xworld_amd.batched import BatchedSimulator
        name, opts = GAMES[game_key]
        dev = torch.device("cuda", 0)
        counts = sharding.shard_counts(total, world)
        start, n = sharding.shard_range(total, world, rank)
        sim = BatchedSimulator(name, opts, num_envs=n, env_gid0=start, seed=77, policy_seed=9)
        full = BatchedSimulator(name, opts, num_envs=total, env_gid0=0, seed=77, policy_seed=9) if rank == 0 else None
        rec = torch.zeros((4, n, 2), dtype=torch.float32, device=dev)
        sim.bind_results_ring(rec)
        results = sharding.ResultGather(counts, rank, dev)
        screens = sharding.ScreensGather(sim, counts, rank)
        assert screens.depth == (2 if sim.cfg.context == 1 else 1)
        assert sharding.backend_info() == {"world_size": world, "backend": "gloo", "version": None}
        expect = []                                   # rank 0: (reward, codes, screens) of the unsharded batch per step
        for t in range(steps):
            screens.bind_next()
            sim.step()
            prev = results.finish()                   # the gather of step t - 1 ran beside this step
            results.start(packed=rec[t % 4])
            sim.reset_done()
            screens.start()
            if rank == 0:
                full.step()
                r, c = full.reward.clone(), full.game_over_codes.clone()
                full.reset_done()
                expect.append((r, c, full.obs.clone()))
                if t > 0:
                    assert torch.equal(prev[0], expect[t - 1][0]) and torch.equal(prev[1], expect[t - 1][1]), t
                    got = screens.latest()            # the previous step's gathered screens, complete by now
                    if screens.depth == 2:
                        assert got is not None and got.is_contiguous() and torch.equal(got, expect[t - 1][2]), t
                    else:
                        assert torch.equal(got, expect[t][2]), t
        last = results.finish()
        final = screens.drain()
        if rank == 0:
            assert torch.equal(last[0], expect[-1][0]) and torch.equal(last[1], expect[-1][1])
            assert final.shape[0] == total and torch.equal(final, expect[-1][2])
            assert int(sum(int((e[1] != 0).sum()) for e in expect)) > 0 or game_key == "xworld8_ctx2"     # episodes did end
            full.close()
        else:
            assert last == (None, None) and final is None
        sim.close()
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:                               # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,game,steps", [(2, 192, "xworld7", 60), (3, 200, "xworld7", 40), (2, 128, "xworld7_ego3", 30),
                                                    (2, 256, "simple_game", 40), (3, 100, "simple_race", 80),
                                                    (2, 64, "xworld8_ctx2", 12)])
def test_sharded_equals_unsharded(world, total, game, steps):
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, game, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.parametrize("workload,extra", [("xworld7", []), pytest.param("simple_game", [], marks=pytest.mark.slow),
                                            ("xworld7", ["--autoreset"]), ("xworld7", ["--c5"])])
def test_bench_two_ranks_on_one_gpu(workload, extra):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run), on one GPU over gloo: the line carries the
    device-resident value, the screens-gather figures, what the exchange ran on, and a clean parity gate."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--repeats", "3", "--spin-seconds", "0.02", "--envs-per-gpu", "1024", "--backend", "gloo", "--workload", workload] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["total_envs"] == 2048 and line["value"] > 0
    rc = line["rccl"]
    assert (rc["world_size"], rc["backend"], rc["version"]) == (2, "gloo", None)
    # the record itself shows which rank ran on which device (here: two ranks sharing the box's one GPU)
    assert rc["ranks_seen"] == 2 and rc["devices_seen"] == 1 and [r["rank"] for r in rc["ranks"]] == [0, 1]
    assert "watchdog" not in line and "screens_gather:screens" in line["phase_seconds"]
    sg = line["screens_gather"]
    assert sg["value"] > 0 and sg["link_bound_ceiling"] > 0 and sg["bytes_into_root_per_step"] > 0 and sg["mode"] == "screens"
    if workload == "xworld7":                             # full observation: the same loop with the cell codes gathered instead
        gg = sg["grids"]
        assert "error" not in gg, gg
        assert gg["mode"] == "grids" and gg["value"] > 0 and gg["root_render_bound_ceiling"] > 0
        assert gg["bytes_into_root_per_step"] == 1024 * (2 * 49 + 1) and gg["bytes_into_root_per_step"] * 200 < sg["bytes_into_root_per_step"]
        nd = sg["grids_no_local_render"]                   # the same with every shard's own pixel stores off (xwb_xw_set_draw)
        assert "error" not in nd and nd["value"] > 0 and nd["local_render"] is False and gg["local_render"] is True
    else:
        assert "skipped" in sg["grids"]
    assert line["regions"]["repetitions"] == 3 and len(line["regions"]["ms_per_step_all"]) == 3
    assert line["parity"]["mismatches"] == 0 and line["parity"]["checked_env_steps"] > 0
    assert "cpu_baseline" not in line                     # rank 0 at N = 1 only
    if "--c5" in extra:                                   # BASELINE C5 (xworld11) as a block of the same line
        c5 = line["c5"]
        assert "error" not in c5, c5
        assert c5["workload"] == "xworld11" and c5["value"] > 0 and c5["action_errors"] == 0
        assert c5["screens_gather"]["value"] > 0 and c5["screens_gather"]["link_bound_ceiling"] > 0
        assert c5["screens_gather"]["grids"]["value"] > 0 and "error" not in c5["screens_gather"]["grids"]
    else:
        assert "c5" not in line
