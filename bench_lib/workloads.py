"""Workloads of bench.py, their algorithmic bytes and the run's own bandwidth anchor."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
XGMI_LINK_GBS = 153.6       # one xGMI link, per direction (MI355X_MICROARCH.md); every remote shard has ONE link to the root

WORKLOADS = {
    # name: (game, opts, envs per GPU)
    "xworld7": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True}, 32768),
    "xworld7_f32": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "obs_format": "float32"}, 32768),
    "xworld7_ego3": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "visible_radius": 3}, 32768),
    "xworld8_ego5": ("xworld", {"color": True, "visible_radius": 5}, 32768),                       # 80x80x3 frames
    "xworld7_ego7": ("xworld", {"max_dim": 7, "num_blocks": 16, "color": True, "visible_radius": 7}, 32768),
    # a geometry the span path cannot take (81 x 81 frames: include/xwb.h xwb_ego_render_path): the one-workgroup-per-env kernel
    "xworld11_ego9": ("xworld", {"max_dim": 11, "num_blocks": 30, "color": True, "visible_radius": 9}, 32768),
    "xworld8": ("xworld", {"color": True}, 32768),
    "xworld11": ("xworld", {"max_dim": 11, "num_blocks": 30, "color": True}, 32768),
    "simple_game": ("simple_game", {"array_size": 64}, 65536),
    "simple_race": ("simple_race", {"track_width": 20.0, "track_length": 100.0, "track_radius": 30.0}, 65536),
}
POLICY_SEED = 0x5EED
REC_BYTES_CAP = 6 << 30      # the per-step (reward, game_over) record: a ring of at most this many bytes


def make_sim(workload, n_envs, device, gid0, seed=0xC0FFEE, **extra):
    from xworld_amd.batched import BatchedSimulator
    game, opts, _ = WORKLOADS[workload]
    opts = dict(opts)
    opts.update(extra)
    if game == "xworld":
        opts["xwd_conf_path"] = os.path.join(ROOT, "xworld_amd", "confs", "navigation2d.json")       # the five XWorld3DNav tasks
        opts["task_mode"] = "lang_acquisition"
    return BatchedSimulator(game, opts, num_envs=n_envs, device=device, env_gid0=gid0, seed=seed, policy_seed=POLICY_SEED)


def algorithmic_bytes(workload, sim):
    """logical bytes per env-step, per env per launch of the dominant kernel, and that kernel's name."""
    game = WORKLOADS[workload][0]
    if game == "simple_game":
        a = sim.cfg.array_size
        return 27 + a, 27 + a, "sg_kernel"
    if game == "simple_race":
        return 57, 57, "race_kernel"
    d = sim.cfg.max_dim
    c = sim.screen_dims[2]
    if sim.cfg.visible_radius:
        # egocentric: the frame is (r * (84 / r))^2 pixels; its algorithmic bytes are the frame written + the grid read.  The
        # whole-batch render is four launches on the span path (timed together, on the stream they run on: cell table,
        # evaluated pixels, terminal frames, gather -- the gather alone moves ~ all the bytes), one otherwise
        obs = c * sim.screen_dims[0] * sim.screen_dims[1]
        name = ("xw_ego_cells_kernel + xw_ego_eval_kernel + xw_ego_gather_list_kernel + xw_ego_gather_kernel"
                if sim.ego_render_path == "span" else "xw_render_ego_kernel")
        return 33 + 2 * d * d + obs, 2 * d * d + obs, name
    obs = c * 144 * d * d * (4 if sim.obs_is_float else 1)      # float32 variant: obs term x 4
    return 33 + 2 * d * d + obs, 2 * d * d + obs, "xw_render_all_kernel"


def dominant_kernel_name(base, path):
    """the kernel the roofline object is about, as this run's step path launched it"""
    if base == "xw_render_all_kernel" and path == "lazy_fused":
        return "xw_step_render_kernel (the whole-batch render with the step's blocks in the same launch)"
    return base


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, gfx950 corrections applied by tools/summarize_prof.py); None if not profiled.
    The number is a STORED measurement, not something this run measured: the line says which file, which commit and which
    source fingerprint it comes from, and `traffic_stale` when the sources this run executes are not those."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "traffic_%s.json" % workload)
        if os.path.exists(f):
            best = f
    if not best:
        return None, {"traffic_source": None}
    with open(best) as fh:
        t = json.load(fh)
    from xworld_amd import build
    now = build.source_fingerprint()
    return t["traffic_bytes_per_launch"], {"traffic_source": os.path.relpath(best, ROOT), "traffic_commit": t.get("commit", "unknown"),
                                           "traffic_source_sha16": t.get("source_sha16"), "source_sha16": now,
                                           "traffic_stale": t.get("source_sha16") != now}


def write_ceiling(buf, reps=24):
    """The bandwidth anchor of THIS run: a pure write stream of the observation batch's size (hipMemsetAsync through
    torch.Tensor.zero_, and a fill kernel), timed with events on the current stream, in this process, on this box -- what
    `roofline.achieved` can be read against besides the 8 TB/s spec.  It runs AFTER the timed and the event regions (round 4
    ran it between the spin and the timed regions: the first regions then paid for whatever it disturbed and the driver's
    20-step median landed in that ramp), on a buffer allocated before the warm-up and kept until the process ends."""
    import torch
    nbytes = buf.numel()
    out = {}
    for name, fn in (("memset", lambda: buf.zero_()), ("fill_kernel", lambda: buf.fill_(7))):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        out[name] = {"us": us, "GBps": nbytes / us / 1e3}
    return out
