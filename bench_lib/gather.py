"""N > 1: the step loop with every shard's frames gathered into one tensor on rank 0 (bench.py's `screens_gather` blocks)."""
import statistics

from .workloads import HBM_PEAK_GBS, XGMI_LINK_GBS

MODES = ("screens", "grids", "grids_nodraw")
LINE_KEY = {"grids": "grids", "grids_nodraw": "grids_no_local_render"}


def gather_block(sim, mode, lib_comm, depth, world, n_local, steps, sg_regions):
    """One screens_gather block from its timed regions: what crossed the links, the bound that applies, the ceilings."""
    shard_bytes = n_local * sim.obs_bytes_per_env
    sg_med = statistics.median(sg_regions)
    if mode == "screens":
        link_bytes = shard_bytes
        bound = {"bound": "xGMI link (every remote shard has ONE link to the root)"}
    else:
        mode_note = {"local_render": mode != "grids_nodraw"}
        link_bytes = n_local * (2 * sim.cfg.max_dim ** 2 + 1)
        # the root writes every frame of the whole batch once more: its HBM write stream bounds the step
        root_s = world * shard_bytes / (HBM_PEAK_GBS * 1e9)
        bound = {"bound": "the root's HBM write stream (it draws all %d frames from the gathered cell codes)" % (n_local * world),
                 "root_render_bound_ms_per_step": root_s * 1e3, "root_render_bound_ceiling": n_local * world / root_s, **mode_note}
    link_s = link_bytes / (XGMI_LINK_GBS * 1e9)
    if lib_comm is not None:
        by = ("libxwb.so (xwb_gather_%s_begin + xwb_comm_mark / _wait: ncclSend / ncclRecv on the communicator's stream)"
              % mode.split("_")[0])
    elif mode.startswith("grids"):
        by = "torch.distributed batch_isend_irecv of the packed cell codes + xwb_xw_render_grids on the root (sharding.GridsGather)"
    else:
        by = "torch.distributed batch_isend_irecv"
    overlap = ("double-buffered: transfer of step t beside the kernels of step t+1" if depth == 2 else
               ("none (context ring)" if sim.cfg.context > 1 else "none (the few MB of cell codes are gathered synchronously)"))
    return {"mode": mode, "ms_per_step": sg_med / steps * 1e3, "value": n_local * world * steps / sg_med, "unit": "env-steps/s",
            "bytes_into_root_per_step": link_bytes * (world - 1), "link_bound_ms_per_step": link_s * 1e3,
            "link_bound_ceiling": n_local * world / link_s, "link_GBps_assumed": XGMI_LINK_GBS,
            "achieved_GBps_per_link": link_bytes / (sg_med / steps) / 1e9, "overlap": overlap, "issued_by": by,
            "regions_ms_per_step": {"min": min(sg_regions) / steps * 1e3, "max": max(sg_regions) / steps * 1e3}, **bound}


def make_gather(sim, mode, lib_comm, counts, rank):
    """mode: screens | grids | grids_nodraw (= grids with every shard's own pixel stores off: xwb_xw_set_draw(sim, 0))"""
    from xworld_amd import sharding
    m = "grids" if mode.startswith("grids") else "screens"
    if mode.startswith("grids"):
        sim.set_draw(mode != "grids_nodraw")
    if lib_comm is not None:
        return sharding.LibScreensGather(sim, lib_comm, counts, rank, mode=m)
    return sharding.GridsGather(sim, counts, rank) if m == "grids" else sharding.ScreensGather(sim, counts, rank)


def redraw_own_frames(sim, n_local):
    """after grids_nodraw: back to a batch that draws, its own buffer made current from its draw state"""
    import torch
    sim.set_draw(True)
    d = sim.cfg.max_dim
    gr = torch.empty((n_local, d * d), dtype=torch.int16, device=sim.obs.device)
    fl = torch.empty((n_local,), dtype=torch.uint8, device=sim.obs.device)
    sim.pack_grids(gr, fl)
    sim.render_grids(gr, fl, sim.obs)


def gather_regions(sim, mode, lib_comm, counts, rank, world, n_local, K, R, steps, set_screens, one_step, timed_region, fence):
    """R timed regions of the loop with every step's frames gathered on rank 0 by `mode`; returns the block."""
    g = make_gather(sim, mode, lib_comm, counts, rank)
    set_screens(g)
    try:
        for _ in range(2 * K):
            one_step()
        regs = [timed_region() for _ in range(R)]
        fence()
    finally:
        set_screens(None)
        if mode == "grids_nodraw":
            redraw_own_frames(sim, n_local)
    return gather_block(sim, mode, lib_comm, g.depth, world, n_local, steps, regs)


def modes_for(gather_arg, grids_ok):
    return [m for m in (MODES if gather_arg == "both" else (gather_arg,)) if m == "screens" or grids_ok]


def merge_blocks(blocks, modes):
    """the line's `screens_gather` object: the first mode's block with the others as sub-objects"""
    out = dict(blocks[modes[0]], mode=modes[0])
    for m in modes[1:]:
        if m in blocks:
            out[LINE_KEY.get(m, m)] = blocks[m]
    return out
