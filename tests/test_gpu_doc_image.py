"""The HIP egocentric render against the reference's own rendered frame (tests/golden/xworld2d_doc.png = the reference's
doc/xworld2d.png, see tests/test_oracle_doc_image.py): the map read off that image is loaded through the C ABI
(xwb_xw_load_map, xwb_xw_set_agent_dir, xwb_xw_set_goal_pose with the fitted poses) and the frame the kernels draw must
equal, byte for byte, the reference's view pixels pushed through XWorldSimulator's two resizes -- and the oracle's frame.
What this pins for the product: atlas decode, XItem::get_item_image's warp at four poses, XMap::image_masking, the
black / white fill, crop and view rotation.  The resizes themselves are the oracle's restatement on both sides."""
import os

import numpy as np
import pytest

from test_gpu_xworld import _torch
from test_oracle_doc_image import AGENT, BLACK, R, doc_fit, doc_world, frame_from_doc

pytestmark = pytest.mark.gpu

CONF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xworld_amd", "confs")


@pytest.mark.parametrize("dim,agent,color", [(5, (2, 4), True), (8, (3, 6), True), (5, (2, 4), False), (8, (3, 6), False)])
def test_product_frame_equals_reference_frame(oracle, dim, agent, color):
    _torch()
    from xworld_amd.batched import BatchedSimulator
    n = 4
    sim = BatchedSimulator("xworld", {"xwd_conf_path": os.path.join(CONF, "nav_target.json"), "max_dim": dim, "dim": dim,
                                      "task_mode": "lang_acquisition", "tasks": ["XWorld3DNavTarget"],
                                      "visible_radius": R, "color": color, "num_goals": 3, "num_blocks": 4}, num_envs=n)
    assert sim.screen_dims == (80, 80, 3 if color else 1)
    w, ents, poses = doc_world(oracle, dim, agent, color=int(color))
    pal = w.pal
    assert [m["path"] for m in pal.meta] == [m["path"] for m in sim.palette.meta]
    g = np.zeros((dim, dim), np.uint16)
    for t, x, y, icon, name, serial in ents:
        g[y, x] = icon + 1
    tc = w.target_cells()
    g[tc != 0] |= 0x8000
    for e in (1, 3):                                                 # two slots of the batch; the others keep their own maps
        sim.load_map(e, g, agent[0], agent[1], dim=dim, task="XWorld3DNavTarget", target=w.target_name())
        sim.set_agent_dir(e, 3)                                      # heading up
        for (t, x, y, icon, name, serial), (yaw, scale, offset) in zip(ents, poses):
            if t == 0:
                sim.set_goal_pose(e, x, y, yaw, scale, offset)
        sim.refresh_obs(e)
    obs = sim.obs.cpu().numpy()
    ref = frame_from_doc(oracle, dim * 64, color)                     # from the reference's pixels
    assert ref.shape == obs[1].shape
    for e in (1, 3):
        assert np.array_equal(obs[e], w.screen()), (e, int((obs[e] != w.screen()).sum()))
        assert np.array_equal(obs[e], ref), (e, int((obs[e] != ref).sum()))
    # the black cells of the image are black in the frame (16-pixel cells at 80 x 80)
    for r, c in BLACK:
        assert (obs[1][:, r * 16:(r + 1) * 16, c * 16:(c + 1) * 16] == 0).all()
    assert len(doc_fit()["cells"]) == 3 and AGENT == (4, 2)
    # one step forward is blocked by nothing: the agent moves up and the frame still equals the oracle's
    import torch
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    acts[1] = 0
    sim.step(acts)
    w.take_actions(0)
    assert np.array_equal(sim.obs.cpu().numpy()[1], w.screen())
    sim.close()
