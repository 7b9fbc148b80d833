"""One pass of the drivers' per-frame loop (test_with_flow.py:120-186) chained END TO END through this build -- event-file reader ->
voxel grid + normalisation on the GPU -> DCEIFlowCistaNet with the fed-back reconstruction and carried states -> uint8 quantisation ->
evaluation metrics -- against the SAME chain run with the reference's own classes (tools/gen_golden.py::run_chain ->
tests/golden/chain_eiflow_100x124.npz; VERDICT r3 missing 3).  Every stage is pinned on its own elsewhere; this pins the plumbing
between them: window boundaries, grid layout, dtype hand-offs, the clone() feedback, the uint8 cast, the metric inputs."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_util as gu
import weights_util as wu

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _targets(seed, H, W, frames):
    """tools/gen_golden.py::chain_targets, restated (the generator imports the reference, which does not travel)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    prev = torch.rand(1, 1, H, W, generator=g)
    for _ in range(frames):
        cur = (prev + 0.08 * torch.randn(1, 1, H, W, generator=g)).clamp(0, 1)
        flow = 2.5 * torch.randn(1, 2, H, W, generator=g)
        flow[0, :, 3:6, 10:30] = 450.0
        flow[0, :, 40:50, 60:70] *= 0.01
        out.append((prev, cur, flow))
        prev = cur
    return out


def test_driver_loop_chain_matches_the_reference_chain(gpu, tmp_path):
    import argparse
    from cista_flow_amd.data_readers.event_readers import FixedSizeEventReader
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
    from cista_flow_amd.loss import FlowL1LossDict, ReconLoss
    from cista_flow_amd.utils.event_process import events_to_voxel_grid_batch
    from cista_flow_amd.utils.flow_utils import FrameWarp
    from cista_flow_amd.utils.image_process import to_uint8
    g = gu.load("chain_eiflow_100x124.npz")
    H, W, bins, frames, nev, wseed, eseed, tseed = [int(v) for v in g["meta"]]
    path = str(tmp_path / "events.txt")
    wu.synth_event_file(path, seed=eseed, n=frames * nev, width=W, height=H, duration=0.4, overshoot=False)
    a = argparse.Namespace(image_dim=[H, W], num_bins=bins, warp_mode="forward", base_channels=64, depth=5, ds=8, is_bi=False)
    m = DCEIFlowCistaNet(a).eval()
    wu.fill_module(m, wseed)
    m = m.to(gpu)
    fw = FrameWarp(mode="forward")
    rec_fn, flow_fn = ReconLoss(fw), FlowL1LossDict([H, W], fw)
    targets = _targets(tseed, H, W, frames)
    reader = iter(FixedSizeEventReader(path, num_events=nev))
    states, prev_image = None, torch.zeros(1, 1, H, W, device=gpu)
    lsb_pixels = 0
    with torch.no_grad():
        for t in range(frames):
            window = np.asarray(next(reader), dtype=np.float64)
            assert np.allclose([len(window), window[0, 0], window[-1, 0]], g["nev_%d" % t], rtol=0, atol=1e-12)     # same window
            evs = events_to_voxel_grid_batch([torch.from_numpy(window).to(gpu)], bins, W, H, normalize=True, filter_hot_pixel=False)
            assert gu.rel_err(evs[0, :, ::4, ::4].cpu(), g["grid_%d" % t]) < 1e-5
            pred, batch_flow, states = m({"event_voxel": evs, "rec_img0": prev_image}, states, {})
            prev_image = pred.clone()
            assert gu.rel_err(pred.squeeze().cpu()[::3, ::3], g["pred_%d" % t]) < 2e-4, t
            # np.uint8(pred * 255.): byte-equal, except where pred * 255 sits within float noise of an integer (counted below)
            u8 = to_uint8(pred).squeeze().cpu().numpy()
            ref = g["u8_%d" % t]
            d = np.abs(u8.astype(int) - ref.astype(int))
            assert d.max() <= 1, (t, d.max())
            lsb_pixels += int((d > 0).sum())
            gt0, gt1, gtf = [x.to(gpu) for x in targets[t]]
            rec_m = rec_fn.evaluate(pred, gt1)
            flow_m = flow_fn.evaluate(batch_flow["flow_final"], dict(gt_img0=gt0, gt_img1=gt1, gt_flow=gtf))
            assert abs(rec_m["mse"] - g["rec_%d" % t][0]) <= 2e-5 * abs(g["rec_%d" % t][0]), (t, rec_m)
            assert abs(rec_m["psnr"] - g["rec_%d" % t][1]) <= 2e-5 * abs(g["rec_%d" % t][1]), (t, rec_m)
            fm = g["flowm_%d" % t]
            for i, k in enumerate(("photo_loss", "epe")):                       # continuous in the model's flow (itself 1e-5 off)
                assert abs(flow_m[k] - fm[i]) <= 1e-4 * abs(fm[i]), (t, k, flow_m[k], fm[i])
            for i, k in ((2, "1px"), (3, "3px"), (4, "5px")):                   # threshold counts: a pixel on the threshold may flip
                assert abs(flow_m[k] - fm[i]) <= 3.0 / (H * W), (t, k, flow_m[k], fm[i])
            assert abs(flow_m["out"] - fm[5]) <= 300.0 / (H * W), (t, flow_m["out"], fm[5])
    # over 4 x 12,400 pixels a handful of reconstructions land on a k / 255 boundary
    assert lsb_pixels <= 40, lsb_pixels
