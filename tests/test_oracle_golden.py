"""The CPU oracle (oracle/cista_oracle.py) against golden vectors produced by running the reference itself
(tools/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare HIP vs oracle/goldens.
Tolerance: 1e-5 relative to tensor scale (same fp32 arithmetic, different summation order only)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu          # noqa: E402
import weights_util as wu         # noqa: E402
from oracle import cista_oracle as orc   # noqa: E402

TOL = 1e-5


def test_warp_golden():
    g = gu.load("warp.npz")
    for i in range(4):
        img, flow = torch.from_numpy(g["img_%d" % i]), torch.from_numpy(g["flow_%d" % i])
        mode = "forward" if int(g["mode_%d" % i][0]) == 0 else "backward"
        out = orc.warp(img, flow, mode)
        assert gu.rel_err(out, g["out_%d" % i]) < TOL


def test_zero_flow_warp_is_not_identity():
    g = gu.load("warp.npz")
    img = torch.from_numpy(g["img_3"])
    out = torch.from_numpy(g["out_3"])      # reference output for an all-zero flow
    assert (out - img).abs().max() > 0.1    # the W-not-(W-1) quirk (flow_utils.py:114-115)


def test_cista_golden():
    g = gu.load("cista_36x52.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    sd = wu.make_state_dict(gu.layout("cista_state_dict_layout.json"), seed)
    states, prev = None, torch.zeros(B, 1, H, W)
    for t in range(frames):
        ev = torch.from_numpy(g["ev_%d" % t])
        assert torch.equal(ev, wu.synth_events(B, 5, H, W, seed * 1000 + t))
        I, states = orc.cista_forward(sd, ev, prev, states, prefix="")
        assert gu.rel_err(I, g["I_%d" % t]) < TOL
        assert gu.rel_err(gu.sub(states[0], 2, 1, 2), g["c_%d" % t]) < TOL
        assert gu.rel_err(gu.sub(states[1], 2, 1, 2), g["z_%d" % t]) < TOL
        assert gu.rel_err(gu.sub(states[2][0], 2, 1, 2), g["h_%d" % t]) < TOL
        assert gu.rel_err(gu.sub(states[2][1], 2, 1, 2), g["cc_%d" % t]) < TOL
        prev = I.clone()


@pytest.mark.parametrize("name,mode", [("eiflow_100x124.npz", "forward"), ("eiflow_128x136_bw.npz", "backward"),
                                       ("eiflow_180x240.npz", "forward")])
def test_eiflow_golden(name, mode):
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    sd = wu.make_state_dict(gu.layout("eiflow_state_dict_layout.json"), seed)
    states, prev = None, torch.zeros(B, 1, H, W)
    for t in range(frames):
        ev = torch.from_numpy(g["ev_%d" % t])
        I, bf, states = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev}, states, warp_mode=mode)
        # flow through 6 GRU iterations and the recurrence: 5e-5 of the tensor scale
        assert gu.rel_err(bf["flow_final"], g["flow_%d" % t]) < 5e-5, t
        assert gu.rel_err(bf["flow_init"], g["flowlow_%d" % t]) < 5e-5, t
        assert gu.rel_err(I, g["I_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[1]), g["z_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[0]), g["c_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[2][0]), g["h_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[2][1]), g["cc_%d" % t]) < 5e-5, t
        if "preds0_%d" % t in g:
            assert gu.rel_err(bf["flow_preds"][0], g["preds0_%d" % t]) < 5e-5
        prev = I.clone()


def test_eraft_golden():
    g = gu.load("eraft_100x124.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    sd = wu.make_state_dict(gu.layout("eraft_state_dict_layout.json"), seed)
    states, prev, ev_old = None, torch.zeros(B, 1, H, W), None
    for t in range(frames):
        ev = torch.from_numpy(g["ev_%d" % t])
        if ev_old is None:
            ev_old = wu.synth_events(B, 5, H, W, seed * 1000 + 999)   # see tools/gen_golden.py::run_eraft
        I, bf, states = orc.eraft_step(sd, {"event_voxel": ev, "event_voxel_old": ev_old, "rec_img0": prev}, states)
        ev_old = ev.clone()
        assert gu.rel_err(bf["flow_final"], g["flow_%d" % t]) < 5e-5, t
        assert gu.rel_err(bf["flow_init"], g["flowlow_%d" % t]) < 5e-5, t
        assert gu.rel_err(I, g["I_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[1]), g["z_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[0]), g["c_%d" % t]) < 5e-5, t
        if t == 1:
            assert len(bf["flow_preds"]) == 12
            assert gu.rel_err(bf["flow_preds"][0], g["preds0_1"]) < 5e-5
            assert gu.rel_err(bf["flow_preds"][6], g["preds6_1"]) < 5e-5
        prev = I.clone()


@pytest.mark.parametrize("name", ["idnet_68x92.npz", "idnet_260x346.npz"])
def test_idnet_golden(name):
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    sd = wu.make_state_dict(gu.layout("idnet_state_dict_layout.json"), seed)
    states, prev, flow_init = None, torch.zeros(B, 1, H, W), None
    for t in range(frames):
        ev = torch.from_numpy(g["ev_%d" % t])
        I, bf, states = orc.idnet_step(sd, {"event_voxel": ev, "rec_img0": prev}, states, flow_init)
        flow_init = bf["next_flow"]
        assert gu.rel_err(bf["flow_final"], g["flow_%d" % t]) < 5e-5, t
        st = 3 if H >= 200 else 1
        assert gu.rel_err(bf["next_flow"][..., ::st, ::st], g["next_%d" % t]) < 5e-5, t
        assert gu.rel_err(bf["delta_flow"][:, 1][..., ::st, ::st], g["delta_%d" % t]) < 5e-5, t
        assert gu.rel_err(I, g["I_%d" % t]) < 5e-5, t
        assert gu.rel_err(gu.sub(states[1]), g["z_%d" % t]) < 5e-5, t
        prev = I.clone()


def test_events_to_voxel_golden():
    g = gu.load("events.npz")
    for i in range(4):
        H, W = [int(v) for v in g["dims_%d" % i]]
        raw = orc.events_to_voxel(g["ev_%d" % i].copy(), 5, W, H, normalize=False)
        nrm = orc.events_to_voxel(g["ev_%d" % i].copy(), 5, W, H, normalize=True)
        assert gu.rel_err(raw, g["raw_%d" % i]) < 1e-6 or (abs(g["raw_%d" % i]).max() == 0 and abs(raw).max() == 0)
        assert gu.rel_err(nrm, g["norm_%d" % i]) < 1e-5 or (abs(g["norm_%d" % i]).max() == 0 and abs(nrm).max() == 0)


def test_state_dict_layout_matches_reference():
    """The shell modules must expose the reference's state_dict keys, shapes and order (262 entries)."""
    import argparse
    from cista_flow_amd.e2v.e2v_model import CistaLSTCNet, DCEIFlowCistaNet, ERAFTCistaNet, IDCistaNet
    a = argparse.Namespace(image_dim=[180, 240], num_bins=5, warp_mode='forward', base_channels=64, depth=5, ds=8, is_bi=False)
    m = DCEIFlowCistaNet(a)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == gu.layout("eiflow_state_dict_layout.json")
    c = CistaLSTCNet([180, 240])
    assert [(k, tuple(v.shape)) for k, v in c.state_dict().items()] == gu.layout("cista_state_dict_layout.json")
    e = ERAFTCistaNet(a)
    assert [(k, tuple(v.shape)) for k, v in e.state_dict().items()] == gu.layout("eraft_state_dict_layout.json")
    d = IDCistaNet(a)
    assert [(k, tuple(v.shape)) for k, v in d.state_dict().items()] == gu.layout("idnet_state_dict_layout.json")
    # the five lista blocks alias one storage and survive a strict load
    sd = wu.make_state_dict(gu.layout("eiflow_state_dict_layout.json"), 5)
    m.load_state_dict(sd, strict=True)
    blocks = m.cista_net.lista_blocks
    assert all(blocks[i].Lambda.data_ptr() == blocks[0].Lambda.data_ptr() for i in range(5))


def test_metrics_golden():
    """f-3: the oracle's restatement of loss.py against values the reference itself produced (tools/gen_golden.py)."""
    g = gu.load("metrics.npz")
    mse, psnr = orc.recon_metrics(torch.from_numpy(g["rec"]), torch.from_numpy(g["tgt"]))
    assert abs(mse - float(g["mse"])) < 1e-6 * float(g["mse"]) + 1e-12
    assert abs(psnr - float(g["psnr"])) < 1e-4
    assert orc.recon_metrics(torch.from_numpy(g["rec"]), torch.from_numpy(g["rec"]))[1] == float(g["psnr_same"]) == 100.0
    for mode in ("forward", "backward"):
        t = {k: torch.from_numpy(g[k + "_" + mode]) for k in ("flow", "gt", "img0", "img1", "valid")}
        m1 = orc.flow_metrics(t["flow"], t["gt"], t["img0"], t["img1"], t["valid"], mode)
        m2 = orc.flow_metrics(t["flow"], t["gt"], t["img0"], t["img1"], None, mode)
        for got, ref in ((m1, g["fm_valid_" + mode]), (m2, g["fm_photo_" + mode])):
            for a, b in zip(got, ref):
                assert abs(a - float(b)) <= 2e-5 * abs(float(b)) + 1e-7, (mode, got, ref)
    v1 = orc.voxel_warping_flow_loss(torch.from_numpy(g["fwl_evs"]), torch.from_numpy(g["fwl_flow"]))
    v0 = orc.voxel_warping_flow_loss(torch.from_numpy(g["fwl_evs"]), torch.zeros_like(torch.from_numpy(g["fwl_flow"])))
    assert abs(v1 - g["fwl"][0]) < 1e-5 * g["fwl"][0] and abs(v0 - g["fwl"][1]) < 1e-5 * g["fwl"][1]
    assert abs(v1 / v0 - g["fwl"][2]) < 1e-5


def test_fullstate_golden():
    """The un-strided sparse code of the eiflow_100x124 sequence (every element of states[1] after 4 frames)."""
    g = gu.load("eiflow_100x124_fullstate.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    sd = wu.make_state_dict(gu.layout("eiflow_state_dict_layout.json"), seed)
    states, prev = None, torch.zeros(B, 1, H, W)
    for t in range(frames):
        ev = wu.synth_events(B, 5, H, W, seed * 1000 + t)
        I, bf, states = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev}, states)
        prev = I.clone()
    assert gu.rel_err(states[1], g["z_full"]) < 5e-5
    assert gu.rel_err(states[2][0], g["h_full"]) < 5e-5


FULLSIZE = [("eraft_180x240.npz", "eraft"), ("eiflow_480x640.npz", "eiflow")]


def fullsize_probe(g, t, I, bf, states):
    """(name, got, ref) triples of one frame of a tools/gen_golden.py::run_fullsize fixture (strided probes)."""
    st, cs = int(g["meta"][5]), int(g["meta"][6])
    return [("I", I[..., ::st, ::st], g["I_%d" % t]), ("flow", bf["flow_final"][..., ::st, ::st], g["flow_%d" % t]),
            ("flowlow", bf["flow_init"], g["flowlow_%d" % t]), ("preds0", bf["flow_preds"][0][..., ::2 * st, ::2 * st], g["preds0_%d" % t]),
            ("c", gu.sub(states[0], cs, 2 * st, 2 * st), g["c_%d" % t]), ("z", gu.sub(states[1], cs, 2 * st, 2 * st), g["z_%d" % t]),
            ("h", gu.sub(states[2][0], cs, 2 * st, 2 * st), g["h_%d" % t]), ("cc", gu.sub(states[2][1], cs, 2 * st, 2 * st), g["cc_%d" % t])]


@pytest.mark.parametrize("name,kind", FULLSIZE)
def test_fullsize_reference_goldens(name, kind):
    """The oracle against reference-run fixtures at the full size of BASELINE configs[2] (eraft 180x240) and configs[3]
    (eiflow 480x640), B = 1, two recurrent frames (round 3)."""
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"][:5]]
    sd = wu.make_state_dict(gu.layout("%s_state_dict_layout.json" % kind), seed)
    states, prev, old = None, torch.zeros(B, 1, H, W), wu.synth_events(B, 5, H, W, seed * 1000 + 999)
    for t in range(frames):
        ev = wu.synth_events(B, 5, H, W, seed * 1000 + t)
        if kind == "eraft":
            I, bf, states = orc.eraft_step(sd, {"event_voxel": ev, "event_voxel_old": old, "rec_img0": prev}, states)
        else:
            I, bf, states = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev}, states)
        old = ev
        for nm, got, ref in fullsize_probe(g, t, I, bf, states):
            # 1e-4: the first iteration's flow at 480x640 (small values, 4800-pixel correlation rows) sits at 5.4e-5 of its scale
            assert gu.rel_err(got, ref) < 1e-4, (t, nm)
        prev = I.clone()
