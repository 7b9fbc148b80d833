"""Model-level parity on the GPU: the HIP path behind the reference's module API against
  (1) golden vectors produced by running the reference itself (tests/golden, tools/gen_golden.py) and
  (2) the CPU oracle on fresh seeded inputs.
Bar (BASELINE.json north_star): 1e-3 relative fp32; observed errors are ~1e-5, asserted at 2e-4 so a
regression in any fused kernel trips the test long before the contractual bar."""
import argparse
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu          # noqa: E402
import weights_util as wu         # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 2e-4


def args_for(H, W, warp_mode="forward"):
    return argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode=warp_mode, base_channels=64, depth=5, ds=8,
                              is_bi=False)


def build_eiflow(H, W, seed, gpu, warp_mode="forward"):
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
    m = DCEIFlowCistaNet(args_for(H, W, warp_mode)).eval()
    wu.fill_module(m, seed)
    return m.to(gpu)


def test_native_library_is_loaded(gpu):
    from cista_flow_amd import lib
    lib.load()
    maps = open("/proc/self/maps").read()
    assert "libcistaflow.so" in maps


def test_framewarp_golden(gpu):
    from cista_flow_amd.utils.flow_utils import FrameWarp
    g = gu.load("warp.npz")
    for i in range(4):
        mode = "forward" if int(g["mode_%d" % i][0]) == 0 else "backward"
        out = FrameWarp(mode).warp_frame(torch.from_numpy(g["img_%d" % i]).to(gpu), torch.from_numpy(g["flow_%d" % i]).to(gpu))
        assert out.shape == g["out_%d" % i].shape
        assert gu.rel_err(out.cpu(), g["out_%d" % i]) < 2e-5, i


def test_cista_golden(gpu):
    from cista_flow_amd.e2v.e2v_model import CistaLSTCNet
    g = gu.load("cista_36x52.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    net = CistaLSTCNet([H, W]).eval()
    wu.fill_module(net, seed)
    net = net.to(gpu)
    states, prev = None, torch.zeros(B, 1, H, W, device=gpu)
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            I, states = net(ev, prev, states)
            assert I.shape == (B, 1, H, W) and states[1].shape == (B, 128, H // 2, W // 2)
            assert gu.rel_err(I.cpu(), g["I_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[0].cpu(), 2, 1, 2), g["c_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[1].cpu(), 2, 1, 2), g["z_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][0].cpu(), 2, 1, 2), g["h_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][1].cpu(), 2, 1, 2), g["cc_%d" % t]) < TOL, t
            prev = I.clone()


@pytest.mark.parametrize("name,mode", [("eiflow_100x124.npz", "forward"), ("eiflow_128x136_bw.npz", "backward"),
                                       ("eiflow_180x240.npz", "forward")])
def test_eiflow_golden_sequence(gpu, name, mode):
    """The driver loop of test_with_flow.py:120-156 (zeros prev image, states=None, feedback of the prediction)."""
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    m = build_eiflow(H, W, seed, gpu, mode)
    states, prev = None, torch.zeros(B, 1, H, W, device=gpu)
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, {})
            assert bf["flow_final"].shape == (B, 2, H, W)
            assert gu.rel_err(bf["flow_final"].cpu(), g["flow_%d" % t]) < TOL, t
            assert gu.rel_err(bf["flow_init"].cpu(), g["flowlow_%d" % t]) < TOL, t
            assert gu.rel_err(I.cpu(), g["I_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[0].cpu()), g["c_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[1].cpu()), g["z_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][0].cpu()), g["h_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][1].cpu()), g["cc_%d" % t]) < TOL, t
            if "preds0_%d" % t in g:
                assert len(bf["flow_preds"]) == 6
                assert gu.rel_err(bf["flow_preds"][0].cpu(), g["preds0_%d" % t]) < TOL
            prev = I.clone()


def test_eiflow_batch8_reference_golden(gpu):
    """Every slot of a B = 8 batch of DIFFERENT sequences (the batch of BASELINE configs[1]) against the reference itself, three
    recurrent frames at 100x124 (tools/gen_golden.py::run_eiflow_batch8; strided probes, inputs regenerated from the seed)."""
    g = gu.load("eiflow_100x124_B8.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    assert B == 8
    m = build_eiflow(H, W, seed, gpu, "forward")
    states, prev = None, torch.zeros(B, 1, H, W, device=gpu)
    with torch.no_grad():
        for t in range(frames):
            ev = wu.synth_events(B, 5, H, W, seed * 1000 + t).to(gpu)
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, {})
            for b in range(B):          # slot by slot: an error confined to one slot cannot hide in the batch's scale
                assert gu.rel_err(I.cpu()[b:b + 1, :, ::2, ::2], g["I_%d" % t][b:b + 1]) < TOL, (t, b)
                assert gu.rel_err(bf["flow_final"].cpu()[b:b + 1, :, ::2, ::2], g["flow_%d" % t][b:b + 1]) < TOL, (t, b)
                assert gu.rel_err(gu.sub(states[1].cpu(), 4, 3, 3)[b:b + 1], g["z_%d" % t][b:b + 1]) < TOL, (t, b)
                assert gu.rel_err(gu.sub(states[2][0].cpu(), 4, 3, 3)[b:b + 1], g["h_%d" % t][b:b + 1]) < TOL, (t, b)
            prev = I.clone()


def test_eraft_golden_sequence(gpu):
    """cista-eraft (BASELINE configs[2]) with the driver's evs_old carry (test_with_flow.py:144-149)."""
    from cista_flow_amd.e2v.e2v_model import ERAFTCistaNet
    g = gu.load("eraft_100x124.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    m = ERAFTCistaNet(args_for(H, W)).eval()
    wu.fill_module(m, seed)
    m = m.to(gpu)
    states, prev, ev_old = None, torch.zeros(B, 1, H, W, device=gpu), None
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            if ev_old is None:
                ev_old = wu.synth_events(B, 5, H, W, seed * 1000 + 999).to(gpu)   # see tools/gen_golden.py::run_eraft
            I, bf, states = m({"event_voxel": ev, "event_voxel_old": ev_old, "rec_img0": prev}, states, {})
            ev_old = ev.clone()
            assert gu.rel_err(bf["flow_final"].cpu(), g["flow_%d" % t]) < TOL, t
            assert gu.rel_err(bf["flow_init"].cpu(), g["flowlow_%d" % t]) < TOL, t
            assert gu.rel_err(I.cpu(), g["I_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[0].cpu()), g["c_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[1].cpu()), g["z_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][0].cpu()), g["h_%d" % t]) < TOL, t
            if t == 1:
                assert len(bf["flow_preds"]) == 12
                assert gu.rel_err(bf["flow_preds"][0].cpu(), g["preds0_1"]) < TOL
                assert gu.rel_err(bf["flow_preds"][6].cpu(), g["preds6_1"]) < TOL
            prev = I.clone()


@pytest.mark.parametrize("name", ["idnet_68x92.npz", "idnet_260x346.npz"])
def test_idnet_golden_sequence(gpu, name):
    """cista-idnet (BASELINE configs[4] geometry 260x346) with the driver's next_flow -> flow_init carry."""
    from cista_flow_amd.e2v.e2v_model import IDCistaNet
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    m = IDCistaNet(args_for(H, W)).eval()
    wu.fill_module(m, seed)
    m = m.to(gpu)
    states, prev, flow_init = None, torch.zeros(B, 1, H, W, device=gpu), None
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, flow_init, {})
            flow_init = bf["next_flow"]
            assert gu.rel_err(bf["flow_final"].cpu(), g["flow_%d" % t]) < TOL, t
            st = 3 if H >= 200 else 1
            assert gu.rel_err(bf["next_flow"].cpu()[..., ::st, ::st], g["next_%d" % t]) < TOL, t
            assert gu.rel_err(bf["delta_flow"][:, 1].cpu()[..., ::st, ::st], g["delta_%d" % t]) < TOL, t
            assert gu.rel_err(I.cpu(), g["I_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[0].cpu()), g["c_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[1].cpu()), g["z_%d" % t]) < TOL, t
            assert gu.rel_err(gu.sub(states[2][0].cpu()), g["h_%d" % t]) < TOL, t
            prev = I.clone()


def test_eraft_last_only_matches_full(gpu):
    """return_flow_preds=False skips the 11 dead mask-head / up-sampling evaluations; flow_final is unchanged."""
    from cista_flow_amd.ERAFT.eraft import ERAFT
    H, W, B = 128, 160, 2
    net = ERAFT(args_for(H, W)).eval()
    wu.fill_module(net, 9)
    net = net.to(gpu)
    a, b = wu.synth_events(B, 5, H, W, 1).to(gpu), wu.synth_events(B, 5, H, W, 2).to(gpu)
    with torch.no_grad():
        full = net(a, b)
        net.return_flow_preds = False
        last = net(a, b)
    assert last["flow_preds"] == []
    assert torch.equal(full["flow_final"], last["flow_final"])
    assert torch.equal(full["flow_preds"][-1][..., net.image_padder.pad_height:, net.image_padder.pad_width:], full["flow_final"])


def test_eiflow_vs_oracle_fresh_inputs(gpu):
    """B=3, 132x164 (pads to 160x192), 3 frames, seeds not used by any fixture; every output and full states."""
    from oracle import cista_oracle as orc
    H, W, B, seed = 132, 164, 3, 77
    m = build_eiflow(H, W, seed, gpu)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    st_g, st_o = None, None
    prev_g, prev_o = torch.zeros(B, 1, H, W, device=gpu), torch.zeros(B, 1, H, W)
    with torch.no_grad():
        for t in range(3):
            ev = wu.synth_events(B, 5, H, W, 5000 + t)
            I_g, bf_g, st_g = m({"event_voxel": ev.to(gpu), "rec_img0": prev_g}, st_g, {})
            I_o, bf_o, st_o = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev_o}, st_o)
            assert gu.rel_err(bf_g["flow_final"].cpu(), bf_o["flow_final"]) < TOL, t
            assert gu.rel_err(I_g.cpu(), I_o) < TOL, t
            assert gu.rel_err(st_g[0].cpu(), st_o[0]) < TOL, t
            assert gu.rel_err(st_g[1].cpu(), st_o[1]) < TOL, t
            assert gu.rel_err(st_g[2][0].cpu(), st_o[2][0]) < TOL, t
            assert gu.rel_err(st_g[2][1].cpu(), st_o[2][1]) < TOL, t
            for a, b in zip(bf_g["flow_preds"], bf_o["flow_preds"]):
                assert gu.rel_err(a.cpu(), b) < TOL
            prev_g, prev_o = I_g.clone(), I_o.clone()


def test_dceiflow_standalone_with_flow_init(gpu):
    from cista_flow_amd.DCEIFlow.DCEIFlow import DCEIFlow
    from oracle import cista_oracle as orc
    H, W, B = 128, 160, 2
    a = args_for(H, W)
    net = DCEIFlow(num_bins=5, args=a).eval()
    wu.fill_module(net, 31)
    net = net.to(gpu)
    sd = {"event_flownet." + k: v.cpu() for k, v in net.state_dict().items()}
    ev = wu.synth_events(B, 5, H, W, 123)
    img = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(4))
    finit = torch.randn(B, 2, H // 8, W // 8, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = net(ev.to(gpu), img.to(gpu), iters=4, flow_init=finit.to(gpu))
    ref = orc.eiflow_forward(sd, ev, img, iters=4, flow_init=finit)
    assert len(out["flow_preds"]) == 4
    assert gu.rel_err(out["flow_final"].cpu(), ref["flow_final"]) < TOL
    assert gu.rel_err(out["flow_init"].cpu(), ref["flow_init"]) < TOL


def test_zero_flow_is_passthrough_and_gt_flow_override(gpu):
    """`if not flow_final.any()` (e2v_model.py:184): all-zero gt_flow must skip both warps -- and because a
    zero-flow warp is NOT the identity (flow_utils quirk) the two branches give different images."""
    from oracle import cista_oracle as orc
    H, W, B, seed = 128, 128, 2, 41
    m = build_eiflow(H, W, seed, gpu)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ev0, ev1 = wu.synth_events(B, 5, H, W, 1), wu.synth_events(B, 5, H, W, 2)
    prev = torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        _, _, st_g = m({"event_voxel": ev0.to(gpu), "rec_img0": prev.to(gpu)}, None, {})
        _, _, st_o = orc.eiflow_step(sd, {"event_voxel": ev0, "rec_img0": prev}, None)
        z_before = st_g[1].clone()
        for gt in (torch.zeros(B, 2, H, W), torch.full((B, 2, H, W), 1.5)):
            sg = [st_g[0], st_g[1], st_g[2]]
            so = [st_o[0], st_o[1], st_o[2]]
            I_g, _, n_g = m({"event_voxel": ev1.to(gpu), "rec_img0": prev.to(gpu)}, sg, {"gt_flow": gt.to(gpu)})
            I_o, _, n_o = orc.eiflow_step(sd, {"event_voxel": ev1, "rec_img0": prev}, so, gt_flow=gt)
            assert gu.rel_err(I_g.cpu(), I_o) < TOL
            assert gu.rel_err(n_g[1].cpu(), n_o[1]) < TOL
            # the caller's list is mutated like the reference does (e2v_model.py:191)
            assert gu.rel_err(sg[1].cpu(), so[1]) < TOL
            if not gt.any():
                assert torch.equal(sg[1], z_before)


def test_weights_follow_load_state_dict(gpu):
    """Packed weights must be refreshed when the module's parameters change."""
    H, W, B = 128, 128, 1
    m = build_eiflow(H, W, 3, gpu)
    ev = wu.synth_events(B, 5, H, W, 8).to(gpu)
    prev = torch.zeros(B, 1, H, W, device=gpu)
    with torch.no_grad():
        I1, _, _ = m({"event_voxel": ev, "rec_img0": prev}, None, {})
        wu.fill_module(m, 4)
        I2, _, _ = m({"event_voxel": ev, "rec_img0": prev}, None, {})
        wu.fill_module(m, 3)
        I3, _, _ = m({"event_voxel": ev, "rec_img0": prev}, None, {})
    assert (I1 - I2).abs().max() > 1e-3
    assert torch.equal(I1, I3)


def test_buffer_reassignment_is_picked_up(gpu):
    """A plain tensor assigned over a BatchNorm buffer goes through no nn.Module hook: `invalidate()` re-packs at
    once, and without it the periodic full re-collection (runtime.REWALK_EVERY frames) notices."""
    from cista_flow_amd import runtime
    H, W, B = 128, 128, 1
    m = build_eiflow(H, W, 3, gpu)
    ev = wu.synth_events(B, 5, H, W, 8).to(gpu)
    prev = torch.zeros(B, 1, H, W, device=gpu)
    bn = m.event_flownet.cnet.norm1
    with torch.no_grad():
        f1 = m({"event_voxel": ev, "rec_img0": prev}, None, {})[1]["flow_final"].clone()
        orig = bn.running_mean.clone()
        bn.running_mean = bn.running_mean + 0.5          # new tensor object, no registration hook
        m._be().invalidate()
        f2 = m({"event_voxel": ev, "rec_img0": prev}, None, {})[1]["flow_final"].clone()
        assert (f1 - f2).abs().max() > 1e-4
        bn.running_mean = orig                           # back, this time WITHOUT invalidate()
        for _ in range(runtime.REWALK_EVERY + 1):
            f3 = m({"event_voxel": ev, "rec_img0": prev}, None, {})[1]["flow_final"]
        assert gu.rel_err(f3.cpu(), f1.cpu()) < 1e-6


@pytest.mark.parametrize("winop", ["1", "2"])
def test_persistent_winograd_in_the_model_is_bit_identical(gpu, monkeypatch, winop):
    """conv_wino_p_kernel<0 | 1> (tiles 48 / 49, CF_WINOP=1 | 2) compute conv_wino_kernel's arithmetic bit for bit, so the WHOLE recurrent
    model must reproduce the default run exactly -- at the headline geometry, where the walkers carry several items, the layers have two
    and three channel segments and every fused epilogue is in play, over enough frames that a hand-off race shows (r04: a VMEM issue-order
    race in the pipelined variant passed every single-operator test and failed here)."""
    H, W, B = 180, 240, 4
    m = build_eiflow(H, W, 5, gpu)
    evs = [wu.synth_events(B, 5, H, W, 300 + i).to(gpu) for i in range(4)]

    def run(n):
        prev, st, res = torch.zeros(B, 1, H, W, device=gpu), None, []
        with torch.no_grad():
            for t in range(n):
                I, bf, st = m({"event_voxel": evs[t % 4], "rec_img0": prev}, st, {})
                prev = I
                res.append((I.clone(), bf["flow_final"].clone(), st[1].clone(), st[2][0].clone()))
        torch.cuda.synchronize()
        return res

    monkeypatch.delenv("CF_WINOP", raising=False)
    ref = run(24)
    monkeypatch.setenv("CF_WINOP", winop)
    got = run(24)
    monkeypatch.delenv("CF_WINOP", raising=False)
    for t, (a, b) in enumerate(zip(ref, got)):
        for x, y in zip(a, b):
            assert torch.equal(x, y), t


@pytest.mark.parametrize("B", [1, 2])
def test_deep_winograd16_in_the_model_is_bit_identical(gpu, monkeypatch, B):
    """conv_wino16_kernel<1> (tile 50: U two chunks ahead, raw ring of four, slot-scheduled chunk step; what launches of at most
    CF_WINO16_DEEP_MAX workgroups take) computes conv_wino16_kernel<0>'s arithmetic bit for bit: the WHOLE recurrent model at the headline
    geometry in the small-batch regime, where ~20 launch sites per frame take it, must reproduce the run with it turned off exactly, over
    enough frames that a counted-vmcnt hand-off race would show (the lesson of tile 49)."""
    H, W = 180, 240
    m = build_eiflow(H, W, 5, gpu)
    evs = [wu.synth_events(B, 5, H, W, 500 + i).to(gpu) for i in range(4)]
    h = m._be().get(B, gpu)
    h.plan_enable(True)

    def run(n):
        prev, st, res = torch.zeros(B, 1, H, W, device=gpu), None, []
        with torch.no_grad():
            for t in range(n):
                I, bf, st = m({"event_voxel": evs[t % 4], "rec_img0": prev}, st, {})
                prev = I
                res.append((I.clone(), bf["flow_final"].clone(), st[1].clone(), st[2][0].clone()))
        torch.cuda.synchronize()
        return res

    monkeypatch.setenv("CF_WINO16_DEEP_MAX", "0")
    ref = run(32)
    monkeypatch.delenv("CF_WINO16_DEEP_MAX", raising=False)
    got = run(32)
    kernels = {r["kernel"] for r in h.plan()["rows"]}
    assert "conv_wino16_kernel<deep>" in kernels and "conv_wino16_kernel" in kernels, kernels      # both instantiations really ran
    for t, (a, b) in enumerate(zip(ref, got)):
        for x, y in zip(a, b):
            assert torch.equal(x, y), t


@pytest.mark.parametrize("knob", ["CF_PYRAMID_FUSED"])
def test_fused_small_launches_are_bit_identical(gpu, monkeypatch, knob):
    """Launch-count reductions on the dependent chains against the launches they replace, same bits:
    CF_PYRAMID_FUSED -- the one-launch correlation pyramid (levels 1..3 + coords1 init + flag reset, corr_pyramid_kernel) vs the cascade
    of corr_pool launches (odd map sizes: 23 x 30 -> 11 x 15 -> 5 x 7 -> 2 x 3), with flow_low riding in the last upflow8 launch."""
    H, W, B = 180, 240, 2
    m = build_eiflow(H, W, 3, gpu)
    evs = [wu.synth_events(B, 5, H, W, 40 + i).to(gpu) for i in range(2)]
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv(knob, fused)
        prev, st, res = torch.zeros(B, 1, H, W, device=gpu), None, []
        with torch.no_grad():
            for ev in evs:
                I, bf, st = m({"event_voxel": ev, "rec_img0": prev}, st, {})
                prev = I
                res += [I.clone(), bf["flow_final"].clone()]
        outs[fused] = res
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)
    assert outs["1"][1].abs().max() > 0


def test_inputs_are_validated(gpu):
    m = build_eiflow(128, 128, 3, gpu)
    with pytest.raises(ValueError):
        m({"event_voxel": torch.zeros(1, 5, 64, 64, device=gpu), "rec_img0": torch.zeros(1, 1, 128, 128, device=gpu)}, None, {})
    with pytest.raises(RuntimeError):
        m({"event_voxel": torch.zeros(1, 5, 128, 128), "rec_img0": torch.zeros(1, 1, 128, 128)}, None, {})
    with pytest.raises(TypeError):
        m({"event_voxel": torch.zeros(1, 5, 128, 128, device=gpu, dtype=torch.float64),
           "rec_img0": torch.zeros(1, 1, 128, 128, device=gpu)}, None, {})


def test_eiflow_f16x3_precision_mode(gpu):
    """Opt-in f16x3 arithmetic (3 x f16 MFMA per product, fp32 accumulate): still far inside the 1e-3 bar."""
    g = gu.load("eiflow_100x124.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
    m = DCEIFlowCistaNet(args_for(H, W)).eval()
    m.precision = "f16x3"
    wu.fill_module(m, seed)
    m = m.to(gpu)
    states, prev = None, torch.zeros(B, 1, H, W, device=gpu)
    worst = 0.0
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, {})
            errs = [gu.rel_err(bf["flow_final"].cpu(), g["flow_%d" % t]), gu.rel_err(I.cpu(), g["I_%d" % t]),
                    gu.rel_err(gu.sub(states[1].cpu()), g["z_%d" % t]), gu.rel_err(gu.sub(states[0].cpu()), g["c_%d" % t]),
                    gu.rel_err(gu.sub(states[2][0].cpu()), g["h_%d" % t])]
            worst = max(worst, max(errs))
            prev = I.clone()
    assert worst < 3e-4, worst


def _build(kind, H, W, seed, gpu, precision="f32"):
    from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet, ERAFTCistaNet, IDCistaNet
    cls = {"eiflow": DCEIFlowCistaNet, "eraft": ERAFTCistaNet, "idnet": IDCistaNet}[kind]
    m = cls(args_for(H, W)).eval()
    m.precision = precision
    wu.fill_module(m, seed)
    m = m.to(gpu)
    if kind != "idnet":
        m.event_flownet.return_flow_preds = True          # like the reference: every iteration's up-flow is produced
    return m


def _drive(kind, m, evs, ev_first_old, dev):
    """The drivers' per-frame loop (test_with_flow.py:120-156) over the voxel grids `evs`: feedback of the prediction,
    eraft's `evs_old` carry, idnet's next_flow -> flow_init carry.  Returns per-frame (I, flow_final, z, h, preds)."""
    B, _, H, W = evs[0].shape
    states, prev, flow_init, old, out = None, torch.zeros(B, 1, H, W, device=dev), None, ev_first_old, []
    for ev in evs:
        if kind == "eiflow":
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, {})
        elif kind == "eraft":
            I, bf, states = m({"event_voxel": ev, "event_voxel_old": old, "rec_img0": prev}, states, {})
            old = ev
        else:
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, flow_init, {})
            flow_init = bf["next_flow"]
        out.append((I, bf["flow_final"], states[1], states[2][0], bf["flow_preds"]))
        prev = I.clone()
    return out


def _oracle_drive(kind, sd, evs, ev_first_old):
    from oracle import cista_oracle as orc
    B, _, H, W = evs[0].shape
    states, prev, flow_init, old, out = None, torch.zeros(B, 1, H, W), None, ev_first_old, []
    for ev in evs:
        if kind == "eiflow":
            I, bf, states = orc.eiflow_step(sd, {"event_voxel": ev, "rec_img0": prev}, states)
        elif kind == "eraft":
            I, bf, states = orc.eraft_step(sd, {"event_voxel": ev, "event_voxel_old": old, "rec_img0": prev}, states)
            old = ev
        else:
            I, bf, states = orc.idnet_step(sd, {"event_voxel": ev, "rec_img0": prev}, states, flow_init)
            flow_init = bf["next_flow"]
        out.append((I, bf["flow_final"], states[1], states[2][0], bf["flow_preds"]))
        prev = I.clone()
    return out


FULL_SIZE = [("eiflow", 180, 240, 8, "f32"),     # BASELINE configs[1]
             ("eraft", 180, 240, 8, "f32"),      # configs[2]
             ("eiflow", 480, 640, 4, "f32"),     # configs[3]: 32 sequences over 8 GPUs = 4 per GPU
             ("idnet", 260, 346, 16, "f32"),     # configs[4] geometry in exact fp32
             ("idnet", 260, 346, 16, "f16")]     # configs[4] as named: f16 MFMA products


@pytest.mark.parametrize("kind,H,W,B,prec", FULL_SIZE, ids=["%s-%dx%d-B%d-%s" % c for c in FULL_SIZE])
def test_full_size_batch_properties(gpu, kind, H, W, B, prec):
    """Every BASELINE config at its FULL size and per-GPU batch (where the oracle is too slow for the whole batch):
    size-independent properties of the step.  Sequences are independent, so (a) identical inputs in different batch
    slots give bit-identical outputs (every kernel -- tiles, convex up-sampling, deblur -- indexes its image correctly
    at this grid size), (b) a slot's result does not depend on what the other slots hold, (c) re-running is
    deterministic, and (d) slot 0 of a two-frame recurrent sequence agrees with the CPU oracle run at B=1."""
    m = _build(kind, H, W, 4242, gpu, prec)
    ev0 = [wu.synth_events(1, 5, H, W, 9100 + t) for t in range(2)]
    ev1 = [wu.synth_events(1, 5, H, W, 9200 + t) for t in range(2)]
    old0, old1 = wu.synth_events(1, 5, H, W, 9300), wu.synth_events(1, 5, H, W, 9301)
    evA = [torch.cat([e] * B, 0).to(gpu) for e in ev0]                           # all slots equal
    evB = [torch.cat([a] + [b] * (B - 1), 0).to(gpu) for a, b in zip(ev0, ev1)]  # slot 0 equal, the rest different
    oldA = torch.cat([old0] * B, 0).to(gpu)
    oldB = torch.cat([old0] + [old1] * (B - 1), 0).to(gpu)
    with torch.no_grad():
        A = _drive(kind, m, evA, oldA, gpu)
        Bo = _drive(kind, m, evB, oldB, gpu)
        A2 = _drive(kind, m, evA, oldA, gpu)
    torch.cuda.synchronize()
    for t in range(2):
        I, flow, z, hh, preds = A[t]
        assert torch.isfinite(I).all() and torch.isfinite(flow).all()
        for k in range(1, B):                                                    # (a), cold and recurrent frame
            assert torch.equal(I[k], I[0]) and torch.equal(flow[k], flow[0]), (t, k)
            assert torch.equal(z[k], z[0]) and torch.equal(hh[k], hh[0]), (t, k)
            for pr in preds:
                assert torch.equal(pr[k], pr[0]), (t, k)
        assert torch.equal(Bo[t][0][0], I[0]) and torch.equal(Bo[t][1][0], flow[0]) and torch.equal(Bo[t][2][0], z[0])   # (b)
        assert not torch.equal(Bo[t][0][1], I[1])
        assert torch.equal(A2[t][0], I) and torch.equal(A2[t][1], flow) and torch.equal(A2[t][2], z)                      # (c)
    # (d) slot 0 against the oracle at B=1 (fp32 modes: the 2e-4 regression bar; plain f16 products: the 1e-2 this
    # reduced-precision mode is documented with)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        O = _oracle_drive(kind, sd, ev0, old0)
        O1 = _oracle_drive(kind, sd, ev1, old1)          # the LAST slot of the mixed batch holds a different sequence
    tol = TOL if prec == "f32" else 1e-2

    def err(got, ref):
        # fp32: the three-norm figure every golden comparison uses; the plain-f16 extra is documented in the max norm only (1e-2 of
        # the tensor's scale, observed 3e-3; its L2 / element-wise figures are up to 4x that and are not part of any claim)
        return gu.rel_err(got, ref) if prec == "f32" else gu.err_norms(got, ref)["inf"]

    for t in range(2):
        for name, got, ref in (("I", A[t][0], O[t][0]), ("flow", A[t][1], O[t][1]), ("z", A[t][2], O[t][2]), ("h", A[t][3], O[t][3])):
            assert err(got[:1].cpu(), ref) < tol, (t, name)
        assert len(A[t][4]) == len(O[t][4])
        for a, b in zip(A[t][4], O[t][4]):
            assert err(a[:1].cpu(), b) < tol, t
        # slot B-1 of the mixed batch (a last-slot indexing error cannot hide behind slot 0)
        for name, got, ref in (("I", Bo[t][0], O1[t][0]), ("flow", Bo[t][1], O1[t][1]), ("z", Bo[t][2], O1[t][2]), ("h", Bo[t][3], O1[t][3])):
            assert err(got[B - 1:].cpu(), ref) < tol, (t, name, "last slot")


@pytest.mark.parametrize("name,kind", [("eraft_180x240.npz", "eraft"), ("eiflow_480x640.npz", "eiflow")])
def test_fullsize_reference_goldens(gpu, name, kind):
    """The HIP path against fixtures the REFERENCE produced at the full size of BASELINE configs[2] (eraft 180x240) and
    configs[3] (eiflow 480x640): B = 1, two recurrent frames, strided probes of every output and state (round 3)."""
    from test_oracle_golden import fullsize_probe
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"][:5]]
    m = _build(kind, H, W, seed, gpu)
    states, prev = None, torch.zeros(B, 1, H, W, device=gpu)
    old = wu.synth_events(B, 5, H, W, seed * 1000 + 999).to(gpu)
    with torch.no_grad():
        for t in range(frames):
            ev = wu.synth_events(B, 5, H, W, seed * 1000 + t).to(gpu)
            data = {"event_voxel": ev, "rec_img0": prev}
            if kind == "eraft":
                data["event_voxel_old"] = old
            I, bf, states = m(data, states, {})
            old = ev
            st = [states[0].cpu(), states[1].cpu(), (states[2][0].cpu(), states[2][1].cpu())]
            bfc = {"flow_final": bf["flow_final"].cpu(), "flow_init": bf["flow_init"].cpu(), "flow_preds": [bf["flow_preds"][0].cpu()]}
            for nm, got, ref in fullsize_probe(g, t, I.cpu(), bfc, st):
                assert gu.rel_err(got, ref) < TOL, (t, nm)
            prev = I.clone()


def test_eiflow_fullstate_golden(gpu):
    """Every element of the sparse code after the 4-frame eiflow_100x124 sequence against the reference's own tensor
    (the per-frame fixtures only keep a strided 1/36 probe of each state)."""
    g = gu.load("eiflow_100x124_fullstate.npz")
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    m = build_eiflow(H, W, seed, gpu)
    evs = [wu.synth_events(B, 5, H, W, seed * 1000 + t).to(gpu) for t in range(frames)]
    with torch.no_grad():
        out = _drive("eiflow", m, evs, None, gpu)
    assert gu.rel_err(out[-1][2].cpu(), g["z_full"]) < TOL
    assert gu.rel_err(out[-1][3].cpu(), g["h_full"]) < TOL
    # element-wise, not only relative to the tensor maximum: the soft threshold produces exact zeros, and an element
    # the reference zeroed must be (numerically) zero here too
    ref = torch.from_numpy(g["z_full"])
    got = out[-1][2].cpu()
    assert ((got - ref).abs() <= TOL * ref.abs().max()).all()
    assert ((ref == 0) == (got == 0)).float().mean() > 0.999


def _idnet_sequence_errors(gpu, name, precision):
    from cista_flow_amd.e2v.e2v_model import IDCistaNet
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    m = IDCistaNet(args_for(H, W)).eval()
    m.precision = precision
    wu.fill_module(m, seed)
    m = m.to(gpu)
    states, prev, flow_init = None, torch.zeros(B, 1, H, W, device=gpu), None
    worst = {}
    with torch.no_grad():
        for t in range(frames):
            ev = torch.from_numpy(g["ev_%d" % t]).to(gpu)
            I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, flow_init, {})
            flow_init = bf["next_flow"]
            # plain f16 is documented in the max norm only (see test_full_size_batch_properties); the fp32-grade modes in all three
            e = (lambda a, b: gu.err_norms(a, b)["inf"]) if precision == "f16" else gu.rel_err
            errs = {"flow": e(bf["flow_final"].cpu(), g["flow_%d" % t]), "I": e(I.cpu(), g["I_%d" % t]),
                    "z": e(gu.sub(states[1].cpu()), g["z_%d" % t]), "c": e(gu.sub(states[0].cpu()), g["c_%d" % t]),
                    "h": e(gu.sub(states[2][0].cpu()), g["h_%d" % t])}
            for k, v in errs.items():
                worst[k] = max(worst.get(k, 0.0), v)
            prev = I.clone()
    return worst


def test_idnet_f16_precision_mode(gpu):
    """BASELINE configs[4] ("cista-idnet 346x260 batch=16, fp16 with MFMA"): plain-f16 operands, fp32 accumulate and
    fp32 tensors, against the fp32 reference golden.  One f16 rounding of each operand is ~5e-4 relative, so this
    mode cannot meet the 1e-3 fp32 bar by construction; the tolerance asserted here is 1e-2 of tensor scale over a
    recurrent 3-frame sequence (observed: <= 3e-3; values are printed by -s).  f16x3 on the same sequence stays under 3e-4."""
    w16 = _idnet_sequence_errors(gpu, "idnet_260x346.npz", "f16")
    w3 = _idnet_sequence_errors(gpu, "idnet_260x346.npz", "f16x3")
    print("idnet 260x346 worst rel err  f16:", w16, " f16x3:", w3)
    assert max(w3.values()) < 3e-4, w3
    assert max(w16.values()) < 1e-2, w16


def test_eraft_prev_feature_reuse_is_bit_identical(gpu):
    """ERAFT driver carry (test_with_flow.py:144-149): when event_voxel_old is the previous call's event_voxel tensor,
    fnet's feature map is reused (cf_hint_prev_grid) -- results must equal the recomputing path bit for bit, and an
    in-place write to the carried tensor must switch the reuse off."""
    from cista_flow_amd.e2v.e2v_model import ERAFTCistaNet
    H, W, B = 100, 124, 2
    outs = {}
    for reuse in (False, True):
        m = ERAFTCistaNet(args_for(H, W)).eval()
        m.reuse_prev_features = reuse
        wu.fill_module(m, 31)
        m = m.to(gpu)
        evs = [wu.synth_events(B, 5, H, W, 700 + t).to(gpu) for t in range(5)]
        states, prev, res = None, torch.zeros(B, 1, H, W, device=gpu), []
        with torch.no_grad():
            for t in range(1, 5):
                if t == 3:
                    evs[2].mul_(1.0)          # in-place op on the carried tensor: version bump, reuse must be skipped
                I, bf, states = m({"event_voxel": evs[t], "event_voxel_old": evs[t - 1], "rec_img0": prev}, states, {})
                res.append((I.clone(), bf["flow_final"].clone(), states[1].clone()))
                prev = I.clone()
        outs[reuse] = res
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_eraft_standalone_with_flow_init_vs_oracle(gpu):
    """ERAFT(cfgs).forward(image1, image2, iters, flow_init) (eraft.py:114) as a stand-alone module, warm-started."""
    from cista_flow_amd.ERAFT.eraft import ERAFT
    from oracle import cista_oracle as orc
    H, W, B = 100, 124, 2
    net = ERAFT(args_for(H, W)).eval()
    wu.fill_module(net, 17)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(gpu)
    a, b = wu.synth_events(B, 5, H, W, 41), wu.synth_events(B, 5, H, W, 42)
    with torch.no_grad():
        g1 = net(a.to(gpu), b.to(gpu), iters=5)
        o1 = orc.eraft_forward(sd, a, b, iters=5, prefix="")
        g2 = net(b.to(gpu), a.to(gpu), iters=5, flow_init=g1["flow_init"])
        o2 = orc.eraft_forward(sd, b, a, iters=5, flow_init=o1["flow_init"], prefix="")
    for g, o in ((g1, o1), (g2, o2)):
        assert gu.rel_err(g["flow_final"].cpu(), o["flow_final"]) < TOL
        assert gu.rel_err(g["flow_init"].cpu(), o["flow_init"]) < TOL
        assert len(g["flow_preds"]) == 5
        assert gu.rel_err(g["flow_preds"][-1].cpu(), o["flow_preds"][-1]) < TOL


def test_idedeqido_standalone_vs_oracle(gpu):
    """IDEDEQIDO(config).forward(event_bins, flow_init) (idedeq.py:124) as a stand-alone module, with the next_flow carry."""
    from types import SimpleNamespace
    from cista_flow_amd.idn.idedeq import IDEDEQIDO
    from oracle import cista_oracle as orc
    H, W, B = 68, 92, 2
    net = IDEDEQIDO(SimpleNamespace(update_iters=1, pred_next_flow=True, image_dim=[H, W], num_bins=5)).eval()
    wu.fill_module(net, 23)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(gpu)
    e1, e2 = wu.synth_events(B, 5, H, W, 51), wu.synth_events(B, 5, H, W, 52)
    with torch.no_grad():
        g1 = net(e1.to(gpu))
        o1 = orc.idnet_forward(sd, e1, prefix="")
        g2 = net(e2.to(gpu), flow_init=g1["next_flow"])
        o2 = orc.idnet_forward(sd, e2, flow_init=o1["next_flow"], prefix="")
    for g, o in ((g1, o1), (g2, o2)):
        for k in ("flow_final", "next_flow", "delta_flow"):
            assert gu.rel_err(g[k].cpu(), o[k]) < TOL, k
        assert gu.rel_err(g["flow_preds"][0].cpu(), o["flow_preds"][0]) < TOL


@pytest.mark.parametrize("kind", ["eiflow", "eraft", "idnet"])
def test_graph_replay_is_bit_identical(gpu, kind):
    """hipGraph replay of cf_step (captured per distinct caller-pointer tuple) against the eager path: the same
    kernels with the same arguments, so every output of every frame must be bit-identical; and the steady-state loop
    of the drivers must actually hit the cache (PyTorch's allocator recycles the per-frame blocks)."""
    H, W, B = (100, 124, 2) if kind != "idnet" else (68, 92, 2)
    res = {}
    T = 48
    for graph in (False, True):
        # a step is captured on the SECOND sighting of its caller-pointer tuple, i.e. once PyTorch's caching allocator hands the per-frame
        # blocks out again; what earlier tests left in its pools changes that period, so start from empty pools
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        m = _build(kind, H, W, 77, gpu)
        if kind == "eraft":
            m.reuse_prev_features = True
        h = m._be().get(B, gpu)
        h.graph_enable(graph)
        # two alternating voxel-grid buffers x the allocator's own period: a handful of distinct pointer tuples, each
        # captured on its second sighting
        evs = [wu.synth_events(B, 5, H, W, 300 + t).to(gpu) for t in range(2)]
        states, prev, flow_init, outs = None, torch.zeros(B, 1, H, W, device=gpu), None, []
        with torch.no_grad():
            for t in range(T):
                ev = evs[t % 2]
                if kind == "eiflow":
                    I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, {})
                elif kind == "eraft":
                    I, bf, states = m({"event_voxel": ev, "event_voxel_old": evs[(t - 1) % 2], "rec_img0": prev}, states, {})
                else:
                    I, bf, states = m({"event_voxel": ev, "rec_img0": prev}, states, flow_init, {})
                    flow_init = bf["next_flow"]
                outs.append([I.cpu(), bf["flow_final"].cpu(), states[0].cpu(), states[1].cpu(), states[2][0].cpu(), states[2][1].cpu()]
                            + [p.cpu() for p in bf["flow_preds"]])
                prev = I
                del I, bf
        torch.cuda.synchronize()
        res[graph] = (outs, h.graph_stats(), h.lib.cf_last_error(h.h).decode())
    assert res[False][1][0] == 0 and res[False][1][1] == 0
    cap, rep, _ = res[True][1]
    assert "turned off" not in res[True][2], res[True][2]      # a failed capture is reported (cf_last_error + stderr), never silent
    assert cap >= 1 and rep >= 4, (cap, rep)       # the loop settled into replays
    for a, b in zip(res[False][0], res[True][0]):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_encoder_pair_mode_matches_two_stream_mode(gpu, monkeypatch):
    """CF_ENC_PAIR=1 runs fnet + enet (eiflow) / fnet on both grids (eraft) as ONE batch of 2B images with grouped weights
    instead of two chains on two streams: same layers and weights, tiles may differ (batch-dependent choice), so the
    results agree to fp32 summation order; within a mode everything stays deterministic."""
    H, W, B = 100, 124, 2
    res = {}
    for kind in ("eiflow", "eraft"):
        for pair in ("0", "1"):
            monkeypatch.setenv("CF_ENC_PAIR", pair)
            m = _build(kind, H, W, 61, gpu)
            evs = [wu.synth_events(B, 5, H, W, 800 + t).to(gpu) for t in range(3)]
            with torch.no_grad():
                res[(kind, pair)] = _drive(kind, m, evs[1:], evs[0], gpu)
        for a, b in zip(res[(kind, "0")], res[(kind, "1")]):
            for x, y in zip(a[:4], b[:4]):
                assert gu.rel_err(x.cpu(), y.cpu()) < 2e-5, kind


@pytest.mark.parametrize("name,kind", [("eiflow_100x124.npz", "eiflow"), ("eraft_100x124.npz", "eraft"), ("idnet_68x92.npz", "idnet")])
def test_direct_conv_path_golden(gpu, monkeypatch, name, kind):
    """CF_WINO=0: every 3x3 convolution on the direct implicit-GEMM kernel (the default runs them as Winograd
    F(2x2,3x3)) -- the reference goldens must hold for both, and the two paths agree to the transforms' rounding."""
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    outs = {}
    for wino in ("0", "1"):
        monkeypatch.setenv("CF_WINO", wino)
        m = _build(kind, H, W, seed, gpu)
        evs = [torch.from_numpy(g["ev_%d" % t]).to(gpu) for t in range(frames)]
        old = wu.synth_events(B, 5, H, W, seed * 1000 + 999).to(gpu) if kind == "eraft" else None
        with torch.no_grad():
            outs[wino] = _drive(kind, m, evs, old, gpu)
        for t in range(frames):
            assert gu.rel_err(outs[wino][t][0].cpu(), g["I_%d" % t]) < TOL, (wino, t)
            assert gu.rel_err(outs[wino][t][1].cpu(), g["flow_%d" % t]) < TOL, (wino, t)
            assert gu.rel_err(gu.sub(outs[wino][t][2].cpu()), g["z_%d" % t]) < TOL, (wino, t)
    for a, b in zip(outs["0"], outs["1"]):
        assert gu.rel_err(a[0].cpu(), b[0].cpu()) < 1e-4 and gu.rel_err(a[2].cpu(), b[2].cpu()) < 1e-4


@pytest.mark.parametrize("name,kind", [("eiflow_100x124.npz", "eiflow"), ("eiflow_180x240.npz", "eiflow")])
def test_winograd_f4x4_path_golden(gpu, monkeypatch, name, kind):
    """CF_WINO4_MIN=1: every 3x3 / stride-1 convolution with 16-channel segments on the opt-in F(4x4,3x3) kernel (conv_wino4_kernel)
    through the whole recurrent network -- the reference goldens hold at the unchanged 2e-4 (the CPU study that preceded the kernel
    predicted 2e-5)."""
    g = gu.load(name)
    H, W, B, frames, seed = [int(v) for v in g["meta"]]
    monkeypatch.setenv("CF_WINO4_MIN", "1")
    m = _build(kind, H, W, seed, gpu)
    evs = [torch.from_numpy(g["ev_%d" % t]).to(gpu) for t in range(frames)]
    with torch.no_grad():
        out = _drive(kind, m, evs, None, gpu)
    worst = 0.0
    for t in range(frames):
        for got, ref in ((out[t][0].cpu(), g["I_%d" % t]), (out[t][1].cpu(), g["flow_%d" % t]), (gu.sub(out[t][2].cpu()), g["z_%d" % t])):
            worst = max(worst, gu.rel_err(got, ref))
    assert worst < TOL, worst
    monkeypatch.delenv("CF_WINO4_MIN")
    m2 = _build(kind, H, W, seed, gpu)              # a new handle: back on the default kernels
    with torch.no_grad():
        out2 = _drive(kind, m2, evs, None, gpu)
    assert not torch.equal(out2[-1][0], out[-1][0])          # the F(4x4,3x3) path really ran (results differ in the last bits)
    assert gu.rel_err(out2[-1][0].cpu(), out[-1][0].cpu()) < 1e-4


def test_long_recurrence_stays_within_contract(gpu):
    """Twelve recurrent frames (states, previous reconstruction and flow all fed back) of cista-eiflow at 128x160, B = 2, against the
    CPU oracle frame by frame: fp32 summation-order differences are amplified by the recurrence, and the contract (BASELINE.json:
    1e-3 relative) has to hold at the END of a sequence, not only after the two to four frames of the reference fixtures."""
    H, W, B, frames = 128, 160, 2, 12
    m = _build("eiflow", H, W, 77, gpu)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    evs = [wu.synth_events(B, 5, H, W, 7700 + t) for t in range(frames)]
    with torch.no_grad():
        G = _drive("eiflow", m, [e.to(gpu) for e in evs], None, gpu)
        O = _oracle_drive("eiflow", sd, evs, None)
    worst = []
    for t in range(frames):
        e = max(gu.rel_err(G[t][0].cpu(), O[t][0]), gu.rel_err(G[t][1].cpu(), O[t][1]), gu.rel_err(G[t][2].cpu(), O[t][2]),
                gu.rel_err(G[t][3].cpu(), O[t][3]))
        worst.append(e)
    assert max(worst) < 1e-3, worst
    assert worst[-1] < 5e-4, worst            # observed ~1e-5 .. 1e-4: no runaway growth over the sequence
