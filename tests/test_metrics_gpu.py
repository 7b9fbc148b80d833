"""f-3: evaluation metrics on the device (csrc/metrics.hip behind cista_flow_amd/loss.py) against
(1) values the reference's loss.py produced (tests/golden/metrics.npz, tools/gen_golden.py) and (2) the CPU oracle on
fresh inputs at batch > 1.  Reductions run in fp64 on the device and in fp32 in the reference: agreement is to fp32
rounding of the reference's own sums (asserted 2e-5 relative)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu          # noqa: E402
import weights_util as wu         # noqa: E402

pytestmark = pytest.mark.gpu
RT = 2e-5


def close(a, b, rt=RT, at=1e-7):
    return abs(float(a) - float(b)) <= rt * abs(float(b)) + at


def test_recon_metrics_golden(gpu):
    from cista_flow_amd.loss import ReconLoss, PSNR
    g = gu.load("metrics.npz")
    rec, tgt = torch.from_numpy(g["rec"]).to(gpu), torch.from_numpy(g["tgt"]).to(gpu)
    m = ReconLoss(None).evaluate(rec, tgt)
    assert set(m) == {"mse", "psnr", "ssim"}          # 'lpips' needs network weights: not built
    assert close(m["mse"], g["mse"]) and abs(m["psnr"] - float(g["psnr"])) < 1e-4
    assert float(PSNR()(rec, rec)) == 100.0 == float(g["psnr_same"])


@pytest.mark.parametrize("mode", ["forward", "backward"])
def test_flow_metrics_golden(gpu, mode):
    from cista_flow_amd.loss import FlowL1LossDict
    from cista_flow_amd.utils.flow_utils import FrameWarp
    g = gu.load("metrics.npz")
    t = {k: torch.from_numpy(g[k + "_" + mode]).to(gpu) for k in ("flow", "gt", "img0", "img1", "valid")}
    L = FlowL1LossDict([t["flow"].shape[2], t["flow"].shape[3]], FrameWarp(mode))
    keys = ["photo_loss", "epe", "1px", "3px", "5px", "out"]
    m1 = L.evaluate(t["flow"], {"gt_flow": t["gt"], "gt_img0": t["img0"], "gt_img1": t["img1"], "flow_valid": t["valid"]})
    m2 = L.evaluate(t["flow"], {"gt_flow": t["gt"], "gt_img0": t["img0"], "gt_img1": t["img1"]})
    for got, ref in ((m1, g["fm_valid_" + mode]), (m2, g["fm_photo_" + mode])):
        for k, r in zip(keys, ref):
            assert close(got[k], r), (mode, k, got[k], float(r))


def test_fwl_golden(gpu):
    from cista_flow_amd.loss import voxel_warping_flow_loss, fwl_metrics
    g = gu.load("metrics.npz")
    evs, flow = torch.from_numpy(g["fwl_evs"]).to(gpu), torch.from_numpy(g["fwl_flow"]).to(gpu)
    # the drivers' expression (test_wo_flow.py:161)
    fwl = voxel_warping_flow_loss(evs, flow) / voxel_warping_flow_loss(evs, torch.zeros_like(flow))
    assert close(fwl, g["fwl"][2], 1e-5)
    m = fwl_metrics(evs, flow).cpu()
    assert close(m[0], g["fwl"][0], 1e-5) and close(m[1], g["fwl"][1], 1e-5) and close(m[2], g["fwl"][2], 1e-5)


def test_metrics_vs_oracle_batched(gpu):
    """B = 3, ragged size, no validity mask for one call: HIP vs the oracle restatement; plus determinism."""
    from cista_flow_amd.loss import flow_metrics, fwl_metrics, recon_metrics
    from oracle import cista_oracle as orc
    gen = torch.Generator().manual_seed(5)
    B, H, W = 3, 67, 93
    flow = 4.0 * torch.randn(B, 2, H, W, generator=gen)
    gt = flow + torch.randn(B, 2, H, W, generator=gen)
    gt[1, :, :3] = 1000.0
    i0 = torch.rand(B, 1, H, W, generator=gen)
    i1 = (i0 + 0.05 * torch.randn(B, 1, H, W, generator=gen)).clamp(0, 1)
    for mode in ("forward", "backward"):
        got = flow_metrics(flow.to(gpu), gt.to(gpu), i0.to(gpu), i1.to(gpu), None, mode)
        again = flow_metrics(flow.to(gpu), gt.to(gpu), i0.to(gpu), i1.to(gpu), None, mode)
        assert torch.equal(got, again)
        ref = orc.flow_metrics(flow, gt, i0, i1, None, mode)
        for a, b in zip(got.cpu().tolist(), ref):
            assert close(a, b), (mode, got, ref)
    evs = wu.synth_events(B, 5, H, W, 99)
    m = fwl_metrics(evs.to(gpu), flow.to(gpu)).cpu()
    assert close(m[0], orc.voxel_warping_flow_loss(evs, flow), 1e-5)
    assert close(m[1], orc.voxel_warping_flow_loss(evs, torch.zeros_like(flow)), 1e-5)
    r = recon_metrics(i0.to(gpu), i1.to(gpu)).cpu()
    mse, psnr = orc.recon_metrics(i0, i1)
    assert close(r[0], mse, 1e-9) and abs(float(r[1]) - psnr) < 1e-9


def test_metrics_noncontiguous_inputs(gpu):
    """Sliced / permuted inputs: every `.contiguous()` copy must outlive the launch -- with both flow and gt
    non-contiguous a freed first copy used to hand its block to the second one (aliased arguments, epe = 0)."""
    from cista_flow_amd.loss import flow_metrics, fwl_metrics, recon_metrics
    gen = torch.Generator().manual_seed(11)
    B, H, W = 2, 48, 80
    flow_w = (3.0 * torch.randn(B, 2, H, 2 * W, generator=gen)).to(gpu)
    gt_w = (flow_w.cpu() + torch.randn(B, 2, H, 2 * W, generator=gen)).to(gpu)
    i0_w = torch.rand(B, 1, H, 2 * W, generator=gen).to(gpu)
    i1_w = torch.rand(B, 1, H, 2 * W, generator=gen).to(gpu)
    flow, gt, i0, i1 = flow_w[..., ::2], gt_w[..., ::2], i0_w[..., ::2], i1_w[..., ::2]
    assert not flow.is_contiguous() and not gt.is_contiguous()
    for mode in ("forward", "backward"):
        got = flow_metrics(flow, gt, i0, i1, None, mode)
        ref = flow_metrics(flow.clone(memory_format=torch.contiguous_format), gt.clone(memory_format=torch.contiguous_format),
                           i0.clone(memory_format=torch.contiguous_format), i1.clone(memory_format=torch.contiguous_format), None, mode)
        assert torch.equal(got, ref) and float(got[1]) > 0.1, (mode, got, ref)
    evs_w = wu.synth_events(B, 5, H, 2 * W, 3).to(gpu)
    evs = evs_w[..., ::2]
    assert torch.equal(fwl_metrics(evs, flow), fwl_metrics(evs.clone(memory_format=torch.contiguous_format),
                                                           flow.clone(memory_format=torch.contiguous_format)))
    assert torch.equal(recon_metrics(i0, i1), recon_metrics(i0.clone(memory_format=torch.contiguous_format),
                                                            i1.clone(memory_format=torch.contiguous_format)))


def test_ssim_vs_oracle(gpu):
    """ReconLoss.evaluate's 'ssim' (pytorch_msssim.SSIM, data_range 1): the HIP kernel against the CPU restatement of the
    package's algorithm (oracle.ssim; fp32 like the package, and fp64) -- the package itself is absent offline, so this metric is
    unpinned by the reference.  Ragged sizes (tile edges), the minimum 11 x 11 image, several planes, identical images = 1."""
    from cista_flow_amd.loss import ReconLoss, SSIM, ssim_metrics
    from oracle import cista_oracle as orc
    gen = torch.Generator().manual_seed(21)
    for (B, C, H, W) in ((1, 1, 180, 240), (3, 1, 67, 93), (2, 2, 11, 11), (1, 1, 27, 140)):
        X = torch.rand(B, C, H, W, generator=gen)
        Y = (X + 0.08 * torch.randn(B, C, H, W, generator=gen)).clamp(0, 1)
        got = ssim_metrics(X.to(gpu), Y.to(gpu)).cpu()
        r32, r64 = orc.ssim(X, Y), orc.ssim(X, Y, dtype=torch.float64)
        assert abs(float(got[0]) - r64[0]) < 2e-5 and abs(float(got[1]) - r64[1]) < 2e-5, ((B, C, H, W), got, r64)
        assert abs(float(got[0]) - r32[0]) < 2e-5
        assert torch.equal(got, ssim_metrics(X.to(gpu), Y.to(gpu)).cpu())          # deterministic
        assert abs(float(ssim_metrics(X.to(gpu), X.to(gpu))[0]) - 1.0) < 1e-6
    X, Y = torch.rand(2, 1, 64, 80, generator=gen), torch.rand(2, 1, 64, 80, generator=gen)
    m = ReconLoss(None).evaluate(X.to(gpu), Y.to(gpu))
    assert abs(m["ssim"] - orc.ssim(X, Y)[0]) < 2e-5 and abs(float(SSIM()(X.to(gpu), Y.to(gpu))) - m["ssim"]) < 1e-6
    with pytest.raises(ValueError):
        ssim_metrics(torch.zeros(1, 1, 10, 40, device=gpu), torch.zeros(1, 1, 10, 40, device=gpu))
