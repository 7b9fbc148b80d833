"""Evaluation metrics of the reference's drivers on the GPU (reference: loss.py).

Same class / function names and return conventions as the reference (`ReconLoss.evaluate` and
`FlowL1LossDict.evaluate` return dicts of Python floats, `voxel_warping_flow_loss` a 0-dim tensor), so
`from loss import ReconLoss, voxel_warping_flow_loss` (test_wo_flow.py:20) can point here.  The arithmetic runs in
csrc/metrics.hip; only a few doubles per frame cross PCIe.

SSIM follows pytorch_msssim's published algorithm (the package is absent offline, so this one metric is UNPINNED by the
reference; oracle/cista_oracle.py::ssim is the CPU restatement it is tested against).  Not built: LPIPS (`lpips` +
torchvision AlexNet weights from the network) -- `ReconLoss.evaluate` returns `mse`, `psnr`, `ssim` -- and the training
losses (`forward` methods).
"""
import torch
import torch.nn as nn

from . import lib as _lib


def _scratch(dev):
    L = _lib.load()
    return torch.empty(int(L.cf_metrics_scratch_doubles()), dtype=torch.float64, device=dev)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d)" % (what, rc))


def recon_metrics(rec_img, target_img):
    """-> float64 CUDA tensor [2] = (mse, psnr); asynchronous (no host sync)."""
    _lib.check_f32_cuda(rec_img, "rec_img")
    _lib.check_f32_cuda(target_img, "target_img", tuple(rec_img.shape))
    a, b = rec_img.contiguous(), target_img.contiguous()
    dev = a.device
    out = torch.empty(2, dtype=torch.float64, device=dev)
    L = _lib.load()
    scratch = _scratch(dev)
    with torch.cuda.device(dev):
        _check(L.cf_metrics_recon(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(out), _lib.ptr(scratch),
                                  _lib.current_stream_ptr(dev)), "cf_metrics_recon")
    return out


def ssim_metrics(rec_img, target_img):
    """-> float64 CUDA tensor [2] = (ssim, cs) of pytorch_msssim.SSIM(data_range=1, size_average=True); asynchronous."""
    _lib.check_f32_cuda(rec_img, "rec_img")
    _lib.check_f32_cuda(target_img, "target_img", tuple(rec_img.shape))
    if rec_img.dim() != 4 or rec_img.shape[2] < 11 or rec_img.shape[3] < 11:
        raise ValueError("ssim: expected [B,C,H,W] with H, W >= 11 (the 11-tap gaussian window is applied without padding)")
    a, b = rec_img.contiguous(), target_img.contiguous()
    dev = a.device
    out = torch.empty(2, dtype=torch.float64, device=dev)
    scratch = _scratch(dev)
    L = _lib.load()
    with torch.cuda.device(dev):
        _check(L.cf_metrics_ssim(_lib.ptr(a), _lib.ptr(b), a.shape[0] * a.shape[1], a.shape[2], a.shape[3], _lib.ptr(out),
                                 _lib.ptr(scratch), _lib.current_stream_ptr(dev)), "cf_metrics_ssim")
    return out


class SSIM(nn.Module):
    """pytorch_msssim.SSIM as loss.py:314 builds it (data_range=1, size_average=True, win 11 / sigma 1.5, K (0.01, 0.03))."""

    def __init__(self, data_range=1, size_average=True, channel=1, nonnegative_ssim=False):
        super().__init__()
        if data_range != 1 or not size_average or nonnegative_ssim:
            raise NotImplementedError("SSIM: only the configuration CISTA-Flow uses (data_range=1, size_average=True)")
        self.channel = channel

    def forward(self, X, Y):
        return ssim_metrics(X, Y)[0].float()


class PSNR(nn.Module):
    """loss.py:15-24 (data_range 1 only, as ReconLoss builds it)."""

    def __init__(self, data_range=1):
        super().__init__()
        if data_range != 1:
            raise NotImplementedError("PSNR: only data_range=1 is used by CISTA-Flow")
        self.data_range = data_range

    def forward(self, imgs1, imgs2):
        return recon_metrics(imgs1, imgs2)[1].float()


class ReconLoss(nn.Module):
    """ReconLoss.evaluate (loss.py:316-328) without 'lpips' (needs network weights)."""

    def __init__(self, frame_warper=None, lpips_net='alex'):
        super().__init__()
        self.warp_fn = frame_warper
        self.psnr_fn = PSNR(data_range=1)
        self.ssim_loss_fn = SSIM(data_range=1, size_average=True, channel=1, nonnegative_ssim=False)

    def evaluate(self, rec_img, target_img):
        m = torch.cat([recon_metrics(rec_img, target_img), ssim_metrics(rec_img, target_img)]).cpu()   # ONE device -> host copy
        return {'mse': float(m[0]), 'psnr': float(m[1]), 'ssim': float(m[2])}

    def forward(self, *a, **k):
        raise NotImplementedError("training loss (loss.py:331-360) is out of scope of the inference hot path")


def flow_metrics(flow_final, gt_flow, gt_img0, gt_img1, flow_valid=None, warp_mode='forward', max_flow=400.0):
    """-> float64 CUDA tensor [6] = (photo_loss, epe, 1px, 3px, 5px, out); asynchronous."""
    _lib.check_f32_cuda(flow_final, "flow_final")
    B, two, H, W = flow_final.shape
    if two != 2:
        raise ValueError("flow_final must be [B,2,H,W]")
    _lib.check_f32_cuda(gt_flow, "gt_flow", (B, 2, H, W))
    _lib.check_f32_cuda(gt_img0, "gt_img0", (B, 1, H, W))
    _lib.check_f32_cuda(gt_img1, "gt_img1", (B, 1, H, W))
    if flow_valid is not None:
        _lib.check_f32_cuda(flow_valid, "flow_valid", (B, 1, H, W))
        flow_valid = flow_valid.contiguous()
    dev = flow_final.device
    out = torch.empty(6, dtype=torch.float64, device=dev)
    L = _lib.load()
    # every contiguous copy and the scratch buffer stay bound to a local until the call has returned: ptr() keeps only the
    # address, and a temporary freed between two arguments would hand its block to the next one (aliased inputs, epe = 0)
    ff, gf, g0, g1, scratch = flow_final.contiguous(), gt_flow.contiguous(), gt_img0.contiguous(), gt_img1.contiguous(), _scratch(dev)
    with torch.cuda.device(dev):
        _check(L.cf_metrics_flow(_lib.ptr(ff), _lib.ptr(gf), _lib.ptr(g0), _lib.ptr(g1), _lib.ptr(flow_valid), B, H, W,
                                 _lib.CF_WARP_FORWARD if warp_mode == 'forward' else _lib.CF_WARP_BACKWARD, float(max_flow),
                                 _lib.ptr(out), _lib.ptr(scratch), _lib.current_stream_ptr(dev)), "cf_metrics_flow")
    return out


class FlowL1LossDict(nn.Module):
    """FlowL1LossDict.evaluate (loss.py:237-265)."""

    def __init__(self, image_dim, frame_warper, ds=8, is_bi=False):
        super().__init__()
        self.gamma = 0.8
        self.isbi = is_bi
        self.max_flow = 400
        self.warp_fn = frame_warper
        self.image_dim = image_dim

    def evaluate(self, flow_final, batch_target):
        m = flow_metrics(flow_final, batch_target['gt_flow'], batch_target['gt_img0'], batch_target['gt_img1'],
                         batch_target.get('flow_valid'), self.warp_fn.mode, self.max_flow).cpu()
        keys = ['photo_loss', 'epe', '1px', '3px', '5px', 'out']
        return {k: float(m[i]) for i, k in enumerate(keys)}

    def forward(self, *a, **k):
        raise NotImplementedError("training loss (loss.py:267-312) is out of scope of the inference hot path")


def fwl_metrics(voxel, flow):
    """-> float64 CUDA tensor [3] = (var with flow, var with zero flow, FWL ratio of test_wo_flow.py:161)."""
    _lib.check_f32_cuda(voxel, "voxel")
    B, C, H, W = voxel.shape
    _lib.check_f32_cuda(flow, "flow", (B, 2, H, W))
    dev = voxel.device
    out = torch.empty(3, dtype=torch.float64, device=dev)
    L = _lib.load()
    vx, fl, scratch = voxel.contiguous(), flow.contiguous(), _scratch(dev)      # bound until the call returns
    with torch.cuda.device(dev):
        _check(L.cf_metrics_fwl(_lib.ptr(vx), _lib.ptr(fl), B, C, H, W, _lib.ptr(out),
                                _lib.ptr(scratch), _lib.current_stream_ptr(dev)), "cf_metrics_fwl")
    return out


def voxel_warping_flow_loss(voxel, displacement, output_images=False, reverse_time=False):
    """loss.py:27-83: variance of the event image obtained by warping voxel channel i along i/(C-1) of the flow."""
    if output_images or reverse_time:
        raise NotImplementedError("output_images / reverse_time are not used by the drivers (test_wo_flow.py:161)")
    return fwl_metrics(voxel, displacement)[0].float()
