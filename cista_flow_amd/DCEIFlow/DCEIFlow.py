"""DCEIFlow on the MI355X hot path (reference: DCEIFlow/DCEIFlow.py:32-44,49-300).

Same constructor, attributes, state_dict layout and forward signature as the reference module; the forward
is one call into libcistaflow (cf_flow_forward): three encoders, EIFusion, all-pairs correlation volume +
pyramid, `iters` x (lookup -> motion encoder -> SepConvGRU -> flow head), x8 bilinear up-flow and un-pad.
Inference branch only: image2 / reversed_event_voxel (bilateral training, DCEIFlow.py:230-270) raise.
"""
import torch
import torch.nn as nn

from .. import lib as _lib
from ..runtime import HipBackend
from ..utils.image_process import ImagePadder
from .core.backbone.raft_encoder import BasicEncoder
from .core.decoder.with_event_updater import BasicUpdateBlockNoMask


class EIFusion(nn.Module):
    def __init__(self, input_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, 192, 1, padding=0)
        self.conv2 = nn.Conv2d(input_dim, 192, 1, padding=0)
        self.convo = nn.Conv2d(192 * 2, input_dim, 3, padding=1)


class DCEIFlow(nn.Module):
    def __init__(self, num_bins, args):
        super().__init__()
        self.image_padder = ImagePadder(image_dim=args.image_dim, min_size=32)
        self.ds = args.ds
        self.is_bi = args.is_bi
        self.args = args
        self.small = False
        self.dropout = 0
        self.alternate_corr = False
        self.event_bins = num_bins
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        self.args.corr_levels = 4
        self.args.corr_radius = 4
        self.args.mixed_precision = False
        if self.ds != 8:
            raise NotImplementedError("only ds=8 is built (utils/configs.py:25 default)")
        self.fnet = BasicEncoder(ds=self.ds, input_dim=1, output_dim=256, norm_fn='instance', dropout=self.dropout)
        self.cnet = BasicEncoder(ds=self.ds, input_dim=1, output_dim=hdim + cdim, norm_fn='batch', dropout=self.dropout)
        self.update_block = BasicUpdateBlockNoMask(self.args, hidden_dim=hdim)
        self.fusion = EIFusion(input_dim=256)
        self.enet = BasicEncoder(ds=self.ds, input_dim=self.event_bins, output_dim=256, norm_fn='instance', dropout=self.dropout)
        self.return_flow_preds = True     # fill batch['flow_preds'] with every iteration's padded flow
        self._backends = {}

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d)):
                m.eval()

    def _backend(self, iters):
        be = self._backends.get(iters)
        if be is None:
            be = HipBackend(self, _lib.CF_MODE_EIFLOW, self.args.image_dim, num_bins=self.event_bins, iters=iters)
            self._backends[iters] = be
        return be

    def forward(self, event_voxel, image1, image2=None, reversed_event_voxel=None, iters=6, flow_init=None, upsample=True):
        """Estimate optical flow from an event voxel grid and the previous frame (DCEIFlow.py:143)."""
        if image2 is not None or reversed_event_voxel is not None:
            raise NotImplementedError("image2 / reversed_event_voxel are training-only branches (DCEIFlow.py:230-270)")
        if self.training and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.cnet.modules()):
            raise RuntimeError("DCEIFlow (HIP) is inference-only: call .eval() first (BatchNorm uses running statistics)")
        H, W = self.args.image_dim
        B = event_voxel.shape[0]
        _lib.check_f32_cuda(event_voxel, "event_voxel", (B, self.event_bins, H, W))
        _lib.check_f32_cuda(image1, "image1", (B, 1, H, W))
        Hp, Wp = self.image_padder.padded_size()
        h8, w8 = Hp // 8, Wp // 8
        if flow_init is not None:
            _lib.check_f32_cuda(flow_init, "flow_init", (B, 2, h8, w8))
            flow_init = flow_init.contiguous()
        dev = event_voxel.device
        h = self._backend(int(iters)).get(B, dev)
        ev, im = event_voxel.contiguous(), image1.contiguous()
        flow_final = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        flow_low = torch.empty((B, 2, h8, w8), dtype=torch.float32, device=dev)
        preds = torch.empty((iters, B, 2, Hp, Wp), dtype=torch.float32, device=dev) if self.return_flow_preds else None
        h.check(h.lib.cf_flow_forward(h.h, _lib.ptr(ev), _lib.ptr(im), _lib.ptr(flow_init), _lib.ptr(flow_final),
                                      _lib.ptr(flow_low), _lib.ptr(preds), _lib.current_stream_ptr(dev)), "cf_flow_forward")
        return dict(flow_preds=[preds[i] for i in range(iters)] if preds is not None else [],
                    flow_init=flow_low, flow_final=flow_final)
