"""E-RAFT on the MI355X hot path (reference: ERAFT/eraft.py:37-178).

Same constructor, attributes, state_dict layout and forward signature; forward = one cf_flow_forward call:
shared instance-norm encoder on both voxel grids, all-pairs correlation volume + pyramid, `iters` x (lookup ->
motion encoder -> SepConvGRU -> flow head), mask head + learned convex x8 up-sampling, un-pad.
"""
from argparse import Namespace

import torch
import torch.nn as nn

from .. import lib as _lib
from ..runtime import HipBackend
from ..utils.image_process import ImagePadder
from .extractor import BasicEncoder
from .update import BasicUpdateBlock


def get_args():
    return Namespace(small=False, dropout=False, mixed_precision=False, clip=1.0)


class ERAFT(nn.Module):
    def __init__(self, cfgs):
        super(ERAFT, self).__init__()
        args = get_args()
        self.args = args
        self.image_dim = cfgs.image_dim
        self.image_padder = ImagePadder(image_dim=cfgs.image_dim, min_size=32)
        self.subtype = 'standard'
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        args.corr_levels = 4
        args.corr_radius = 4
        self.event_bins = cfgs.num_bins
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=0, n_first_channels=self.event_bins)
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn='batch', dropout=0, n_first_channels=self.event_bins)
        self.update_block = BasicUpdateBlock(self.args, hidden_dim=hdim)
        # True: like the reference, every iteration's mask head + convex up-sampling is evaluated and returned in
        # flow_preds; False: only the last one (the 11 others are dead work at inference)
        self.return_flow_preds = True
        self._backends = {}

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _backend(self, iters):
        be = self._backends.get(iters)
        if be is None:
            be = HipBackend(self, _lib.CF_MODE_ERAFT, self.image_dim, num_bins=self.event_bins, iters=iters)
            self._backends[iters] = be
        return be

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True):
        """image1 / image2: old / new event voxel grids [B,bins,H,W] (eraft.py:114)."""
        if self.training and any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.cnet.modules()):
            raise RuntimeError("ERAFT (HIP) is inference-only: call .eval() first")
        H, W = self.image_dim
        B = image1.shape[0]
        _lib.check_f32_cuda(image1, "image1", (B, self.event_bins, H, W))
        _lib.check_f32_cuda(image2, "image2", (B, self.event_bins, H, W))
        Hp, Wp = self.image_padder.padded_size()
        h8, w8 = Hp // 8, Wp // 8
        if flow_init is not None:
            _lib.check_f32_cuda(flow_init, "flow_init", (B, 2, h8, w8))
            flow_init = flow_init.contiguous()
        dev = image1.device
        h = self._backend(int(iters)).get(B, dev)
        a, b = image1.contiguous(), image2.contiguous()
        flow_final = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        flow_low = torch.empty((B, 2, h8, w8), dtype=torch.float32, device=dev)
        preds = torch.empty((iters, B, 2, Hp, Wp), dtype=torch.float32, device=dev) if self.return_flow_preds else None
        h.check(h.lib.cf_flow_forward(h.h, _lib.ptr(a), _lib.ptr(b), _lib.ptr(flow_init), _lib.ptr(flow_final),
                                      _lib.ptr(flow_low), _lib.ptr(preds), _lib.current_stream_ptr(dev)), "cf_flow_forward")
        return dict(flow_preds=[preds[i] for i in range(iters)] if preds is not None else [],
                    flow_init=flow_low, flow_final=flow_final)
