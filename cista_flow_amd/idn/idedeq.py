"""IDNet (iterative deblurring, no correlation volume) on the MI355X hot path (reference: idn/idedeq.py:13-227).

Same constructor (a config object read with getattr), attributes, state_dict layout and forward signature;
forward = one cf_flow_forward call: per-bin deblur warp, LiteEncoder on all B*5 bins in one batch, ConvGRU over
the bins, two flow heads with learned convex x8 up-sampling, flow_total = flow_init + delta_flow.
Built for the configuration e2v_model.py:256-262 uses: update_iters=1, pred_next_flow=True, downsample=8.
"""
import torch
import torch.nn as nn

from .. import lib as _lib
from ..runtime import HipBackend
from ..utils.image_process import ImagePadder
from .extractor import LiteEncoder
from .update import LiteUpdateBlock


class IDEDEQIDO(nn.Module):
    def __init__(self, config):
        super(IDEDEQIDO, self).__init__()
        self.image_dim = config.image_dim
        self.image_padder = ImagePadder(image_dim=config.image_dim, min_size=32)
        self.hidden_dim = getattr(config, 'hidden_dim', 96)
        self.input_dim = 64
        self.downsample = getattr(config, 'downsample', 8)
        self.input_flowmap = getattr(config, 'input_flowmap', False)
        self.pred_next_flow = getattr(config, 'pred_next_flow', False)
        self.deblur_iters = getattr(config, 'update_iters', 1)
        self.deblur = getattr(config, "deblur", True)
        self.deblur_mode = getattr(config, "deblur_mode", "voxel")
        if (self.hidden_dim != 96 or self.downsample != 8 or self.input_flowmap or not self.pred_next_flow
                or self.deblur_iters != 1 or not self.deblur or self.deblur_mode != "voxel"
                or getattr(config, "co_mode", False) or getattr(config, "conr_mode", False)):
            raise NotImplementedError("the HIP path is built for CISTA-Flow's IDNet config (e2v_model.py:256-262)")
        self.fnet = LiteEncoder(output_dim=self.input_dim // 2, dropout=0, n_first_channels=2, stride=2)
        self.update_net = LiteUpdateBlock(hidden_dim=self.hidden_dim, input_dim=self.input_dim, num_outputs=2,
                                          downsample=self.downsample)
        self.cnet = None
        self.num_bins = getattr(config, 'num_bins', 5)
        self._backend = None

    def _be(self):
        if self._backend is None:
            self._backend = HipBackend(self, _lib.CF_MODE_IDNET, self.image_dim, num_bins=self.num_bins, iters=1)
        return self._backend

    def forward(self, event_bins, flow_init=None, deblur_iters=None, net_co=None):
        """event_bins [B,5,H,W]; flow_init: padded [B,2,Hp,Wp] (the previous call's next_flow) or None."""
        if net_co is not None or (deblur_iters is not None and deblur_iters != 1):
            raise NotImplementedError("net_co / deblur_iters != 1 are not used by CISTA-Flow")
        H, W = self.image_dim
        B = event_bins.shape[0]
        _lib.check_f32_cuda(event_bins, "event_bins", (B, self.num_bins, H, W))
        Hp, Wp = self.image_padder.padded_size()
        if flow_init is not None:
            _lib.check_f32_cuda(flow_init, "flow_init", (B, 2, Hp, Wp))
            flow_init = flow_init.contiguous()
        dev = event_bins.device
        h = self._be().get(B, dev)
        flow_final = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        next_flow = torch.empty((B, 2, Hp, Wp), dtype=torch.float32, device=dev)
        hist = torch.empty((2, B, 2, Hp, Wp), dtype=torch.float32, device=dev)
        bins_c = event_bins.contiguous()                   # bound until the call returns (ptr() keeps only the address)
        h.check(h.lib.cf_flow_forward(h.h, _lib.ptr(bins_c), None, _lib.ptr(flow_init), _lib.ptr(flow_final),
                                      _lib.ptr(next_flow), _lib.ptr(hist), _lib.current_stream_ptr(dev)), "cf_flow_forward")
        d0 = flow_init if flow_init is not None else torch.zeros_like(hist[1])
        return {'flow_final': flow_final, 'next_flow': next_flow, 'delta_flow': torch.stack([d0, hist[1]], 1),
                'flow_preds': [hist[0]]}
