"""CISTA-LSTC and the flow-compensated wrappers on the MI355X hot path (reference: e2v/e2v_model.py).

Same class names, constructor arguments, sub-module attribute names, state_dict keys and forward
signatures as the reference, so `from e2v.e2v_model import *` in test_with_flow.py / test_wo_flow.py can
point here unchanged.  forward() is one asynchronous call into libcistaflow:

    CistaLSTCNet.forward        -> cf_cista_forward   (e2v_model.py:49-98)
    DCEIFlowCistaNet.forward    -> cf_step            (e2v_model.py:144-196)

PyTorch only allocates the output tensors and carries pointers.  Recurrent states are returned as
logical [B,C,h,w] tensors in channels_last (NHWC) memory -- the layout the kernels compute in -- so the
frame-to-frame recurrence never reshuffles them.
"""
import torch
import torch.nn as nn

from .. import lib as _lib
from ..runtime import HipBackend, empty_nhwc, nhwc_state
from ..utils.flow_utils import FrameWarp
from ..DCEIFlow.DCEIFlow import DCEIFlow
from ..ERAFT.eraft import ERAFT
from ..idn.idedeq import IDEDEQIDO
from .base_layers import *   # noqa: F401,F403  (the reference re-exports the layer library the same way)
from .base_layers import ConvLayer, ConvLSTC, IstaBlock, RecurrentConvLayer, UpsampleConvLayer


class CistaLSTCNet(nn.Module):
    def __init__(self, image_dim, base_channels=64, depth=5, num_bins=5):
        super(CistaLSTCNet, self).__init__()
        self.num_bins = num_bins
        self.depth = depth
        self.base_channels = base_channels
        self.height, self.width = image_dim
        self.num_states = 3
        self.We = ConvLayer(in_channels=self.num_bins, out_channels=int(base_channels / 2), kernel_size=3, stride=1, padding=1, groups=1)
        self.Wi = ConvLayer(in_channels=1, out_channels=int(base_channels / 2), kernel_size=3, stride=1, padding=1)
        self.W0 = ConvLayer(in_channels=base_channels, out_channels=base_channels, kernel_size=3, stride=2, padding=1)
        self.P0 = ConvLSTC(x_size=base_channels, z_size=2 * base_channels, output_size=2 * base_channels, kernel_size=3)
        lista_block = IstaBlock(base_channels=base_channels, is_recurrent=False)
        # the SAME block `depth` times: lista_blocks.0..depth-1 alias one storage (e2v_model.py:34-35)
        self.lista_blocks = nn.ModuleList([lista_block for i in range(self.depth)])
        self.Dg = RecurrentConvLayer(in_channels=2 * base_channels, out_channels=base_channels, kernel_size=3, stride=1, padding=1, activation='relu')
        self.upsamp_conv = UpsampleConvLayer(in_channels=base_channels, out_channels=base_channels, kernel_size=3, stride=1, padding=0, activation='relu')
        self.final_conv = ConvLayer(in_channels=base_channels, out_channels=1, kernel_size=3, stride=1, padding=1)
        self.sigmoid = nn.Sigmoid()
        self._backend = None

    def _be(self):
        if self._backend is None:
            self._backend = HipBackend(self, _lib.CF_MODE_CISTA, (self.height, self.width), num_bins=self.num_bins,
                                       base_channels=self.base_channels, depth=self.depth)
        return self._backend

    def state_shapes(self, B):
        h, w, c = self.height // 2, self.width // 2, self.base_channels
        return (B, 2 * c, h, w), (B, c, h, w)

    def unpack_states(self, prev_states, B):
        """reference layout: [c (2c ch), z (2c ch), (h, cc) (c ch each)] or None / list of None."""
        s2, s1 = self.state_shapes(B)
        c_prev = z_prev = h_prev = cc_prev = None
        if prev_states is not None:
            if len(prev_states) != self.num_states:
                raise ValueError("prev_states must have %d entries" % self.num_states)
            if prev_states[0] is not None:
                c_prev = nhwc_state(prev_states[0], "states[0]", s2)
            if prev_states[1] is not None:
                z_prev = nhwc_state(prev_states[1], "states[1]", s2)
            if prev_states[2] is not None:
                h_prev = nhwc_state(prev_states[2][0], "states[2][0]", s1)
                cc_prev = nhwc_state(prev_states[2][1], "states[2][1]", s1)
        return c_prev, z_prev, h_prev, cc_prev

    def forward(self, events, prev_image, prev_states):
        """events [B,bins,H,W], prev_image [B,1,H,W], prev_states None | [c, z, (h, cc)]
        -> rec_I [B,1,H,W], states [c, z, (h, cc)]   (e2v_model.py:49-98)"""
        B = events.shape[0]
        H, W = self.height, self.width
        _lib.check_f32_cuda(events, "events", (B, self.num_bins, H, W))
        _lib.check_f32_cuda(prev_image, "prev_image", (B, 1, H, W))
        dev = events.device
        c_prev, z_prev, h_prev, cc_prev = self.unpack_states(prev_states, B)
        s2, s1 = self.state_shapes(B)
        I = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        c, z = empty_nhwc(*s2, dev), empty_nhwc(*s2, dev)
        hh, cc = empty_nhwc(*s1, dev), empty_nhwc(*s1, dev)
        h = self._be().get(B, dev)
        p = _lib.ptr
        # contiguous copies stay bound until the call returns: ptr() keeps only the address, and a freed temporary's block
        # would be handed to the next .contiguous() (two arguments aliasing one buffer)
        ev_c, img_c = events.contiguous(), prev_image.contiguous()
        h.check(h.lib.cf_cista_forward(h.h, p(ev_c), p(img_c), p(c_prev), p(z_prev),
                                       p(h_prev), p(cc_prev), p(I), p(c), p(z), p(hh), p(cc),
                                       _lib.current_stream_ptr(dev)), "cf_cista_forward")
        return I, [c, z, (hh, cc)]


class BaseFlowRec(nn.Module):
    def __init__(self, args):
        super(BaseFlowRec, self).__init__()
        self.image_dim = args.image_dim
        self.num_bins = args.num_bins
        self.warp_mode = args.warp_mode
        self.frame_warp = FrameWarp(mode=args.warp_mode)
        self.fix_net_name = None
        self.scale_factor = 0.5
        self.cista_net = CistaLSTCNet(image_dim=args.image_dim, base_channels=args.base_channels, depth=args.depth, num_bins=args.num_bins)
        self.event_flownet = None

    def fix_params(self, net_name):
        raise NotImplementedError("fix_params is training-only (e2v_model.py:116-133); this build is the inference hot path")


class _HipFlowRec(BaseFlowRec):
    """Shared forward of the flow-compensated wrappers: one cf_step call (a5)."""
    _mode = None
    flow_iters = 0

    def _be(self):
        if self._backend is None:
            self._backend = HipBackend(self, self._mode, self.image_dim, num_bins=self.num_bins,
                                       base_channels=self.cista_net.base_channels, depth=self.cista_net.depth,
                                       iters=self.flow_iters, warp_mode=self.warp_mode)
        return self._backend

    def _step(self, in0, in1, rec0, states, flow_init, gt_flow):
        H, W = self.image_dim
        B = rec0.shape[0]
        fnet = self.event_flownet
        Hp, Wp = fnet.image_padder.padded_size()
        h8, w8 = Hp // 8, Wp // 8
        if flow_init is not None:
            _lib.check_f32_cuda(flow_init, "flow_init", (B, 2, h8, w8))
            flow_init = flow_init.contiguous()
        if gt_flow is not None:
            _lib.check_f32_cuda(gt_flow, "gt_flow", (B, 2, H, W))
            gt_flow = gt_flow.contiguous()
        dev = rec0.device
        cn = self.cista_net
        c_prev, z_prev, h_prev, cc_prev = cn.unpack_states(states, B)
        s2, s1 = cn.state_shapes(B)
        iters = self.flow_iters
        I = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        flow_final = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        flow_low = torch.empty((B, 2, h8, w8), dtype=torch.float32, device=dev)
        preds = torch.empty((iters, B, 2, Hp, Wp), dtype=torch.float32, device=dev) if fnet.return_flow_preds else None
        z_warp = empty_nhwc(*s2, dev) if z_prev is not None else None
        c, z = empty_nhwc(*s2, dev), empty_nhwc(*s2, dev)
        hh, cc = empty_nhwc(*s1, dev), empty_nhwc(*s1, dev)
        h = self._be().get(B, dev)
        p = _lib.ptr
        in0_c, in1_c, rec0_c = in0.contiguous(), in1.contiguous(), rec0.contiguous()   # bound until the call returns
        h.check(h.lib.cf_step(h.h, p(in0_c), p(in1_c), p(rec0_c), p(flow_init),
                              p(gt_flow), p(c_prev), p(z_prev), p(h_prev), p(cc_prev), p(I), p(flow_final), p(flow_low),
                              p(preds), p(z_warp), p(c), p(z), p(hh), p(cc), _lib.current_stream_ptr(dev)), "cf_step")
        if z_warp is not None:
            states[1] = z_warp      # e2v_model.py:191 (pass-through copy when flow_final is all zero)
        batch_flow = dict(flow_preds=[preds[i] for i in range(iters)] if preds is not None else [],
                          flow_init=flow_low, flow_final=flow_final)
        return I, batch_flow, [c, z, (hh, cc)]


class DCEIFlowCistaNet(_HipFlowRec):
    '''CISTA-Flow: CISTA-LSTC + DCEIFlow  (e2v_model.py:138-196)'''
    _mode = _lib.CF_MODE_EIFLOW

    def __init__(self, args):
        super(DCEIFlowCistaNet, self).__init__(args)
        self.event_flownet = DCEIFlow(num_bins=self.num_bins, args=args)
        self.flow_iters = 6          # DCEIFlow.forward default iters (DCEIFlow.py:143)
        self._backend = None

    def forward(self, batch_data, states, batch_gt=dict([])):
        '''batch_data: event_voxel [B,bins,H,W], rec_img0 [B,1,H,W], optional flow_init;
        states: None | [c, z, (h, cc)]; batch_gt: optional gt_img0 (flow-net image), gt_flow (warp override).
        Returns (I_rec, batch_flow dict, states).  Like the reference, a non-None `states` list is mutated:
        states[1] becomes the warped sparse code (e2v_model.py:191).'''
        if 'event_voxel_bw' in batch_data or 'gt_img1' in batch_gt:
            raise NotImplementedError("event_voxel_bw / gt_img1 feed the training-only bilateral branch")
        ev = batch_data['event_voxel']
        rec0 = batch_data['rec_img0']
        img_flow = batch_gt['gt_img0'] if 'gt_img0' in batch_gt else rec0
        H, W = self.image_dim
        B = ev.shape[0]
        _lib.check_f32_cuda(ev, "event_voxel", (B, self.num_bins, H, W))
        _lib.check_f32_cuda(rec0, "rec_img0", (B, 1, H, W))
        _lib.check_f32_cuda(img_flow, "gt_img0", (B, 1, H, W))
        return self._step(ev, img_flow, rec0, states, batch_data.get('flow_init'), batch_gt.get('gt_flow'))


class ERAFTCistaNet(_HipFlowRec):
    '''CISTA-Flow: CISTA-LSTC + E-RAFT  (e2v_model.py:200-248)'''
    _mode = _lib.CF_MODE_ERAFT

    def __init__(self, args):
        super(ERAFTCistaNet, self).__init__(args)
        self.event_flownet = ERAFT(args)
        self.flow_iters = 12         # ERAFT.forward default iters (eraft.py:114)
        self._backend = None
        # Opt-in: reuse fnet(event_voxel_old) from the previous call when event_voxel_old IS the tensor that
        # was event_voxel then.  The check is object identity + tensor._version, which cannot see writes that bypass
        # autograd's version counter (numpy / DLPack / __cuda_array_interface__ views, other libraries' kernels,
        # .data-level copies): enable only when the driver never writes into a voxel grid after handing it over.
        self.reuse_prev_features = False
        self._last_ev = None

    def forward(self, batch_data, states, batch_gt=dict([])):
        '''batch_data: event_voxel_old, event_voxel [B,bins,H,W], rec_img0 [B,1,H,W]; batch_gt: optional gt_flow.'''
        ev_old = batch_data['event_voxel_old']
        ev = batch_data['event_voxel']
        rec0 = batch_data['rec_img0']
        H, W = self.image_dim
        B = ev.shape[0]
        _lib.check_f32_cuda(ev_old, "event_voxel_old", (B, self.num_bins, H, W))
        _lib.check_f32_cuda(ev, "event_voxel", (B, self.num_bins, H, W))
        _lib.check_f32_cuda(rec0, "rec_img0", (B, 1, H, W))
        # The driver carries `evs_old = evs` (test_with_flow.py:144-149): when event_voxel_old IS the tensor object that
        # was event_voxel in the previous call (and has not been written since), fnet's feature map of it is still in
        # the handle and is reused.  Holding the reference keeps that memory from being recycled under us.
        last = getattr(self, "_last_ev", None)
        reuse = (self.reuse_prev_features and last is not None and last[0] is ev_old and last[1] == ev_old._version
                 and last[2] == B and states is not None)
        h = self._be().get(B, ev.device)
        h.check(h.lib.cf_hint_prev_grid(h.h, 1 if reuse else 0), "cf_hint_prev_grid")
        out = self._step(ev_old, ev, rec0, states, None, batch_gt.get('gt_flow'))
        self._last_ev = (ev, ev._version, B)
        return out


class IDCistaNet(_HipFlowRec):
    '''CISTA-Flow: CISTA-LSTC + IDNet  (e2v_model.py:252-308)'''
    _mode = _lib.CF_MODE_IDNET

    def __init__(self, args):
        super(IDCistaNet, self).__init__(args)
        from types import SimpleNamespace
        # the reference builds this with OmegaConf.create; IDEDEQIDO only reads it through getattr
        config = SimpleNamespace(update_iters=1, pred_next_flow=True, image_dim=args.image_dim, num_bins=args.num_bins)
        self.event_flownet = IDEDEQIDO(config)
        self.flow_iters = 1
        self._backend = None

    def forward(self, batch_data, states, flow_init=None, batch_gt=dict([])):
        '''batch_data: event_voxel [B,bins,H,W], rec_img0 [B,1,H,W]; flow_init: padded [B,2,Hp,Wp] = the previous
        frame's batch_flow['next_flow'] (test_with_flow.py:151-154) or None.'''
        ev = batch_data['event_voxel']
        rec0 = batch_data['rec_img0']
        gt_flow = batch_gt.get('gt_flow')
        H, W = self.image_dim
        B = ev.shape[0]
        _lib.check_f32_cuda(ev, "event_voxel", (B, self.num_bins, H, W))
        _lib.check_f32_cuda(rec0, "rec_img0", (B, 1, H, W))
        fnet = self.event_flownet
        Hp, Wp = fnet.image_padder.padded_size()
        if flow_init is not None:
            _lib.check_f32_cuda(flow_init, "flow_init", (B, 2, Hp, Wp))
            flow_init = flow_init.contiguous()
        if gt_flow is not None:
            _lib.check_f32_cuda(gt_flow, "gt_flow", (B, 2, H, W))
            gt_flow = gt_flow.contiguous()
        dev = ev.device
        cn = self.cista_net
        c_prev, z_prev, h_prev, cc_prev = cn.unpack_states(states, B)
        s2, s1 = cn.state_shapes(B)
        I = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
        flow_final = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        next_flow = torch.empty((B, 2, Hp, Wp), dtype=torch.float32, device=dev)
        hist = torch.empty((2, B, 2, Hp, Wp), dtype=torch.float32, device=dev)
        z_warp = empty_nhwc(*s2, dev) if z_prev is not None else None
        c, z = empty_nhwc(*s2, dev), empty_nhwc(*s2, dev)
        hh, cc = empty_nhwc(*s1, dev), empty_nhwc(*s1, dev)
        h = self._be().get(B, dev)
        p = _lib.ptr
        evc, rec0_c = ev.contiguous(), rec0.contiguous()                                # bound until the call returns
        h.check(h.lib.cf_step(h.h, p(evc), p(evc), p(rec0_c), p(flow_init), p(gt_flow), p(c_prev), p(z_prev),
                              p(h_prev), p(cc_prev), p(I), p(flow_final), p(next_flow), p(hist), p(z_warp), p(c), p(z),
                              p(hh), p(cc), _lib.current_stream_ptr(dev)), "cf_step")
        if z_warp is not None:
            states[1] = z_warp
        d0 = flow_init if flow_init is not None else torch.zeros_like(hist[1])
        batch_flow = {'flow_final': flow_final, 'next_flow': next_flow, 'delta_flow': torch.stack([d0, hist[1]], 1),
                      'flow_preds': [hist[0]]}
        return I, batch_flow, [c, z, (hh, cc)]
