"""FrameWarp -- forward / backward flow warp on the HIP path (reference: utils/flow_utils.py:193-221).

warp_frame(I, flow) = grid_sample(I, g, bilinear, align_corners=True, padding_mode='reflection') with
g = 2*((x -/+ u)/W - 0.5), i.e. the W (not W-1) normalisation quirk of flow_utils.py:114-115,184-185
is reproduced bit for bit in csrc/pointwise.hip::warp_kernel.
"""
import torch

from .. import lib as _lib


class FrameWarp(object):
    def __init__(self, mode):
        """mode: 'forward' or 'backward' (reference: anything but 'forward' means backward)."""
        self.mode = mode
        self.flowWarp_dict = dict()   # kept for attribute compatibility (reference caches per (W,H))

    def warp_frame(self, I, flow):
        _lib.check_f32_cuda(I, "I")
        _lib.check_f32_cuda(flow, "flow")
        if I.dim() != 4 or flow.dim() != 4 or flow.shape[1] != 2 or flow.shape[0] != I.shape[0]:
            raise ValueError("warp_frame expects I [B,C,H,W] and flow [B,2,Hf,Wf]")
        B, Cc, H, W = I.shape
        # the reference needs flow at the image's resolution; (Hf,Wf) != (H,W) is the fused
        # interpolate(align_corners=True) path used for states[1]
        Hf, Wf = flow.shape[-2:]
        L = _lib.load()
        src = I.contiguous(memory_format=torch.channels_last) if Cc > 1 else I.contiguous()
        fl = flow.contiguous()
        out = torch.empty_like(src)
        if fl.device != src.device:
            raise RuntimeError("I and flow must live on the same GPU")
        with torch.cuda.device(src.device):    # the stateless entry points launch on the CURRENT device
            rc = L.cf_warp(None, _lib.ptr(src), _lib.ptr(fl), _lib.ptr(out), B, Cc, H, W, Hf, Wf,
                           _lib.CF_WARP_FORWARD if self.mode == 'forward' else _lib.CF_WARP_BACKWARD,
                           _lib.current_stream_ptr(src.device))
        if rc != 0:
            raise RuntimeError("cf_warp failed (%d)" % rc)
        return out
