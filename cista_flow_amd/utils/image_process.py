"""ImagePadder -- shape logic only (reference: utils/image_process.py:60-107).

The padding itself (zeros on the TOP and LEFT up to a multiple of `min_size`) is folded into the
first convolution of each encoder and the crop into the flow up-sampling kernel; this class keeps
the reference's attributes so callers that inspect pad_height / pad_width keep working.
"""


class ImagePadder(object):
    def __init__(self, image_dim, min_size=64):
        self.height, self.width = image_dim
        if isinstance(min_size, (tuple, list)):
            self.pad_height = (min_size[0] - self.height % min_size[0]) % min_size[0]
            self.pad_width = (min_size[1] - self.width % min_size[1]) % min_size[1]
        else:
            self.pad_height = (min_size - self.height % min_size) % min_size
            self.pad_width = (min_size - self.width % min_size) % min_size
        self.min_size = min_size

    def padded_size(self):
        return self.height + self.pad_height, self.width + self.pad_width

    def pad(self, image):
        """Materialised pad (plumbing helper for callers; the HIP path never calls it)."""
        import torch.nn.functional as F
        return F.pad(image, (self.pad_width, 0, self.pad_height, 0))

    def unpad(self, image):
        return image[..., self.pad_height:, self.pad_width:]


def to_uint8(pred):
    """`np.uint8(pred * 255.)` of the drivers' output stage (test_with_flow.py:174) on the GPU: pred float32 CUDA
    tensor in [0, 1] -> uint8 tensor of the same shape (D2H then moves 1 byte per pixel instead of 4)."""
    import torch
    from .. import lib as _lib
    L = _lib.load()
    _lib.check_f32_cuda(pred, "pred", tuple(pred.shape))
    x = pred.contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):      # the stateless entry points launch on the CURRENT device
        rc = L.cf_quantize_u8(_lib.ptr(x), out.data_ptr(), x.numel(), _lib.current_stream_ptr(x.device))
    if rc != 0:
        raise RuntimeError("cf_quantize_u8 failed (%d)" % rc)
    return out
