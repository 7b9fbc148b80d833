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
