"""Output stage of the drivers (reference: utils/data_io.py:9-29,105-192; callers test_with_flow.py:174-178).

`merge_optical_flow` -- FlowWriter's flow -> HSV -> BGR colour coding -- runs on the GPU (cf_flow_to_bgr); the writers keep the
reference's names, constructor arguments (cfgs.output_folder / is_write_image / is_write_flow, model_name, dataset_name), file names
(frame_%010d.png, flow/flow_%010d.png) and call signatures, and encode the PNGs on the host with PIL (the reference uses PIL for frames
and cv2.imwrite for the colour-coded flow; cv2 is not a dependency here).

UNPINNED: cv2 is not installed where the golden vectors are generated, so the colour coding follows OpenCV's PUBLISHED arithmetic
(cartToPolar's angle in [0, 2 pi), numpy's truncating uint8 casts, the 8-bit HSV -> BGR of cvtColor) and is tested against a numpy
restatement of that (oracle.cista_oracle.merge_optical_flow), not against cv2 itself."""
import os

import numpy as np


def merge_optical_flow(flow):
    """flow: [2, H, W] or [B, 2, H, W] float32 (CUDA tensor, or numpy / CPU tensor that is moved to the GPU) -> BGR uint8 numpy array
    [H, W, 3] / [B, H, W, 3], the array the reference hands to cv2.imwrite (utils/data_io.py:9-29)."""
    return merge_optical_flow_gpu(flow).cpu().numpy()


def merge_optical_flow_gpu(flow):
    """As merge_optical_flow, but the result stays on the device (uint8 tensor): 3 bytes per pixel cross PCIe instead of 8."""
    import torch
    from .. import lib as _lib
    L = _lib.load()
    if not isinstance(flow, torch.Tensor):
        flow = torch.as_tensor(np.asarray(flow), dtype=torch.float32)
    if not flow.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("merge_optical_flow runs on the GPU only (cista_flow_amd has no CPU path)")
        flow = flow.cuda()
    single = flow.dim() == 3
    x = (flow[None] if single else flow).contiguous()
    if x.dim() != 4 or x.shape[1] != 2:
        raise ValueError("flow must be [2,H,W] or [B,2,H,W], got %s" % (tuple(flow.shape),))
    _lib.check_f32_cuda(x, "flow")
    B, _, H, W = x.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=x.device)
    scratch = torch.empty((B,), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.cf_flow_to_bgr(_lib.ptr(x), B, H, W, out.data_ptr(), scratch.data_ptr(), _lib.current_stream_ptr(x.device))
    if rc != 0:
        raise RuntimeError("cf_flow_to_bgr failed (%d)" % rc)
    return out[0] if single else out


class Writer(object):
    """utils/data_io.py:105-117: output_folder / model_name [/ dataset_name]."""

    def __init__(self, cfgs, model_name, dataset_name=None):
        self.output_folder = cfgs.output_folder
        self.dataset_name = dataset_name
        self.output_data_folder = os.path.join(self.output_folder, model_name)
        if dataset_name is not None:
            self.output_data_folder = os.path.join(self.output_data_folder, dataset_name)


class ImageWriter(Writer):
    """utils/data_io.py:139-161: frame_%010d.png of np.uint8(img)."""

    def __init__(self, cfgs, model_name, dataset_name=None):
        super(ImageWriter, self).__init__(cfgs, model_name, dataset_name)
        self.is_write_image = cfgs.is_write_image
        if self.is_write_image and not os.path.exists(self.output_data_folder):
            os.makedirs(self.output_data_folder)

    def __call__(self, img, img_id):
        if not self.is_write_image:
            return
        from PIL import Image
        if hasattr(img, "cpu"):
            img = img.detach().cpu().numpy()
        Image.fromarray(np.uint8(img)).save(os.path.join(self.output_data_folder, 'frame_{:010d}.png'.format(img_id)))


class FlowWriter(Writer):
    """utils/data_io.py:164-192: flow/flow_%010d.png of merge_optical_flow(flow) (BGR array, written as cv2.imwrite would: the file's
    channels are R, G, B)."""

    def __init__(self, cfgs, model_name, dataset_name=None):
        super(FlowWriter, self).__init__(cfgs, model_name, dataset_name)
        self.is_write_flow = cfgs.is_write_flow
        if self.is_write_flow:
            self.output_data_folder = os.path.join(self.output_data_folder, 'flow')
            if not os.path.exists(self.output_data_folder):
                os.makedirs(self.output_data_folder)

    def __call__(self, flow, img_id):
        if not self.is_write_flow:
            return
        from PIL import Image
        bgr = merge_optical_flow(flow)
        Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])).save(os.path.join(self.output_data_folder, 'flow_{:010d}.png'.format(img_id)))
