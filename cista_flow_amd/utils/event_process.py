"""events -> voxel grid on the GPU (reference: utils/event_process.py:15-72 events_to_voxel_grid and :193-216
event_preprocess('std')): the step right before the hot path.  Raw events (16-32 B each) are
uploaded instead of dense voxel grids and the temporal-bilinear scatter + non-zero normalisation run as HIP
kernels (float atomics: sums may differ from numpy's sequential np.add.at in the last bit)."""
import torch

from .. import lib as _lib


def events_to_voxel_grid_batch(event_list, num_bins, width, height, normalize=True, filter_hot_pixel=False):
    """event_list: B tensors [N_b, 4] float64 on the GPU, rows (timestamp, x, y, polarity) in time order.
    filter_hot_pixel: event_preprocess(filter_hot_pixel=True) -- zero voxels with |v| > 25 / num_bins before normalising.
    Returns [B, num_bins, height, width] float32."""
    if len(event_list) == 0:
        raise ValueError("empty batch")
    for e in event_list:
        if not (isinstance(e, torch.Tensor) and e.is_cuda and e.dtype == torch.float64 and e.dim() == 2 and e.shape[1] == 4):
            raise TypeError("events must be CUDA float64 tensors of shape [N, 4]")
    dev = event_list[0].device
    B = len(event_list)
    ev = torch.cat([e.contiguous() for e in event_list], 0) if B > 1 else event_list[0].contiguous()
    if ev.shape[0] == 0:
        ev = torch.zeros((1, 4), dtype=torch.float64, device=dev)
    off = [0]
    for e in event_list:
        off.append(off[-1] + int(e.shape[0]))
    offsets = torch.tensor(off, dtype=torch.int64, device=dev)
    voxel = torch.empty((B, num_bins, height, width), dtype=torch.float32, device=dev)
    stats = torch.empty((B, 3), dtype=torch.float64, device=dev)
    L = _lib.load()
    with torch.cuda.device(dev):           # the stateless entry points launch on the CURRENT device
        rc = L.cf_events_to_voxel_ex(_lib.ptr(ev), _lib.ptr(offsets), B, num_bins, height, width, _lib.ptr(voxel), _lib.ptr(stats),
                                     1 if normalize else 0, (25.0 / num_bins) if filter_hot_pixel else 0.0,
                                     _lib.current_stream_ptr(dev))
    if rc != 0:
        raise RuntimeError("cf_events_to_voxel failed (%d)" % rc)
    return voxel


def event_preprocess(event_voxel_grid, mode='std', filter_hot_pixel=False):
    """event_preprocess of the reference (utils/event_process.py:193-216) for grids that already sit on the GPU:
    [bins, H, W] or [B, bins, H, W] float32 CUDA tensor -> new tensor of the same shape; each grid is filtered
    (|v| > 25 / bins -> 0) and normalised to mean 0 / std 1 over its non-zero voxels on its own."""
    if mode != 'std':
        raise NotImplementedError("event_preprocess: only mode='std' is used by CISTA-Flow")
    _lib.check_f32_cuda(event_voxel_grid, "event_voxel_grid")
    if event_voxel_grid.dim() not in (3, 4):
        raise ValueError("event_voxel_grid must be [bins,H,W] or [B,bins,H,W]")
    g = event_voxel_grid.contiguous().clone()
    B = 1 if g.dim() == 3 else g.shape[0]
    bins = g.shape[-3]
    dev = g.device
    stats = torch.empty((B, 3), dtype=torch.float64, device=dev)
    L = _lib.load()
    with torch.cuda.device(dev):
        rc = L.cf_voxel_preprocess(_lib.ptr(g), B, g.numel() // B, _lib.ptr(stats), 1, (25.0 / bins) if filter_hot_pixel else 0.0,
                                   _lib.current_stream_ptr(dev))
    if rc != 0:
        raise RuntimeError("cf_voxel_preprocess failed (%d)" % rc)
    return g
