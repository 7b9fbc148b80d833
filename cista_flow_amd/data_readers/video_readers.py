"""Frame / event packaging for one reconstruction (reference: data_readers/video_readers.py:10-282, class VR).

`VR` keeps the reference's attributes and method names; a dataset-specific reader overrides `update_frame`,
`update_events` (and `update_flow`) exactly as the reference's ImageReader / VideoReader subclasses do -- those
subclasses themselves (cv2.imread / cv2.VideoCapture / h5py loaders) are file-format plumbing that needs packages
absent here and are not restated.  What IS on the way to the hot path is the event side of `update_event_frame_pack` /
`update_event_frame_pack_fix` / `update_event_frame_flow_pack` (the one test_with_flow.py:121 calls): accumulate windows up to an event budget, crop to the sensor, split by the budget and
build the normalised voxel grids -- the last step runs on the GPU here (one cf_events_to_voxel call for all windows of
a pack: float atomics, so sums may differ from np.add.at in the last bit) and the grids are returned as CUDA tensors.
"""
import numpy as np
import torch

from ..utils.event_process import events_to_voxel_grid_batch
from .event_readers import FixedSizeEventReader, RefTimeEventReaderZip, SingleEventReaderNpz   # noqa: F401


def read_timestamps_file(path_to_timestamps, unit='s'):
    """Timestamps in seconds from a text file: second column of 'timestamps.txt', first column otherwise
    (video_readers.py:10-37)."""
    col = 1 if path_to_timestamps.split('/')[-1] == 'timestamps.txt' else 0
    with open(path_to_timestamps, 'r') as f:
        ts = np.array([float(line.strip().split()[col]) for line in f if line.strip()])
    if unit == 'us':
        ts = ts / 1e6
    elif unit == 'ns':
        ts = ts / 1e9
    return list(ts)


class VR:
    def __init__(self, image_dim, num_bins=5, device="cuda:0"):
        self.height, self.width = image_dim
        self.prev_ts_cache = np.zeros(1, dtype=np.float64)
        self.frame_id = 0
        self.num_frames = -1
        self.timestamps = []
        self.device = device
        self.num_bins = num_bins
        self.ending = False
        self.num_events = 0
        self.prev_frame = None

    # ---- hooks of the dataset-specific readers ----
    def update_frame(self):
        return np.zeros((self.height, self.width), dtype=np.uint8), 0

    def update_flow(self, *args):
        return np.zeros((2, self.height, self.width), dtype=np.uint8), 0

    def update_events(self):
        return None

    # ---- voxel grids of a list of windows: ONE device call ----
    def _voxels(self, windows, filter_hot_pixel):
        dev = torch.device(self.device)
        evs = [torch.as_tensor(np.ascontiguousarray(w, dtype=np.float64).reshape(-1, 4)).to(dev) for w in windows]
        grids = events_to_voxel_grid_batch(evs, self.num_bins, self.width, self.height, normalize=True,
                                           filter_hot_pixel=filter_hot_pixel)
        return [grids[i] for i in range(len(windows))]

    def _gather(self, limit_num_events, mode, frame_pack, budget_factor, keep_frames):
        """The accumulation loop shared by the two packers; returns (event_window, gt_frame)."""
        total, pack, window, gt_frame = 0, [], None, None
        while (total < budget_factor * limit_num_events) and (self.frame_id < self.num_frames):
            gt_frame, _ = self.update_frame()
            events = self.update_events()
            if keep_frames:
                frame_pack.append(gt_frame)
            if events is not None:
                pack.append(events)
                total += len(events)
            window = np.concatenate(pack, 0) if len(pack) > 1 else pack[0]
            if not keep_frames and self.frame_id >= self.num_frames:
                self.ending = True
        if keep_frames:
            frame_pack.pop(-1)
        return window, gt_frame

    def update_event_frame_pack(self, limit_num_events=-1, mode='upsampled'):
        """video_readers.py:69-143: events between frames, at most ~limit_num_events per reconstruction."""
        frame_pack = []
        if self.frame_id == 0:
            self.prev_frame, _ = self.update_frame()
        frame_pack.append(self.prev_frame)
        if limit_num_events > 0 and mode == 'upsampled':
            window, gt_frame = self._gather(limit_num_events, mode, frame_pack, 0.8, True)
            self.prev_frame = gt_frame
        else:
            gt_frame, _ = self.update_frame()
            window = self.update_events()
            self.prev_frame = gt_frame
            if window is None:
                window = np.zeros((0, 4))
        if self.frame_id >= self.num_frames:
            self.ending = True
        self.num_events = len(window)
        if limit_num_events <= 0 or mode == 'upsampled':
            return self._voxels([window], False), frame_pack, gt_frame
        return self._voxels(self._split(window, limit_num_events), True), frame_pack, gt_frame

    def update_event_frame_pack_fix(self, limit_num_events=-1, mode='upsampled'):
        """video_readers.py:146-234: a fixed number of events per reconstruction ('real': accumulate whole inter-frame
        windows up to the budget, crop to the sensor, split evenly and filter hot pixels)."""
        frame_pack = []
        if self.frame_id == 0:
            self.prev_frame, _ = self.update_frame()
        frame_pack.append(self.prev_frame)
        if limit_num_events > 0 and mode == 'upsampled':
            window, gt_frame = self._gather(limit_num_events, mode, frame_pack, 0.8, True)
        elif limit_num_events > 0 and mode == 'real':
            window, gt_frame = self._gather(limit_num_events, mode, frame_pack, 1.0, False)
        else:
            gt_frame, _ = self.update_frame()
            window = self.update_events()
            if self.frame_id >= self.num_frames:
                self.ending = True
        self.prev_frame = gt_frame
        window = window[window[:, 1] < self.width]
        window = window[window[:, 2] < self.height]
        self.num_events = len(window)
        if limit_num_events <= 0 or mode == 'upsampled':
            return self._voxels([window], False), frame_pack, gt_frame
        return self._voxels(self._split(window, limit_num_events), True), frame_pack, gt_frame

    def update_event_frame_flow_pack(self, mode='upsampled'):
        """video_readers.py:237-282 -- what test_with_flow.py:121 calls: one ground-truth frame, the flow from the previous
        frame to it (`update_flow(prev, gt)` of the dataset-specific reader) and ALL events in between as one normalised
        voxel grid (no hot-pixel filter).  Returns (event_windows, frame_pack, gt_frame, flow_list)."""
        assert mode == 'upsampled', "Data mode can not be 'real'!"
        frame_pack = []
        if self.frame_id == 0:
            self.prev_frame, _ = self.update_frame()
        frame_pack.append(self.prev_frame)
        gt_frame, _ = self.update_frame()
        flow = self.update_flow(self.prev_frame, gt_frame)
        self.prev_frame = gt_frame
        window = self.update_events()
        if window is None:
            window = np.zeros((0, 4))
        if self.frame_id >= self.num_frames:
            self.ending = True
        self.num_events = len(window)
        return self._voxels([window], False), frame_pack, gt_frame, [flow]

    @staticmethod
    def _split(window, limit_num_events):
        n = round(window.shape[0] / limit_num_events)
        return np.array_split(window, n if n > 0 else 1, axis=0)
