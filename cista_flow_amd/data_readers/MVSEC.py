"""MVSEC sequences cut by a fixed number of events (reference: data_readers/MVSEC.py:292-543, class MVSEC_NE, the
dataset test_mvsec.py:116 builds; helpers data_readers/MVSEC_utils.py:97-167,366-381).

Same constructor arguments, attributes (`INDEX_MAP`, `raw_index_shift`, `skip_num`, ...), item layout
(`raw_events_list, batch`) and `events_to_voxel` / `get_raw_events` methods as the reference.  Two differences, both
on purpose:

* storage: the reference opens `<split>_data.hdf5` / `<split>_gt.hdf5` with h5py.  h5py is not installed here, so the
  class also accepts `source=(data, gt)` -- two mappings with `.get('davis/left/events')` etc. that return anything
  indexable like an h5py dataset (numpy arrays in the tests).  With `source=None` it opens the files exactly like the
  reference and raises if h5py is missing;
* `events_to_voxel` runs on the GPU (cf_events_to_voxel_ex: scatter, centre crop, hot-pixel filter, non-zero std
  normalisation) and returns a CUDA tensor [1, bins, crop_h, crop_w]; there is no CPU path.

Ground-truth flow: `generate_corresponding_gt_flow` is restated in full.  Its single-interval branch (pure time
scaling) is pinned by the golden fixture; the multi-interval branch propagates pixels with `cv2.remap(INTER_NEAREST)`,
restated here as round-half-even + zero border -- cv2 is absent offline, so that branch is UNPINNED.
"""
import os

import numpy as np
import torch
import torch.utils.data.dataset as dataset

from ..utils.event_process import events_to_voxel_grid_batch

DatasetMapping = {}
for _n, _long in (("1", "indoor_flying/indoor_flying1"), ("2", "indoor_flying/indoor_flying2"),
                  ("3", "indoor_flying/indoor_flying3"), ("4", "indoor_flying/indoor_flying4")):
    for _k in ("in", "inday", "indoor", "indoor_flying"):
        DatasetMapping[_k + _n] = _long
for _n, _long in (("1", "outdoor_day/outdoor_day1"), ("2", "outdoor_day/outdoor_day2")):
    for _k in ("out", "outday", "outdoor", "outdoor_day"):
        DatasetMapping[_k + _n] = _long

# first / one-past-last usable image index per sequence (MVSEC.py:58-65; only the long names are keys there too)
Valid_Time_Index = {
    'indoor_flying1': [314, 2199], 'indoor_flying2': [314, 2199], 'indoor_flying3': [314, 1870],
    'indoor_flying4': [196, 570], 'outdoor_day1': [245, 3000], 'outdoor_day2': [4375, 7002],
}


def eventsToTXYP(events, process=False):
    """MVSEC rows are (x, y, t, p): reorder to (t, x, y, p) with integer-valued x / y / p (MVSEC_utils.py:366-381)."""
    t = events[:, 2]
    if process:
        t = (t - t[0]) / (t[-1] - t[0])
    col = lambda i: events[:, i].astype(np.int32)          # noqa: E731
    return np.stack([t, col(0), col(1), col(3)], axis=1)


def _remap_nearest(img, xs, ys):
    """cv2.remap(img, xs, ys, INTER_NEAREST) with the default constant-0 border (UNPINNED: cv2 absent)."""
    xi, yi = np.rint(xs).astype(np.int64), np.rint(ys).astype(np.int64)
    ok = (xi >= 0) & (xi < img.shape[1]) & (yi >= 0) & (yi < img.shape[0])
    out = np.zeros(xs.shape, dtype=img.dtype)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def _prop_flow(x_flow, y_flow, xs, ys, x_mask, y_mask, scale=1.0):
    fx, fy = _remap_nearest(x_flow, xs, ys), _remap_nearest(y_flow, xs, ys)
    x_mask[fx == 0] = False
    y_mask[fy == 0] = False
    xs += fx * scale
    ys += fy * scale


def generate_corresponding_gt_flow(flows, flows_ts, start_time, end_time):
    """Displacement between two image timestamps from the (unsynchronised) ground-truth flow maps
    (MVSEC_utils.py:97-167).  NOTE: like the reference, the single-interval branch scales flows[0] IN PLACE."""
    n = len(flows)
    assert n == len(flows_ts) - 1, "Assertion failed: %d is not equal to %d" % (n, len(flows_ts) - 1)
    x_flow, y_flow = flows[0][0], flows[0][1]
    gt_dt = flows_ts[1] - flows_ts[0]
    if start_time > flows_ts[0] and end_time <= flows_ts[1]:
        s = (end_time - start_time) / gt_dt
        x_flow *= s
        y_flow *= s
        return np.concatenate((x_flow[np.newaxis, :], y_flow[np.newaxis, :]), axis=0)
    xs, ys = np.meshgrid(np.arange(x_flow.shape[1]), np.arange(x_flow.shape[0]))
    xs, ys = xs.astype(np.float32), ys.astype(np.float32)
    x0, y0 = xs.copy(), ys.copy()
    x_mask, y_mask = np.ones(xs.shape, dtype=bool), np.ones(ys.shape, dtype=bool)
    _prop_flow(x_flow, y_flow, xs, ys, x_mask, y_mask, (flows_ts[1] - start_time) / gt_dt)
    for i in range(1, n - 1):
        _prop_flow(flows[i][0], flows[i][1], xs, ys, x_mask, y_mask)
    gt_dt = flows_ts[n] - flows_ts[n - 1]
    _prop_flow(flows[n - 1][0], flows[n - 1][1], xs, ys, x_mask, y_mask, (end_time - flows_ts[n - 1]) / gt_dt)
    dx, dy = xs - x0, ys - y0
    dx[~x_mask] = 0
    dy[~y_mask] = 0
    return np.concatenate((dx[np.newaxis, :], dy[np.newaxis, :]), axis=0)


class MVSEC_NE(dataset.Dataset):
    """Events between consecutive DAVIS frames, split into windows of ~args.num_events (MVSEC.py:292-543)."""

    def __init__(self, args, data_root, data_split='in1', data_mode='full', train_ratio=0.6, skip_num=None,
                 aug_params=None, source=None, device="cuda:0"):
        super().__init__()
        self.args = args
        self.width, self.height = 346, 260
        self.args.crop_size = [260, 346]
        self.data_root, self.data_split = data_root, data_split
        assert data_split in DatasetMapping.keys()
        self.data_filepath = os.path.join(data_root, data_split + '_data.hdf5')
        self.gt_filepath = os.path.join(data_root, data_split + '_gt.hdf5')
        self._source = source
        self.device = device
        if source is None:
            assert os.path.isfile(self.data_filepath)
            assert os.path.isfile(self.gt_filepath)
        self.data_mode, self.train_ratio = data_mode, train_ratio
        self.num_events = args.num_events
        self.event_bins = args.num_bins
        self.event_polarity = False
        args.skip_num = 1
        self.skip_num = args.skip_num if skip_num is None else skip_num
        args.skip_mode = 'i'
        self.skip_mode = 'i'
        self.raw_index_shift = Valid_Time_Index[data_split][0]
        self.raw_index_max = Valid_Time_Index[data_split][1] - 1
        self.data_length = (self.raw_index_max - self.raw_index_shift) // self.skip_num - 1
        np.random.seed(20)          # the reference seeds numpy's GLOBAL generator here; kept (callers may rely on it)
        split_index = np.random.rand(self.data_length) <= self.train_ratio
        if data_mode == 'full':
            self.INDEX_MAP = list(range(self.data_length))
        elif data_mode == 'train':
            self.INDEX_MAP = [i for i in range(self.data_length) if split_index[i]]
        elif data_mode == 'val':
            self.INDEX_MAP = [i for i in range(self.data_length) if not split_index[i]]
        else:
            raise NotImplementedError("unknow data mode {}".format(data_mode))
        self.data_length = len(self.INDEX_MAP)

    def open_hdf5(self):
        if self._source is not None:
            data_file, gt_file = self._source
        else:
            try:
                import h5py
            except ImportError as e:
                raise RuntimeError("MVSEC_NE needs h5py to open %s (or pass source=(data, gt) mappings)" % self.data_filepath) from e
            data_file, gt_file = h5py.File(self.data_filepath, 'r'), h5py.File(self.gt_filepath, 'r')
        self.events_data = data_file.get('davis/left/events')
        self.image_data = data_file.get('davis/left/image_raw')
        self.image_ts_data = data_file.get('davis/left/image_raw_ts')
        self.image_event_inds = data_file.get('davis/left/image_raw_event_inds')
        assert len(self.image_data) == len(self.image_ts_data)
        self.flow_dist_data = gt_file.get('davis/left/flow_dist')
        self.flow_dist_ts = gt_file.get('davis/left/flow_dist_ts')
        self.flow_dist_ts_numpy = np.array(self.flow_dist_ts, dtype=np.float32)
        self.image_length, self.event_length, self.flow_length = len(self.image_data), len(self.events_data), len(self.flow_dist_data)
        assert self.data_length <= self.image_length

    def events_to_voxel(self, events, height, width):
        """(t, x, y, p) rows -> [1, bins, crop_h, crop_w] CUDA tensor: voxel grid at (height, width), centre crop, hot-pixel
        filter, non-zero mean/std normalisation (MVSEC.py:389-403) -- one device call."""
        ch, cw = self.args.crop_size[:2]
        y0, x0 = (int(height) - ch) // 2, (int(width) - cw) // 2
        ev = torch.as_tensor(np.ascontiguousarray(events, dtype=np.float64).reshape(-1, 4)).to(self.device)
        if y0 == 0 and x0 == 0:
            return events_to_voxel_grid_batch([ev], self.event_bins, int(width), int(height), normalize=True, filter_hot_pixel=True)
        # the normalisation statistics are those of the CROPPED grid: build the raw grid, crop, then filter + normalise
        raw = events_to_voxel_grid_batch([ev], self.event_bins, int(width), int(height), normalize=False, filter_hot_pixel=False)
        from ..utils.event_process import event_preprocess
        return event_preprocess(raw[:, :, y0:y0 + ch, x0:x0 + cw].contiguous(), mode='std', filter_hot_pixel=True)

    def _raw_index(self, index):
        raw_index = self.INDEX_MAP[index] * self.skip_num + self.raw_index_shift
        assert raw_index < self.raw_index_max
        return raw_index

    def get_raw_events(self, index):
        if not hasattr(self, 'events_data'):
            self.open_hdf5()
        r = self._raw_index(index)
        i1, i2 = self.image_event_inds[r], self.image_event_inds[r + self.skip_num]
        assert i1 < i2
        assert i2 < self.event_length
        return self.events_data[i1:i2]

    def __getitem__(self, index):
        if not hasattr(self, 'events_data'):
            self.open_hdf5()
        r = self._raw_index(index)
        image1, image1_ts = self.image_data[r], self.image_ts_data[r]
        image2, next_ts = self.image_data[r + self.skip_num], self.image_ts_data[r + self.skip_num]
        events = eventsToTXYP(np.asarray(self.get_raw_events(index)))
        NE = self.num_events if self.num_events > 0 else events.shape[0]
        num_evs = max(1, round(events.shape[0] / NE))
        raw_events_list = [[w, w.shape[0]] for w in np.array_split(events, num_evs, axis=0)]

        left = np.searchsorted(self.flow_dist_ts_numpy, image1_ts, side='right') - 1
        right = np.searchsorted(self.flow_dist_ts_numpy, next_ts, side='right')
        assert left <= right
        assert left < self.flow_length
        assert right < self.flow_length
        flows = np.array(self.flow_dist_data[left:right])          # a copy: the single-interval branch scales in place
        final_flow = generate_corresponding_gt_flow(flows, self.flow_dist_ts_numpy[left:right + 1], image1_ts, next_ts)

        def gray(img):
            img = np.asarray(img)
            return img[..., None] if img.ndim == 2 else img[..., :3].mean(-1, keepdims=True)

        image1 = torch.from_numpy(gray(image1)).permute(2, 0, 1).float() / 255.
        image2 = torch.from_numpy(gray(image2)).permute(2, 0, 1).float() / 255.
        final_flow = torch.from_numpy(np.ascontiguousarray(final_flow)).float()
        flow_valid = ((torch.norm(final_flow, p=2, dim=0, keepdim=False) > 0) & (final_flow[0].abs() < 1000)
                      & (final_flow[1].abs() < 1000)).float().unsqueeze(0)
        ch, cw = self.args.crop_size[:2]
        if not (self.height == ch and self.width == cw):
            assert ch < self.height and cw < self.width
            y0, x0 = (self.height - ch) // 2, (self.width - cw) // 2
            image1, image2 = image1[:, y0:y0 + ch, x0:x0 + cw], image2[:, y0:y0 + ch, x0:x0 + cw]
            final_flow, flow_valid = final_flow[:, y0:y0 + ch, x0:x0 + cw], flow_valid[:, y0:y0 + ch, x0:x0 + cw]
        batch = dict(gt_img0=image1, gt_img1=image2, org_width=self.width, org_height=self.height, gt_flow=final_flow,
                     flow_valid=flow_valid)
        return raw_events_list, batch

    def __len__(self):
        return self.data_length
