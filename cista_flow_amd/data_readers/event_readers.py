"""Event-file readers with the reference's class names and iteration protocol (reference:
data_readers/event_readers.py:6-128).  Each iterator yields one event window as a float64 array [N, 4] with rows
(timestamp, x, y, polarity) -- what `VR.update_events` hands to the voxel-grid stage.

Host-side I/O only; the windows go to the GPU through cista_flow_amd.data_readers.video_readers.VR (raw events are
uploaded, 32 bytes each, and turned into normalised voxel grids by csrc/pointwise.hip::events_scatter_kernel).
"""
from os.path import splitext

import numpy as np
import pandas as pd

_COLS = ['t', 'x', 'y', 'pol']
_DTYPES = {'t': np.float64, 'x': np.int16, 'y': np.int16, 'pol': np.int16}


class FixedSizeEventReader:
    """Non-overlapping (k_shift <= 0) or sliding (a new window every `k_shift` events) windows of `num_events`
    events out of a whitespace-separated '.txt' / '.zip' file whose first line is a header (event_readers.py:6-48)."""

    def __init__(self, path_to_event_file, num_events=10000, k_shift=-1, start_index=0):
        self.iterator = pd.read_csv(path_to_event_file, sep=r'\s+', header=None, iterator=True, names=_COLS, dtype=_DTYPES,
                                    engine='c', index_col=False, skiprows=start_index + 1, nrows=None, memory_map=True)
        self.num_events = num_events
        self.k_shift = k_shift
        self.prev_events_size = num_events - k_shift
        self.frame_idx = 0
        self.prev_events = None

    def __iter__(self):
        return self

    def __next__(self):
        if self.k_shift > 0:
            if self.frame_idx == 0:
                window = np.array(self.iterator.get_chunk(self.num_events))
            else:
                fresh = np.array(self.iterator.get_chunk(self.k_shift))
                window = np.concatenate((self.prev_events, fresh), 0)
            self.prev_events = window[-self.prev_events_size:].copy()
            self.frame_idx += 1
            return window
        return np.array(self.iterator.get_chunk(self.num_events))


class RefTimeEventReaderZip:
    """Windows bounded by the timestamps of the reference intensity frames: window i holds the events with
    T_image[i] <= t < T_image[i+1]; timestamps are returned relative to T_image[0] (event_readers.py:51-101)."""

    def __init__(self, path_to_event_file, T_image):
        assert splitext(path_to_event_file)[1] in ['.txt', '.csv', '.zip']
        frame = pd.read_csv(path_to_event_file, iterator=False, delimiter=' ', names=['t', 'x', 'y', 'p'],
                            dtype={'t': np.float64, 'x': np.int16, 'y': np.int16, 'p': np.int16}, engine='c', index_col=False)
        self.values = frame.values                      # float64 [N, 4]
        self.t0 = T_image[0]
        self.T_image = np.array(T_image) - T_image[0]
        self.len = len(T_image) - 1
        rel = self.values[:, 0] - self.t0
        n = len(rel)
        # first event at or after each frame time; past the last event: the last index (the reference's fallback)
        idx = np.searchsorted(rel, self.T_image, side='left')
        self.bound_index = [int(i) if i < n else n - 1 for i in idx]
        self.frame_id = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.frame_id >= self.len:
            raise StopIteration
        a, b = self.bound_index[self.frame_id], self.bound_index[self.frame_id + 1]
        window = self.values[a:b].copy()
        window[:, 0] -= self.t0
        self.frame_id += 1
        return window


class SingleEventReaderNpz:
    """One '.npz' file (arrays t, x, y, p) per window, for simulated sequences (event_readers.py:104-128)."""

    def __init__(self, path_to_events):
        self.path_to_events = path_to_events
        self.len = len(path_to_events)
        self.frame_id = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.frame_id >= self.len:
            raise StopIteration
        z = np.load(self.path_to_events[self.frame_id])
        self.frame_id += 1
        return np.stack((z["t"], z["x"], z["y"], z["p"]), axis=1)
