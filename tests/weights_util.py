"""Deterministic synthetic weights / inputs shared by the golden generator, the tests and bench.py.

No trained checkpoints exist offline (pretrained/*.tar are missing blobs), so parity is pinned with seeded
random weights.  Values depend only on (seed, canonical key) through numpy's PCG64 stream, so the reference
model in tools/gen_golden.py and the models under test get bit-identical parameters without shipping them.
BatchNorm running statistics are randomised and Lambda is drawn from U(0.05, 0.5): the reference's default
init (0.001*rand, e2v/base_layers.py:31) would make the soft threshold a no-op.
"""
import re
import zlib

import numpy as np
import torch


def canonical_key(k):
    # lista_blocks.0..4 alias one IstaBlock (e2v/e2v_model.py:34-35); downsample.1 aliases norm3
    k = re.sub(r"lista_blocks\.\d+\.", "lista_blocks.0.", k)
    k = k.replace(".downsample.1.", ".norm3.")
    return k


def make_tensor(key, shape, seed):
    ck = canonical_key(key)
    rng = np.random.default_rng([int(seed), zlib.crc32(ck.encode())])
    shape = tuple(shape)
    if ck.endswith("num_batches_tracked"):
        return torch.tensor(0, dtype=torch.int64)
    if ck.endswith("running_var"):
        a = rng.uniform(0.5, 1.5, shape)
    elif ck.endswith("running_mean"):
        a = rng.normal(0.0, 0.2, shape)
    elif "Lambda" in ck:
        a = rng.uniform(0.05, 0.5, shape)
    elif len(shape) == 4:
        # gains tuned (tools/gen_golden.py header) so that the synthetic network is well conditioned:
        # I_rec spans ~[0.05, 0.95], ~50 % of the sparse code survives the soft threshold, flows are a few
        # pixels, and a 1e-6 input perturbation stays ~1e-5 after 4 recurrent frames (no chaotic blow-up).
        fan_in = shape[1] * shape[2] * shape[3]
        if ".D.conv2d" in ck or ".P.conv2d" in ck:
            gain = 0.5
        elif "final_conv" in ck:
            gain = 6.0
        elif "flow_head.conv2" in ck:
            gain = 0.08
        elif "flow_head2.conv2" in ck:
            gain = 0.08
        elif "update_block.mask.2" in ck or "update_net.mask.2" in ck or "update_net.mask2.2" in ck:
            gain = 3.0       # spread the convex-upsampling logits so the softmax is not uniform
        elif ck.startswith("cista_net") or not ("fnet" in ck or "cnet" in ck or "enet" in ck or "update_block" in ck or "update_net" in ck or "fusion" in ck):
            gain = 1.5
        else:
            gain = 1.0
        a = rng.normal(0.0, gain / np.sqrt(fan_in), shape)
    elif ck.endswith("flow_head.conv2.bias") or ck.endswith("flow_head2.conv2.bias"):
        a = rng.normal(0.0, 0.008, shape)
    elif ck.endswith(".weight"):      # BatchNorm scale
        a = rng.uniform(0.5, 1.5, shape)
    else:                             # biases
        a = rng.normal(0.0, 0.1, shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_state_dict(keys_shapes, seed):
    return {k: make_tensor(k, s, seed) for k, s in keys_shapes}


def fill_module(module, seed):
    """Overwrite every parameter / buffer of `module` in place (works for reference and shell modules)."""
    sd = module.state_dict()
    new = make_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed)
    module.load_state_dict(new, strict=True)
    return module


def synth_events(B, bins, H, W, seed, density=0.13):
    """Event voxel grid stand-in: ~13 % non-zero voxels, non-zero entries normalised to
    mean 0 / std 1 like utils/event_process.py:193-216 ('std' mode); zeros stay zero."""
    rng = np.random.default_rng([int(seed), 977])
    v = rng.normal(0.0, 1.0, (B, bins, H, W)) * (rng.uniform(0, 1, (B, bins, H, W)) < density)
    out = np.zeros_like(v)
    for b in range(B):
        nz = v[b] != 0
        if nz.any():
            m, s = v[b][nz].mean(), v[b][nz].std()
            out[b][nz] = (v[b][nz] - m) / (s if s > 0 else 1.0)
    return torch.from_numpy(out.astype(np.float32))


def synth_event_file(path, seed=5, n=6000, width=36, height=28, duration=0.5, overshoot=True):
    """A small whitespace-separated event file ('t x y pol', one header line) for the reader tests: sorted
    timestamps, a few events beyond the sensor (x >= width / y >= height: the readers crop them) and one "hot" pixel
    that fires every few events with the same polarity (|voxel| > 25 / bins: the hot-pixel filter zeroes it)."""
    rng = np.random.default_rng([int(seed), 4242])
    t = np.sort(rng.uniform(0.0, duration, n))
    x = rng.integers(0, width + (3 if overshoot else 0), n)
    y = rng.integers(0, height + (2 if overshoot else 0), n)
    p = rng.integers(0, 2, n)
    hot = np.arange(0, n, 9)
    x[hot], y[hot], p[hot] = 7, 11, 1
    with open(path, "w") as f:
        f.write("%d %d\n" % (width, height))
        for i in range(n):
            f.write("%.9f %d %d %d\n" % (t[i], x[i], y[i], p[i]))
    return np.stack([t, x, y, p], 1).astype(np.float64)


# ---------------------------------------------------------------------------------------------------------
# A synthetic stand-in for one MVSEC sequence (data_readers/MVSEC.py reads `<split>_data.hdf5` / `<split>_gt.hdf5`):
# mappings with .get('davis/left/...') whose values index like h5py datasets (slices come back as COPIES).
# Frames and flow maps are generated on access from (seed, index), so nothing large is ever held.
# ---------------------------------------------------------------------------------------------------------
class _LazyDataset(object):
    def __init__(self, n, make, dtype):
        self.n, self.make, self.dtype = n, make, dtype

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            idx = range(*i.indices(self.n))
            return np.stack([self.make(j) for j in idx]) if len(idx) else np.zeros((0,), self.dtype)
        return self.make(int(i))


class _ArrayDataset(object):
    def __init__(self, a):
        self.a = a

    def __len__(self):
        return len(self.a)

    def __getitem__(self, i):
        return np.array(self.a[i])            # h5py hands out copies

    def __array__(self, dtype=None, copy=None):
        return np.array(self.a, dtype=dtype)


class _Mapping(object):
    def __init__(self, d):
        self.d = d

    def get(self, k):
        return self.d[k]


MVSEC_SPLIT, MVSEC_FIRST = 'indoor_flying4', 196      # Valid_Time_Index['indoor_flying4'] = [196, 570]: the shortest sequence


def synth_mvsec_source(seed=3, n_items=10, width=346, height=260):
    """-> (data, gt): `n_items` image intervals of ~4-7 k events each after image MVSEC_FIRST.  Image period 20 ms, flow maps
    every 50 ms; some image intervals lie inside one flow interval (single-interval ground truth), others straddle two."""
    rng = np.random.default_rng([int(seed), 5151])
    n_img = 572
    img_ts = 10.0 + 0.02 * np.arange(n_img, dtype=np.float64)
    counts = rng.integers(4000, 7000, n_items)
    rows = []
    for k in range(n_items):
        t0, t1 = img_ts[MVSEC_FIRST + k], img_ts[MVSEC_FIRST + k + 1]
        n = int(counts[k])
        t = np.sort(rng.uniform(t0, t1, n))
        x, y = rng.integers(0, width, n), rng.integers(0, height, n)
        p = rng.integers(0, 2, n) * 2 - 1
        hot = np.arange(0, n, 11)
        x[hot], y[hot], p[hot] = 123, 77, 1           # one hot pixel: |voxel| > 25 / bins
        rows.append(np.stack([x, y, t, p], 1).astype(np.float64))
    tail = np.stack([np.zeros(8), np.zeros(8), img_ts[MVSEC_FIRST + n_items] + 1e-4 * np.arange(8), np.ones(8)], 1)
    events = np.concatenate(rows + [tail], 0)          # MVSEC rows are (x, y, t, p)
    inds = np.zeros(n_img, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    inds[MVSEC_FIRST:MVSEC_FIRST + n_items + 1] = cum
    inds[MVSEC_FIRST + n_items + 1:] = cum[-1]
    yy, xx = np.mgrid[0:height, 0:width]

    def frame(i):
        return ((yy * 3 + xx * 5 + i * 7) % 256).astype(np.uint8)

    n_flow = 12
    flow_ts = (img_ts[MVSEC_FIRST] - 0.013 + 0.05 * np.arange(n_flow + 1)).astype(np.float64)

    def flow(j):
        f = np.stack([2.0 * np.sin(0.05 * xx + 0.3 * j) + 0.01 * yy, 1.5 * np.cos(0.04 * yy - 0.2 * j)]).astype(np.float64)
        f[:, 40:60, 100:140] = 0.0                     # no ground truth here
        f[0, 200:210, 300:320] = 5000.0                # beyond the |flow| < 1000 validity bound
        return f

    data = _Mapping({'davis/left/events': _ArrayDataset(events), 'davis/left/image_raw': _LazyDataset(n_img, frame, np.uint8),
                     'davis/left/image_raw_ts': _ArrayDataset(img_ts), 'davis/left/image_raw_event_inds': _ArrayDataset(inds)})
    gt = _Mapping({'davis/left/flow_dist': _LazyDataset(n_flow, flow, np.float64), 'davis/left/flow_dist_ts': _ArrayDataset(flow_ts)})
    return data, gt
