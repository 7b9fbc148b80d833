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
