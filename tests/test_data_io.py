"""f-2 output stage: FlowWriter's colour coding (utils/data_io.py:9-29) and the PNG writers.  UNPINNED by the reference (cv2 is not
installed where goldens are generated): the oracle restates OpenCV's published arithmetic and is itself checked against the textbook
HSV -> RGB of the standard library; the GPU kernel is held to the oracle."""
import colorsys
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cista_oracle as orc   # noqa: E402


def _flow(seed, H, W, scale=3.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((2, H, W)) * scale).astype(np.float32)


def test_oracle_colour_coding_matches_textbook_hsv():
    f = _flow(0, 24, 40)
    out = orc.merge_optical_flow(f)
    assert out.shape == (24, 40, 3) and out.dtype == np.uint8
    ang = np.arctan2(f[1], f[0])
    ang[ang < 0] += 2 * np.pi
    Hh = (ang * 180 / np.pi / 2).astype(np.uint8)
    mag = np.hypot(f[0], f[1])
    V = (255 * mag / mag.max()).astype(np.uint8)
    for y in range(0, 24, 3):
        for x in range(0, 40, 3):
            r, g, b = colorsys.hsv_to_rgb(Hh[y, x] / 180.0, 1.0, V[y, x] / 255.0)
            ref = np.array([round(b * 255), round(g * 255), round(r * 255)])
            assert np.abs(ref - out[y, x].astype(int)).max() <= 1, (y, x)
    # the brightest pixel carries V = 255, an all-zero flow is black, pure +x flow is hue 0 = red (B, G, R) = (0, 0, 255)
    assert out.max() == 255
    assert orc.merge_optical_flow(np.zeros((2, 4, 4), np.float32)).max() == 0
    px = np.zeros((2, 2, 2), np.float32)
    px[0] = 1.0
    assert (orc.merge_optical_flow(px) == np.array([0, 0, 255], np.uint8)).all()


@pytest.mark.gpu
def test_flow_colour_coding_gpu_matches_oracle(gpu):
    import torch
    from cista_flow_amd.utils import data_io
    for seed, (H, W) in enumerate([(180, 240), (37, 53), (8, 8)]):
        fl = np.stack([_flow(10 + seed, H, W), _flow(20 + seed, H, W, 0.2)])          # two images with different maxima
        got = data_io.merge_optical_flow(torch.from_numpy(fl).to(gpu))
        assert got.shape == (2, H, W, 3) and got.dtype == np.uint8
        for b in range(2):
            ref = orc.merge_optical_flow(fl[b])
            d = np.abs(got[b].astype(int) - ref.astype(int)).max(axis=-1)
            # device atan2f / sqrtf against numpy's: a pixel on an integer hue (or value) boundary may land in the neighbouring bucket;
            # everything else is byte-exact
            assert (d > 0).mean() < 2e-3, (seed, b, (d > 0).mean())
            assert d.max() <= 12, d.max()
    single = data_io.merge_optical_flow(torch.from_numpy(_flow(3, 16, 20)).to(gpu))
    assert single.shape == (16, 20, 3)
    assert data_io.merge_optical_flow(torch.zeros(2, 9, 9, device=gpu)).max() == 0


@pytest.mark.gpu
def test_writers_write_the_reference_file_names(gpu, tmp_path):
    import argparse
    import torch
    from PIL import Image
    from cista_flow_amd.utils import data_io
    cfgs = argparse.Namespace(output_folder=str(tmp_path), is_write_image=True, is_write_flow=True)
    iw = data_io.ImageWriter(cfgs, "cista-eiflow", "seq0")
    fw = data_io.FlowWriter(cfgs, "cista-eiflow", "seq0")
    img = (np.arange(12 * 16).reshape(12, 16) % 256).astype(np.float32)
    iw(img, 7)
    fl = _flow(5, 12, 16)
    fw(torch.from_numpy(fl).to(gpu), 7)
    p_img = os.path.join(str(tmp_path), "cista-eiflow", "seq0", "frame_0000000007.png")
    p_flow = os.path.join(str(tmp_path), "cista-eiflow", "seq0", "flow", "flow_0000000007.png")
    assert (np.asarray(Image.open(p_img)) == np.uint8(img)).all()
    rgb = np.asarray(Image.open(p_flow))
    assert (rgb[..., ::-1] == data_io.merge_optical_flow(torch.from_numpy(fl).to(gpu))).all()      # file channels are R, G, B of the BGR array
    off = data_io.ImageWriter(argparse.Namespace(output_folder=str(tmp_path / "x"), is_write_image=False, is_write_flow=False), "m")
    off(img, 1)
    assert not os.path.exists(str(tmp_path / "x"))
