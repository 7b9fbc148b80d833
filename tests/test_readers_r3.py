"""f-4 remainder: `VR.update_event_frame_flow_pack` -- the packer test_with_flow.py:121
calls -- and the MVSEC dataset `MVSEC_NE` (test_mvsec.py:116), against tests/golden/readers_r3.npz, which
tools/gen_golden.py::run_readers_r3 produced by running the reference's own classes (h5py replaced by an in-memory
source; the synthetic inputs are regenerated here from their seeds).  Host logic runs on the CPU; the voxel grids come
from the HIP kernels (GPU tests)."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu          # noqa: E402
import weights_util as wu         # noqa: E402


def _event_file(tmp_path, g):
    seed, n, W, H, bins = [int(v) for v in g["meta"]]
    p3 = str(tmp_path / "events_in.txt")
    wu.synth_event_file(p3, seed=seed, n=n, width=W, height=H, duration=0.5, overshoot=False)
    with open(p3) as f:
        lines = f.readlines()[1:]
    with open(p3, "w") as o:
        o.writelines(lines)
    return p3, W, H, bins


def _fake_vr(base, H, W, bins, reader, n, device, record=None):
    class Fake(base):
        def update_frame(self):
            self.frame_id += 1
            return np.full((self.height, self.width), self.frame_id % 250, np.uint8), 0

        def update_flow(self, prev, cur):
            return np.full((2, self.height, self.width), float(cur[0, 0]) - 0.5 * float(prev[0, 0]), np.float32)

        def update_events(self):
            try:
                return np.asarray(next(reader), dtype=np.float64)
            except StopIteration:
                return None

    if record is not None:
        def _voxels(self, windows, filter_hot_pixel):
            record.extend((np.asarray(w), filter_hot_pixel) for w in windows)
            return [None] * len(windows)
        Fake._voxels = _voxels
    vr = Fake([H, W], num_bins=bins, device=device)
    vr.num_frames = n
    return vr


def _run_flow_pack(vr):
    calls, grids = [], []
    for _ in range(40):
        if vr.ending or vr.frame_id >= vr.num_frames:
            break
        ev, pack, gt, flows = vr.update_event_frame_flow_pack()
        grids += list(ev)
        calls.append([len(ev), len(pack), vr.num_events, int(gt[0, 0]), int(pack[0][0, 0]), len(flows), float(flows[0][0, 0, 0])])
    return np.array(calls, dtype=np.float64), grids


def test_flow_pack_windowing_matches_reference(tmp_path):
    from cista_flow_amd.data_readers import VR, RefTimeEventReaderZip
    from oracle import cista_oracle as orc
    g = gu.load("readers_r3.npz")
    p3, W, H, bins = _event_file(tmp_path, g)
    T = list(g["T_image"])
    windows = []
    vr = _fake_vr(VR, H, W, bins, iter(RefTimeEventReaderZip(p3, T)), len(T), "cpu", record=windows)
    calls, _ = _run_flow_pack(vr)
    assert np.array_equal(calls, g["flowpack_calls"])
    assert all(hot is False for _, hot in windows)           # video_readers.py:277: filter_hot_pixel=False
    grids = np.stack([orc.events_to_voxel(w, bins, W, H, True, False) for w, _ in windows])
    assert grids.shape == g["flowpack_grids"].shape and np.abs(grids - g["flowpack_grids"]).max() < 2e-5
    with pytest.raises(AssertionError):
        vr.update_event_frame_flow_pack(mode='real')


@pytest.mark.gpu
def test_flow_pack_voxels_on_gpu(gpu, tmp_path):
    from cista_flow_amd.data_readers import VR, RefTimeEventReaderZip
    g = gu.load("readers_r3.npz")
    p3, W, H, bins = _event_file(tmp_path, g)
    T = list(g["T_image"])
    vr = _fake_vr(VR, H, W, bins, iter(RefTimeEventReaderZip(p3, T)), len(T), "cuda:0")
    calls, grids = _run_flow_pack(vr)
    assert np.array_equal(calls, g["flowpack_calls"])
    got, ref = np.stack([e.cpu().numpy() for e in grids]), g["flowpack_grids"]
    assert got.shape == ref.shape and np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got == 0, ref == 0)


def _mvsec(device="cpu", **kw):
    from cista_flow_amd.data_readers.MVSEC import MVSEC_NE
    a = argparse.Namespace(num_events=2000, num_bins=5)
    return MVSEC_NE(a, data_root="/nonexistent", data_split=wu.MVSEC_SPLIT, source=wu.synth_mvsec_source(seed=3), device=device, **kw)


def test_mvsec_items_match_reference():
    """Index arithmetic, the train/val split (numpy's global seed 20), event windows (t, x, y, p rows split by ~num_events),
    images, single-interval ground-truth flow and its validity mask."""
    g = gu.load("readers_r3.npz")
    ds = _mvsec()
    assert [len(ds), ds.raw_index_shift, ds.raw_index_max, ds.skip_num] == [int(v) for v in g["mv_len"]]
    val = _mvsec(data_mode='val')
    assert [len(val)] + val.INDEX_MAP[:24] == [int(v) for v in g["mv_val_index"]]
    assert ds.args.crop_size == [260, 346] and ds.args.skip_mode == 'i'
    for it in [int(v) for v in g["mv_items"]]:
        raw_list, batch = ds[it]
        win = np.array([[n, w[0, 0], w[-1, 0], w[:, 1].sum(), w[:, 2].sum(), w[:, 3].sum()] for w, n in raw_list], dtype=np.float64)
        assert np.array_equal(win, g["mv%d_windows" % it]), it
        assert np.array_equal(batch["gt_img0"][:, ::7, ::9].numpy(), g["mv%d_img0" % it])
        assert np.array_equal(batch["gt_img1"][:, ::7, ::9].numpy(), g["mv%d_img1" % it])
        assert np.array_equal(batch["gt_flow"][:, ::5, ::6].numpy(), g["mv%d_flow" % it]), it
        v = g["mv%d_valid" % it]
        assert [float(batch["flow_valid"].sum()), batch["flow_valid"].shape[1], batch["flow_valid"].shape[2], batch["org_width"],
                batch["org_height"]] == [float(x) for x in v]
        assert len(ds.get_raw_events(it)) == sum(n for _, n in raw_list)
    with pytest.raises(AssertionError):
        _mvsec().__class__(argparse.Namespace(num_events=1, num_bins=5), "/nonexistent", data_split='indoor_flying4')   # no files, no source


def test_mvsec_multi_interval_flow_runs():
    """Items whose image interval straddles two ground-truth flow maps take the propagation branch (cv2.remap in the
    reference: UNPINNED here) -- it must run and zero the flow where the ground truth is absent."""
    ds = _mvsec()
    _, batch = ds[1]
    f = batch["gt_flow"]
    assert f.shape == (2, 260, 346) and torch.isfinite(f).all()
    assert float(f[:, 45:55, 105:135].abs().max()) == 0.0 and float(f.abs().max()) > 0.1


@pytest.mark.gpu
def test_mvsec_events_to_voxel_on_gpu(gpu):
    g = gu.load("readers_r3.npz")
    ds = _mvsec(device="cuda:0")
    for it in [int(v) for v in g["mv_items"]]:
        raw_list, _ = ds[it]
        evs = ds.events_to_voxel(raw_list[0][0], 260, 346)
        assert evs.is_cuda and list(evs.shape) == [int(v) for v in g["mv%d_voxstat" % it][3:]]
        ref = g["mv%d_vox" % it]
        got = evs[0, :, 60:100, 100:160].cpu().numpy()
        assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max()) and np.array_equal(got == 0, ref == 0)
        s = g["mv%d_voxstat" % it]
        e = evs.double()
        assert abs(float(e.abs().sum()) - s[1]) < 1e-4 * s[1] and float((evs != 0).sum()) == s[2]
        assert float(evs[0, :, 77, 123].abs().max()) == 0.0          # the hot pixel is filtered (|v| > 25 / bins)


@pytest.mark.gpu
def test_event_preprocess_on_device_and_cropped_grids(gpu):
    """cf_voxel_preprocess = event_preprocess('std') of device-resident grids (the crop-then-normalise order of
    MVSEC.py:389-403), and the hot-pixel filter WITHOUT normalisation (it used to be skipped silently)."""
    from cista_flow_amd.utils.event_process import event_preprocess, events_to_voxel_grid_batch
    from oracle import cista_oracle as orc
    data, _ = wu.synth_mvsec_source(seed=4, n_items=2)
    ev = np.asarray(data.get('davis/left/events')[0:5000])
    txyp = np.stack([ev[:, 2], ev[:, 0], ev[:, 1], ev[:, 3]], 1)
    raw_ref = orc.events_to_voxel(txyp, 5, 346, 260, False, False)
    dev_ev = [torch.as_tensor(txyp).to(gpu)]
    raw = events_to_voxel_grid_batch(dev_ev, 5, 346, 260, normalize=False)
    # float atomics vs np.add.at: the hot pixel sums several hundred contributions, so compare relative to the grid's scale
    scale = max(1.0, float(np.abs(raw_ref).max()))
    assert np.abs(raw[0].cpu().numpy() - raw_ref).max() < 5e-6 * scale
    # hot filter alone
    hot = events_to_voxel_grid_batch(dev_ev, 5, 346, 260, normalize=False, filter_hot_pixel=True)[0].cpu().numpy()
    ref_hot = raw_ref.copy()
    ref_hot[np.abs(ref_hot) > 25.0 / 5] = 0
    assert np.abs(raw_ref).max() > 5.0 and np.abs(hot).max() <= 5.0 and np.abs(hot - ref_hot).max() < 2e-5
    # crop, then filter + normalise (two grids in one call; each grid on its own statistics)
    crop = raw[:, :, 2:258, 45:301].contiguous()
    both = torch.cat([crop, 0.5 * crop], 0)
    out = event_preprocess(both, filter_hot_pixel=True).cpu().numpy()
    for k, scale in enumerate((1.0, 0.5)):
        c = (scale * raw_ref[:, 2:258, 45:301]).astype(np.float32)
        c[np.abs(c) > 5.0] = 0
        nz = c != 0
        mean = c.sum() / nz.sum()
        sd = np.sqrt((c.astype(np.float64) ** 2).sum() / nz.sum() - mean ** 2)
        want = nz * (c - mean) / (sd + 1e-8)
        assert np.abs(out[k] - want).max() < 5e-5 * max(1.0, np.abs(want).max())
    single = event_preprocess(crop[0], filter_hot_pixel=True)
    assert single.shape == crop[0].shape and torch.equal(single.cpu(), torch.from_numpy(out[0]))
    # MVSEC_NE with a crop smaller than the sensor takes this path
    ds = _mvsec(device="cuda:0")
    ds.args.crop_size = [256, 256]
    got = ds.events_to_voxel(txyp, 260, 346)
    assert got.shape == (1, 5, 256, 256) and np.abs(got[0].cpu().numpy() - out[0]).max() < 1e-6
