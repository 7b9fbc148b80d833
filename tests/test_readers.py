"""f-4: event readers + the event side of VR.update_event_frame_pack[_fix] against goldens produced by
the reference's own data_readers (tools/gen_golden.py::run_readers) on a synthetic event file regenerated here from
its seed.  The reader / windowing logic is host code (CPU tests); the voxel grids of the windows come from the HIP
kernels (GPU test) and are compared with the reference's numpy grids."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_util as gu          # noqa: E402
import weights_util as wu         # noqa: E402


def _files(tmp_path):
    g = gu.load("readers.npz")
    seed, n, W, H, bins = [int(v) for v in g["meta"]]
    p1 = str(tmp_path / "events.txt")
    wu.synth_event_file(p1, seed=seed, n=n, width=W, height=H, duration=0.5)
    p2 = str(tmp_path / "events_nohdr.txt")
    with open(p1) as f, open(p2, "w") as o:
        o.writelines(f.readlines()[1:])
    p3 = str(tmp_path / "events_in.txt")
    wu.synth_event_file(p3, seed=6, n=n, width=W, height=H, duration=0.5, overshoot=False)
    with open(p3) as f:
        lines = f.readlines()[1:]
    with open(p3, "w") as o:
        o.writelines(lines)
    return g, p1, p2, p3, W, H, bins


def _summary(ws):
    return np.array([[len(w), w[0, 0] if len(w) else -1, w[-1, 0] if len(w) else -1, w[:, 1].sum() if len(w) else 0,
                      w[:, 2].sum() if len(w) else 0, w[:, 3].sum() if len(w) else 0] for w in ws], dtype=np.float64)


def _take(it, n):
    out = []
    for _ in range(n):
        try:
            out.append(np.asarray(next(it), dtype=np.float64))
        except StopIteration:
            break
    return out


def test_event_readers_match_reference(tmp_path):
    from cista_flow_amd.data_readers import FixedSizeEventReader, RefTimeEventReaderZip
    g, p1, p2, _, W, H, bins = _files(tmp_path)
    a = _summary(_take(FixedSizeEventReader(p1, num_events=700), 8))
    assert a.shape == g["fixed_700"].shape and np.array_equal(a, g["fixed_700"])
    b = _summary(_take(FixedSizeEventReader(p1, num_events=700, k_shift=250), 12))
    assert np.array_equal(b, g["fixed_700_shift250"])
    c = _summary(_take(RefTimeEventReaderZip(p2, list(g["T_image"])), 20))
    assert c.shape == g["reftime"].shape and np.allclose(c, g["reftime"], rtol=0, atol=1e-12)


def test_npz_reader_and_timestamps(tmp_path):
    from cista_flow_amd.data_readers import SingleEventReaderNpz, read_timestamps_file
    paths = []
    for k in range(3):
        p = str(tmp_path / ("w%d.npz" % k))
        np.savez(p, t=np.arange(4.0) + k, x=np.arange(4), y=np.arange(4) * 2, p=np.array([0, 1, 0, 1]))
        paths.append(p)
    ws = list(SingleEventReaderNpz(paths))
    assert len(ws) == 3 and ws[1].shape == (4, 4) and ws[1][0, 0] == 1.0 and ws[2][3, 2] == 6
    ts = tmp_path / "timestamps.txt"
    ts.write_text("0 1000000\n1 2000000\n")
    assert read_timestamps_file(str(ts), unit="us") == [1.0, 2.0]
    other = tmp_path / "images.txt"
    other.write_text("0.5 a.png\n0.75 b.png\n")
    assert read_timestamps_file(str(other)) == [0.5, 0.75]


class _CountingVR(object):
    """VR with the voxel stage replaced by a recorder: checks the windowing alone (no GPU)."""

    @staticmethod
    def make(H, W, bins, reader, n):
        from cista_flow_amd.data_readers import VR

        class Fake(VR):
            def start(self):
                self.r = iter(reader)
                self.num_frames = n
                self.frame_id = 0
                self.windows = []

            def update_frame(self):
                self.frame_id += 1
                return np.full((self.height, self.width), self.frame_id % 250, np.uint8), 0

            def update_events(self):
                try:
                    return np.asarray(next(self.r), dtype=np.float64)
                except StopIteration:
                    return None

            def _voxels(self, windows, filter_hot_pixel):
                self.windows += [(np.asarray(w), filter_hot_pixel) for w in windows]
                return [None] * len(windows)

        vr = Fake([H, W], num_bins=bins, device="cpu")
        vr.start()
        return vr


CASES = [("fix_real", "update_event_frame_pack_fix", "real", 900), ("fix_ups", "update_event_frame_pack_fix", "upsampled", 900),
         ("pack_real", "update_event_frame_pack", "real", 300), ("pack_plain", "update_event_frame_pack", "upsampled", -1)]


@pytest.mark.parametrize("tag,method,mode,limit", CASES)
def test_windowing_matches_reference_and_oracle(tmp_path, tag, method, mode, limit):
    """Per call: number of voxel grids, frames in the pack, events kept, ground-truth frame -- and (through the CPU
    oracle's voxel-grid restatement) the grids themselves."""
    from cista_flow_amd.data_readers import RefTimeEventReaderZip
    from oracle import cista_oracle as orc
    g, _, p2, p3, W, H, bins = _files(tmp_path)
    T = list(g["T_image"])
    vr = _CountingVR.make(H, W, bins, RefTimeEventReaderZip(p2 if tag.startswith("fix") else p3, T), len(T))
    calls = []
    for _ in range(40):
        if vr.ending or vr.frame_id >= vr.num_frames:
            break
        n0 = len(vr.windows)
        ev, pack, gt = getattr(vr, method)(limit_num_events=limit, mode=mode)
        calls.append([len(vr.windows) - n0, len(pack), vr.num_events, int(gt[0, 0])])
    assert np.array_equal(np.array(calls), g[tag + "_calls"])
    grids = np.stack([orc.events_to_voxel(w, bins, W, H, True, hot) for w, hot in vr.windows])
    assert grids.shape == g[tag + "_grids"].shape
    assert np.abs(grids - g[tag + "_grids"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,method,mode,limit", CASES)
def test_windows_to_voxels_on_gpu(gpu, tmp_path, tag, method, mode, limit):
    """The same loops with the real VR: windows -> normalised voxel grids by the HIP kernels (one call per pack)."""
    from cista_flow_amd.data_readers import VR, RefTimeEventReaderZip
    g, _, p2, p3, W, H, bins = _files(tmp_path)
    T = list(g["T_image"])
    reader = iter(RefTimeEventReaderZip(p2 if tag.startswith("fix") else p3, T))

    class Fake(VR):
        def update_frame(self):
            self.frame_id += 1
            return np.full((self.height, self.width), self.frame_id % 250, np.uint8), 0

        def update_events(self):
            try:
                return np.asarray(next(reader), dtype=np.float64)
            except StopIteration:
                return None

    vr = Fake([H, W], num_bins=bins, device="cuda:0")
    vr.num_frames = len(T)
    grids = []
    for _ in range(40):
        if vr.ending or vr.frame_id >= vr.num_frames:
            break
        ev, pack, gt = getattr(vr, method)(limit_num_events=limit, mode=mode)
        assert all(e.is_cuda and e.shape == (bins, H, W) for e in ev)
        grids += [e.cpu().numpy() for e in ev]
    got = np.stack(grids)
    ref = g[tag + "_grids"]
    assert got.shape == ref.shape
    # float atomics vs np.add.at: sums differ in the last bits; hot-pixel / zero pattern must agree exactly
    assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got == 0, ref == 0)
