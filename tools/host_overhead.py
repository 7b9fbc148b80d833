"""Host enqueue time per step, eager vs hipGraph replay (tuning tool, GPU box only): python tools/host_overhead.py"""
import sys, os, time, argparse
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import weights_util as wu
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet
for B in (1, 2, 8):
    for graph in (0, 1):
        H, W = 180, 240
        args = argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode="forward", base_channels=64, depth=5, ds=8, is_bi=False)
        m = DCEIFlowCistaNet(args).eval(); wu.fill_module(m, 1234); m = m.cuda()
        ev = wu.synth_events(B, 5, H, W, 1).cuda()
        prev = torch.zeros(B, 1, H, W, device="cuda"); st = None
        hd = m._be().get(B, ev.device)
        hd.graph_enable(bool(graph))
        with torch.no_grad():
            for _ in range(8):
                I, bf, st = m({"event_voxel": ev, "rec_img0": prev}, st, {}); prev = I
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 60
            for _ in range(n):
                I, bf, st = m({"event_voxel": ev, "rec_img0": prev}, st, {}); prev = I
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("B=%d graph=%d: host enqueue %.3f ms/step, total %.3f ms/step (%.1f frames/s), graph stats %s" % (
            B, graph, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, B * n / (t2 - t0), hd.graph_stats()), flush=True)
        del m, hd
