import sys, os, argparse
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import weights_util as wu
from cista_flow_amd.e2v.e2v_model import DCEIFlowCistaNet, ERAFTCistaNet, IDCistaNet
H, W, B = 180, 240, 8
args = argparse.Namespace(image_dim=[H, W], num_bins=5, warp_mode="forward", base_channels=64, depth=5, ds=8, is_bi=False)
for name, cls in (("eiflow", DCEIFlowCistaNet), ("eraft", ERAFTCistaNet), ("idnet", IDCistaNet)):
    res = []
    for rep in range(2):
        m = cls(args).eval(); wu.fill_module(m, 1234); m = m.cuda()
        evs = [wu.synth_events(B, 5, H, W, 100 + i).cuda() for i in range(16)]
        prev = torch.zeros(B, 1, H, W, device="cuda"); st = None; fi = None
        with torch.no_grad():
            for t in range(300):
                ev = evs[t % 16]
                if name == "eiflow":
                    I, bf, st = m({"event_voxel": ev, "rec_img0": prev}, st, {})
                elif name == "eraft":
                    I, bf, st = m({"event_voxel": ev, "event_voxel_old": evs[(t - 1) % 16], "rec_img0": prev}, st, {})
                else:
                    I, bf, st = m({"event_voxel": ev, "rec_img0": prev}, st, fi, {}); fi = bf["next_flow"]
                prev = I
        torch.cuda.synchronize()
        assert torch.isfinite(I).all() and torch.isfinite(st[1]).all() and torch.isfinite(bf["flow_final"]).all()
        res.append((I.clone(), st[1].clone(), bf["flow_final"].clone()))
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    I, z, f = res[0]
    print("%s: 300 frames finite; I in [%.3f, %.3f] mean %.3f; |z| max %.2f; |flow| max %.2f; run-to-run bit-identical: %s" % (
        name, I.min().item(), I.max().item(), I.mean().item(), z.abs().max().item(), f.abs().max().item(), same))
