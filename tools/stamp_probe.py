import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
buf = torch.zeros(8 * 4 * 20000, dtype=torch.int64, device="cuda")
os.environ["CF_STAMP_BUF"] = str(buf.data_ptr())
import conv_bench
for idx, tile in [(0, 9), (0, 4), (1, 8), (10, 1)]:
    buf.zero_()
    r = conv_bench.run(conv_bench.SHAPES[idx], tile, 3)
    torch.cuda.synchronize()
    d = buf.view(-1, 8).cpu()
    d = d[d[:, 5] > 0].double()
    n = d[:, 5].mean().item()
    m = d[:, :5].mean(0) / n
    print(conv_bench.SHAPES[idx][0], "tile", tile, "us %.1f" % r[0], "stages %d" % n,
          "per-stage cycles: load-issue %.0f | lds-read+mfma %.0f | vmcnt wait %.0f | lds-store %.0f | barrier %.0f | total %.0f" % (m[0], m[1], m[2], m[3], m[4], m.sum()))
