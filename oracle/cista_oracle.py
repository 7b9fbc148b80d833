"""CPU oracle of the CISTA-Flow inference hot path (cista-eiflow) -- TEST INFRASTRUCTURE ONLY.

A functional restatement, written from the reference's semantics, of
    CistaLSTCNet.forward            /root/reference/e2v/e2v_model.py:49-98
    ConvLSTC / ConvLSTM / ConvLayer /root/reference/e2v/base_layers.py:38-71, 75-132, 137-212
    forwardWarp / backWarp          /root/reference/utils/flow_utils.py:83-120, 153-190
    ImagePadder                     /root/reference/utils/image_process.py:60-107
    DCEIFlow.forward                /root/reference/DCEIFlow/DCEIFlow.py:32-44, 143-227, 295-299
    BasicEncoder / ResidualBlock    /root/reference/DCEIFlow/core/backbone/raft_encoder.py:6-59, 125-203
    CorrBlock                       /root/reference/DCEIFlow/core/corr/raft_corr.py:15-65
    bilinear_sampler, upflow8       /root/reference/DCEIFlow/utils/sample_utils.py:38-68
    BasicUpdateBlockNoMask          /root/reference/DCEIFlow/core/decoder/with_event_updater.py:6-14, 35-67, 90-112, 156-171
    DCEIFlowCistaNet.forward (a5)   /root/reference/e2v/e2v_model.py:144-196

Weights come in as a plain dict keyed by the reference's state_dict names.  The only "heavy"
primitive used is F.conv2d; every resampling op (grid_sample, interpolate, avg_pool, instance /
batch norm) is written out as explicit index arithmetic so the semantics the HIP kernels must
reproduce are stated here and nowhere else.

Pinning: tests/test_oracle_golden.py checks this file against tests/golden/*.npz, which
tools/gen_golden.py produced by importing and running the reference itself in the build
container (the reference has no tests / golden vectors of its own -- SURVEY.md section 4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------------------------


def conv2d(x, w, b, stride=1, pad=(0, 0), mode="zeros"):
    """nn.Conv2d with padding_mode 'zeros' | 'reflect' (pad = (padH, padW))."""
    ph, pw = pad
    if mode == "reflect":
        if ph or pw:
            x = reflect_pad(x, ph, pw)
        return F.conv2d(x, w, b, stride=stride)
    return F.conv2d(x, w, b, stride=stride, padding=(ph, pw))


def reflect_pad(x, ph, pw):
    """mirror without repeating the edge sample: index -1 -> 1, H -> H-2."""
    H, W = x.shape[-2:]
    iy = torch.arange(-ph, H + ph).abs()
    iy = torch.where(iy >= H, 2 * (H - 1) - iy, iy)
    ix = torch.arange(-pw, W + pw).abs()
    ix = torch.where(ix >= W, 2 * (W - 1) - ix, ix)
    return x[..., iy, :][..., ix]


def softshrink(x, lambd):
    """base_layers.py:11-12"""
    return torch.relu(x - lambd) - torch.relu(-x - lambd)


def _lin_src(out_size, in_size, align_corners):
    """source indices / weights of 1-D linear interpolation as ATen computes them."""
    d = torch.arange(out_size, dtype=torch.float32)
    if align_corners:
        scale = (in_size - 1) / (out_size - 1) if out_size > 1 else 0.0
        src = torch.tensor(scale, dtype=torch.float32) * d
    else:
        scale = in_size / out_size
        src = torch.tensor(scale, dtype=torch.float32) * (d + 0.5) - 0.5
        src = src.clamp(min=0)
    i0 = src.floor().long().clamp(max=in_size - 1)
    i1 = torch.where(i0 < in_size - 1, i0 + 1, i0)
    l1 = src - i0.float()
    l0 = 1.0 - l1
    return i0, i1, l0, l1


def interp_bilinear(x, out_h, out_w, align_corners):
    """F.interpolate(x, size=(out_h,out_w), mode='bilinear', align_corners=...)."""
    H, W = x.shape[-2:]
    y0, y1, ly0, ly1 = _lin_src(out_h, H, align_corners)
    x0, x1, lx0, lx1 = _lin_src(out_w, W, align_corners)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    t = lx0 * top[..., x0] + lx1 * top[..., x1]
    b = lx0 * bot[..., x0] + lx1 * bot[..., x1]
    return ly0[:, None] * t + ly1[:, None] * b


def _reflect_coord(x, size):
    """ATen grid_sampler reflection about [0, size-1] (align_corners=True), then clip."""
    if size <= 1:
        return torch.zeros_like(x)
    twice_span = float(2 * (size - 1))
    a = x.abs()
    flips = torch.trunc(a / twice_span)
    extra = a - flips * twice_span
    r = torch.minimum(extra, twice_span - extra)
    return r.clamp(0, size - 1)


def warp(img, flow, mode="forward"):
    """a4: grid_sample(img, g, bilinear, align_corners=True, padding_mode='reflection') with
    g = 2*((x -/+ u)/W - 0.5): the sampled pixel is (x -/+ u)*(W-1)/W -- zero flow is NOT identity."""
    B, C, H, W = img.shape
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    u, v = flow[:, 0], flow[:, 1]
    if mode == "forward":
        xs, ys = gx[None].float() - u, gy[None].float() - v
    else:
        xs, ys = gx[None].float() + u, gy[None].float() + v
    xs = 2 * (xs / W - 0.5)
    ys = 2 * (ys / H - 0.5)
    ix = (xs + 1) * ((W - 1) / 2)
    iy = (ys + 1) * ((H - 1) / 2)
    ix = _reflect_coord(ix, W)
    iy = _reflect_coord(iy, H)
    x0, y0 = ix.floor(), iy.floor()
    tx, ty = ix - x0, iy - y0
    x0, y0 = x0.long(), y0.long()
    x1, y1 = x0 + 1, y0 + 1
    flat = img.reshape(B, C, H * W)

    def tap(yy, xx):
        ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)).float()
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, H * W).expand(B, C, H * W)
        return flat.gather(2, idx).reshape(B, C, H, W) * ok[:, None]

    w00 = ((1 - ty) * (1 - tx))[:, None]
    w01 = ((1 - ty) * tx)[:, None]
    w10 = (ty * (1 - tx))[:, None]
    w11 = (ty * tx)[:, None]
    return tap(y0, x0) * w00 + tap(y0, x1) * w01 + tap(y1, x0) * w10 + tap(y1, x1) * w11


# ----------------------------------------------------------------------------------------------
# CISTA-LSTC  (a3)
# ----------------------------------------------------------------------------------------------


def _cl(sd, pre, x, stride=1, pad=1):
    return conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], stride, (pad, pad), "reflect")


def cista_forward(sd, events, prev_image, prev_states, depth=5, prefix="cista_net."):
    """CistaLSTCNet.forward.  Returns (rec_I, [c, z, (h, cc)])."""
    p = prefix
    if prev_states is None:
        prev_states = [None, None, None]
    x_E = _cl(sd, p + "We.conv2d", events)
    x_I = _cl(sd, p + "Wi.conv2d", prev_image)
    x1 = _cl(sd, p + "W0.conv2d", torch.cat((x_E, x_I), 1), stride=2)
    # ConvLSTC
    z_prev, c_prev = prev_states[1], prev_states[0]
    B, _, h, w = x1.shape
    zc = sd[p + "P0.P0.weight"].shape[0]
    if z_prev is None:
        z_prev = torch.zeros(B, zc, h, w)
    gates = _cl(sd, p + "P0.gates", torch.cat((x1, z_prev), 1))
    in_gate, forget_gate = torch.sigmoid(gates[:, :zc]), torch.sigmoid(gates[:, zc:])
    z0 = _cl(sd, p + "P0.P0", x1)
    out_gate = torch.sigmoid(_cl(sd, p + "P0.out_gates", torch.cat((z0, z_prev), 1)))
    if c_prev is None:
        c_prev = torch.zeros_like(z0)
    c = forget_gate * c_prev + in_gate * z0
    z = out_gate * torch.tanh(c)
    # unrolled ISTA: lista_blocks.0..4 alias one IstaBlock
    lam = sd[p + "lista_blocks.0.Lambda"]
    tmp = z
    for i in range(depth):
        k = p + "lista_blocks.%d." % i
        tmp = _cl(sd, k + "D.conv2d", tmp)
        x = _cl(sd, k + "P.conv2d", x1 - tmp) + z
        z = softshrink(x, sd[k + "Lambda"])
        tmp = z
    # Dg: conv+relu then ConvLSTM (chunk order in, remember, out, cell)
    x = torch.relu(_cl(sd, p + "Dg.conv.conv2d", z))
    hc = x.shape[1]
    if prev_states[2] is None:
        h_prev, cc_prev = torch.zeros_like(x), torch.zeros_like(x)
    else:
        h_prev, cc_prev = prev_states[2]
    g = _cl(sd, p + "Dg.recurrent_block.Gates", torch.cat((x, h_prev), 1))
    ig, rg, og, cg = g[:, :hc], g[:, hc:2 * hc], g[:, 2 * hc:3 * hc], g[:, 3 * hc:]
    cc = torch.sigmoid(rg) * cc_prev + torch.sigmoid(ig) * torch.tanh(cg)
    hh = torch.sigmoid(og) * torch.tanh(cc)
    # upsample x2 (align_corners=False) + ReflectionPad2d(1) + conv(pad 0) + relu ; final conv + sigmoid
    up = interp_bilinear(hh, 2 * hh.shape[2], 2 * hh.shape[3], align_corners=False)
    up = torch.relu(F.conv2d(reflect_pad(up, 1, 1), sd[p + "upsamp_conv.conv2d.weight"], sd[p + "upsamp_conv.conv2d.bias"]))
    rec = torch.sigmoid(_cl(sd, p + "final_conv.conv2d", up))
    return rec, [c, z, (hh, cc)]


# ----------------------------------------------------------------------------------------------
# DCEIFlow  (a6 - a13)
# ----------------------------------------------------------------------------------------------


def image_pad(x, H, W, min_size=32):
    """ImagePadder.pad: zero pad TOP and LEFT up to a multiple of min_size."""
    ph = (min_size - H % min_size) % min_size
    pw = (min_size - W % min_size) % min_size
    return F.pad(x, (pw, 0, ph, 0)), ph, pw


def instance_norm(x, eps=1e-5):
    m = x.mean(dim=(2, 3), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - m) / torch.sqrt(v + eps)


def batch_norm_eval(x, sd, pre, eps=1e-5):
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    m, v = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + eps) * w[None, :, None, None] + b[None, :, None, None]


def encoder(sd, pre, x, norm):
    """BasicEncoder (ds=8).  norm: 'instance' | 'batch'."""

    def nrm(t, name):
        return instance_norm(t) if norm == "instance" else batch_norm_eval(t, sd, name)

    x = conv2d(x, sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], 2, (3, 3))
    x = torch.relu(nrm(x, pre + ".norm1"))
    for L, stride0 in ((1, 1), (2, 2), (3, 2)):
        for blk in range(2):
            k = "%s.layer%d.%d" % (pre, L, blk)
            stride = stride0 if blk == 0 else 1
            y = conv2d(x, sd[k + ".conv1.weight"], sd[k + ".conv1.bias"], stride, (1, 1))
            y = torch.relu(nrm(y, k + ".norm1"))
            y = conv2d(y, sd[k + ".conv2.weight"], sd[k + ".conv2.bias"], 1, (1, 1))
            y = torch.relu(nrm(y, k + ".norm2"))
            if stride != 1:
                x = conv2d(x, sd[k + ".downsample.0.weight"], sd[k + ".downsample.0.bias"], stride, (0, 0))
                x = nrm(x, k + ".downsample.1")
            x = torch.relu(x + y)
    return conv2d(x, sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], 1, (0, 0))


def corr_pyramid(fmap1, fmap2, levels=4):
    """all-pairs correlation / sqrt(D) and its 2x2 average pyramid: list of [B*N,1,h,w]."""
    B, D, h, w = fmap1.shape
    f1 = fmap1.reshape(B, D, h * w)
    f2 = fmap2.reshape(B, D, h * w)
    corr = torch.matmul(f1.transpose(1, 2), f2) * (1.0 / math.sqrt(D))
    corr = corr.reshape(B * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(levels - 1):
        hh, ww = corr.shape[-2] // 2, corr.shape[-1] // 2
        c = corr[..., : 2 * hh, : 2 * ww]
        corr = (c[..., 0::2, 0::2] + c[..., 0::2, 1::2] + c[..., 1::2, 0::2] + c[..., 1::2, 1::2]) / 4
        pyr.append(corr)
    return pyr


def corr_lookup(pyr, coords, radius=4):
    """channel lvl*81 + a*9 + b = zero-padded bilinear sample of level lvl at
    (x/2^lvl + a - r, y/2^lvl + b - r)  (first window axis steps x)."""
    B, _, h, w = coords.shape
    r = radius
    d = 2 * r + 1
    cx = coords[:, 0].reshape(B * h * w, 1, 1)
    cy = coords[:, 1].reshape(B * h * w, 1, 1)
    off = torch.arange(-r, r + 1, dtype=torch.float32)
    outs = []
    for lvl, corr in enumerate(pyr):
        Hl, Wl = corr.shape[-2:]
        x = cx / 2 ** lvl + off.view(1, d, 1)     # [Q, a, 1]
        y = cy / 2 ** lvl + off.view(1, 1, d)     # [Q, 1, b]
        gx = 2 * x / (Wl - 1) - 1
        gy = 2 * y / (Hl - 1) - 1
        ix = ((gx + 1) * ((Wl - 1) / 2)).expand(-1, d, d)
        iy = ((gy + 1) * ((Hl - 1) / 2)).expand(-1, d, d)
        x0, y0 = ix.floor(), iy.floor()
        tx, ty = ix - x0, iy - y0
        x0, y0 = x0.long(), y0.long()
        flat = corr.reshape(B * h * w, Hl * Wl)

        def tap(yy, xx):
            ok = ((xx >= 0) & (xx < Wl) & (yy >= 0) & (yy < Hl)).float()
            idx = (yy.clamp(0, Hl - 1) * Wl + xx.clamp(0, Wl - 1)).reshape(B * h * w, d * d)
            return flat.gather(1, idx).reshape(B * h * w, d, d) * ok

        s = tap(y0, x0) * ((1 - tx) * (1 - ty)) + tap(y0, x0 + 1) * (tx * (1 - ty)) \
            + tap(y0 + 1, x0) * ((1 - tx) * ty) + tap(y0 + 1, x0 + 1) * (tx * ty)
        outs.append(s.reshape(B, h, w, d * d))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous()


def coords_grid(B, h, w):
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([gx, gy], 0).float()[None].repeat(B, 1, 1, 1)


def update_block(sd, pre, net, inp, corr, emap, flow):
    """BasicUpdateBlockNoMask: BasicMotionEncoder + SepConvGRU + FlowHead."""
    e = pre + ".encoder."

    def c(name, x, pad):
        return conv2d(x, sd[name + ".weight"], sd[name + ".bias"], 1, pad)

    cor = torch.relu(c(e + "convc1", corr, (0, 0)))
    cor = torch.relu(c(e + "convc2", cor, (1, 1)))
    ema = torch.relu(c(e + "conve1", emap, (0, 0)))
    ema = torch.relu(c(e + "conve2", ema, (1, 1)))
    flo = torch.relu(c(e + "convf1", flow, (3, 3)))
    flo = torch.relu(c(e + "convf2", flo, (1, 1)))
    out = torch.relu(c(e + "conv", torch.cat([cor, ema, flo], 1), (1, 1)))
    x = torch.cat([inp, out, flow], 1)
    g = pre + ".gru."
    h = net
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(c(g + "convz" + sfx, hx, pad))
        r = torch.sigmoid(c(g + "convr" + sfx, hx, pad))
        q = torch.tanh(c(g + "convq" + sfx, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    fh = pre + ".flow_head."
    delta = c(fh + "conv2", torch.relu(c(fh + "conv1", h, (1, 1))), (1, 1))
    return h, delta


def eiflow_forward(sd, event_voxel, image1, iters=6, flow_init=None, prefix="event_flownet."):
    """DCEIFlow.forward (image2 / reversed voxel = None).  Returns dict like the reference."""
    p = prefix
    B, _, H, W = image1.shape
    image1 = 2 * image1 - 1.0
    image1, ph, pw = image_pad(image1, H, W)
    event_voxel, _, _ = image_pad(event_voxel, H, W)
    emap = encoder(sd, p + "enet", event_voxel, "instance")
    fmap1 = encoder(sd, p + "fnet", image1, "instance")
    # EIFusion
    c1 = torch.relu(conv2d(fmap1, sd[p + "fusion.conv1.weight"], sd[p + "fusion.conv1.bias"]))
    c2 = torch.relu(conv2d(emap, sd[p + "fusion.conv2.weight"], sd[p + "fusion.conv2.bias"]))
    pf2 = torch.relu(conv2d(torch.cat([c1, c2], 1), sd[p + "fusion.convo.weight"], sd[p + "fusion.convo.bias"], 1, (1, 1))) + fmap1
    pyr = corr_pyramid(fmap1, pf2)
    cnet = encoder(sd, p + "cnet", image1, "batch")
    net, inp = torch.tanh(cnet[:, :128]), torch.relu(cnet[:, 128:])
    h8, w8 = image1.shape[2] // 8, image1.shape[3] // 8
    coords0 = coords_grid(B, h8, w8)
    coords1 = coords_grid(B, h8, w8)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    preds = []
    flow_up = None
    for _ in range(iters):
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        net, delta = update_block(sd, p + "update_block", net, inp, corr, emap, flow)
        coords1 = coords1 + delta
        f = coords1 - coords0
        up = 8 * interp_bilinear(f, 8 * h8, 8 * w8, align_corners=True)
        preds.append(up)
        flow_up = up[..., ph:, pw:]
    return dict(flow_preds=preds, flow_init=coords1 - coords0, flow_final=flow_up)


def eiflow_step(sd, batch_data, states, warp_mode="forward", depth=5, iters=6, gt_flow=None):
    """DCEIFlowCistaNet.forward (a5).  Mutates states[1] in place like the reference."""
    bf = eiflow_forward(sd, batch_data["event_voxel"], batch_data["rec_img0"], iters=iters,
                        flow_init=batch_data.get("flow_init"))
    flow_final = bf["flow_final"] if gt_flow is None else gt_flow
    if not flow_final.any():
        warped_I = batch_data["rec_img0"]
    else:
        warped_I = warp(batch_data["rec_img0"], flow_final, warp_mode)
        if states is not None:
            H, W = flow_final.shape[-2:]
            dflow = interp_bilinear(flow_final, H // 2, W // 2, align_corners=True)   # values NOT halved
            states[1] = warp(states[1], dflow, warp_mode)
    I_rec, new_states = cista_forward(sd, batch_data["event_voxel"], warped_I, states, depth=depth)
    return I_rec, bf, new_states


# ----------------------------------------------------------------------------------------------
# E-RAFT  (a12, a14)   /root/reference/ERAFT/eraft.py:77-88,114-178 ; ERAFT/update.py:63-106 ;
#                      ERAFT/extractor.py:119-189 ; ERAFT/corr.py:12-60 ; e2v/e2v_model.py:206-248
# ----------------------------------------------------------------------------------------------


def convex_upsample(flow, mask):
    """ERAFT.upsample_flow: softmax over the 9 neighbours of mask [B,576,h,w] (channel k*64+i*8+j) applied to
    unfold(8*flow, 3x3, padding=1); output [B,2,8h,8w]."""
    B, _, h, w = flow.shape
    m = torch.softmax(mask.reshape(B, 1, 9, 8, 8, h, w), dim=2)
    fp = F.pad(8 * flow, (1, 1, 1, 1))
    nb = torch.stack([fp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], dim=2)   # [B,2,9,h,w]
    up = (m * nb.reshape(B, 2, 9, 1, 1, h, w)).sum(dim=2)            # [B,2,8,8,h,w]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * h, 8 * w)


def eraft_update_block(sd, pre, net, inp, corr, flow):
    """ERAFT BasicUpdateBlock: motion encoder (no emap branch) + SepConvGRU + FlowHead + mask head."""
    e = pre + ".encoder."

    def c(name, x, pad):
        return conv2d(x, sd[name + ".weight"], sd[name + ".bias"], 1, pad)

    cor = torch.relu(c(e + "convc1", corr, (0, 0)))
    cor = torch.relu(c(e + "convc2", cor, (1, 1)))
    flo = torch.relu(c(e + "convf1", flow, (3, 3)))
    flo = torch.relu(c(e + "convf2", flo, (1, 1)))
    out = torch.relu(c(e + "conv", torch.cat([cor, flo], 1), (1, 1)))
    x = torch.cat([inp, out, flow], 1)
    g = pre + ".gru."
    h = net
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(c(g + "convz" + sfx, hx, pad))
        r = torch.sigmoid(c(g + "convr" + sfx, hx, pad))
        q = torch.tanh(c(g + "convq" + sfx, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    fh = pre + ".flow_head."
    delta = c(fh + "conv2", torch.relu(c(fh + "conv1", h, (1, 1))), (1, 1))
    mask = 0.25 * c(pre + ".mask.2", torch.relu(c(pre + ".mask.0", h, (1, 1))), (0, 0))
    return h, mask, delta


def eraft_forward(sd, image1, image2, iters=12, flow_init=None, prefix="event_flownet."):
    """ERAFT.forward: image1 / image2 are the old / new event voxel grids."""
    p = prefix
    B, _, H, W = image1.shape
    image1, ph, pw = image_pad(image1, H, W)
    image2, _, _ = image_pad(image2, H, W)
    fmap1 = encoder(sd, p + "fnet", image1, "instance")
    fmap2 = encoder(sd, p + "fnet", image2, "instance")
    pyr = corr_pyramid(fmap1, fmap2)
    cnet = encoder(sd, p + "cnet", image2, "batch")
    net, inp = torch.tanh(cnet[:, :128]), torch.relu(cnet[:, 128:])
    h8, w8 = image1.shape[2] // 8, image1.shape[3] // 8
    coords0 = coords_grid(B, h8, w8)
    coords1 = coords_grid(B, h8, w8)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    preds = []
    flow_up = None
    for _ in range(iters):
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        net, mask, delta = eraft_update_block(sd, p + "update_block", net, inp, corr, flow)
        coords1 = coords1 + delta
        up = convex_upsample(coords1 - coords0, mask)
        preds.append(up)
        flow_up = up[..., ph:, pw:]
    return dict(flow_preds=preds, flow_init=coords1 - coords0, flow_final=flow_up)


def eraft_step(sd, batch_data, states, warp_mode="forward", depth=5, iters=12, gt_flow=None):
    """ERAFTCistaNet.forward: flow from (event_voxel_old, event_voxel), then the shared warp + CISTA tail."""
    bf = eraft_forward(sd, batch_data["event_voxel_old"], batch_data["event_voxel"], iters=iters)
    flow_final = bf["flow_final"] if gt_flow is None else gt_flow
    if not flow_final.any():
        warped_I = batch_data["rec_img0"]
    else:
        warped_I = warp(batch_data["rec_img0"], flow_final, warp_mode)
        if states is not None:
            H, W = flow_final.shape[-2:]
            dflow = interp_bilinear(flow_final, H // 2, W // 2, align_corners=True)
            states[1] = warp(states[1], dflow, warp_mode)
    I_rec, new_states = cista_forward(sd, batch_data["event_voxel"], warped_I, states, depth=depth)
    return I_rec, bf, new_states


# ----------------------------------------------------------------------------------------------
# IDNet  (a15)   /root/reference/idn/idedeq.py:48-61,74-92,124-227 ; idn/extractor.py:63-125 ;
#                idn/update.py:29-85 ; e2v/e2v_model.py:252-308
# ----------------------------------------------------------------------------------------------


def idn_deblur(x, flow):
    """deblur_tensor (voxel mode): bin t sampled at p + flow*t/(T-1); grid normalised with (W-1) but
    grid_sample(align_corners=False, zeros) -> pixel x*W/(W-1) - 0.5 (zero flow is not an identity)."""
    B, T, H, W = x.shape
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    out = torch.zeros_like(x)
    for t in range(T):
        dp = flow * t / (T - 1)
        sx = (gx[None].float() + dp[:, 0]) / (W - 1) * 2 - 1
        sy = (gy[None].float() + dp[:, 1]) / (H - 1) * 2 - 1
        ix = (sx + 1) * (W / 2) - 0.5
        iy = (sy + 1) * (H / 2) - 0.5
        x0, y0 = ix.floor(), iy.floor()
        tx, ty = ix - x0, iy - y0
        x0, y0 = x0.long(), y0.long()
        flat = x[:, t].reshape(B, H * W)

        def tap(yy, xx):
            ok = ((xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)).float()
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, H * W)
            return flat.gather(1, idx).reshape(B, H, W) * ok

        out[:, t] = tap(y0, x0) * ((1 - tx) * (1 - ty)) + tap(y0, x0 + 1) * (tx * (1 - ty)) \
            + tap(y0 + 1, x0) * ((1 - tx) * ty) + tap(y0 + 1, x0 + 1) * (tx * ty)
    return out


def lite_encoder(sd, pre, x):
    """LiteEncoder(stride=2): conv 7x7 s2 + relu, two stages of norm-free residual blocks (s2 each)."""

    def c(name, t, stride, pad):
        return conv2d(t, sd[name + ".weight"], sd[name + ".bias"], stride, (pad, pad))

    x = torch.relu(c(pre + ".conv1", x, 2, 3))
    for L in (1, 2):
        for blk in range(2):
            k = "%s.layer%d.%d" % (pre, L, blk)
            stride = 2 if blk == 0 else 1
            y = torch.relu(c(k + ".conv1", x, stride, 1))
            y = torch.relu(c(k + ".conv2", y, 1, 1))
            if stride != 1:
                x = c(k + ".downsample.0", x, stride, 0)
            x = torch.relu(x + y)
    return x


def idnet_forward(sd, event_bins, flow_init=None, prefix="event_flownet."):
    """IDEDEQIDO.forward with update_iters=1, pred_next_flow=True."""
    p = prefix
    B, T, H, W = event_bins.shape
    x_raw, ph, pw = image_pad(event_bins, H, W)
    Hp, Wp = x_raw.shape[-2:]
    flow_total = torch.zeros(B, 2, Hp, Wp) if flow_init is None else flow_init.clone()
    delta0 = flow_total
    x_deblur = idn_deblur(x_raw, flow_total)
    net = torch.zeros(B, 96, Hp // 8, Wp // 8)
    u = p + "update_net."

    def c(name, t, pad):
        return conv2d(t, sd[name + ".weight"], sd[name + ".bias"], 1, (pad, pad))

    for t in range(T):
        sl = torch.stack([x_deblur[:, t], x_deblur[:, t]], dim=1)       # the two input channels are identical
        f = lite_encoder(sd, p + "fnet", sl)
        hx = torch.cat([net, f], 1)
        z = torch.sigmoid(c(u + "gru.convz", hx, 1))
        r = torch.sigmoid(c(u + "gru.convr", hx, 1))
        q = torch.tanh(c(u + "gru.convq", torch.cat([r * net, f], 1), 1))
        net = (1 - z) * net + z * q

    def head(fh, mk):
        d = c(u + fh + ".conv2", torch.relu(c(u + fh + ".conv1", net, 1)), 1)
        m = c(u + mk + ".2", torch.relu(c(u + mk + ".0", net, 1)), 0)
        return convex_upsample(d, m)

    delta = head("flow_head", "mask")
    next_flow = head("flow_head2", "mask2")
    flow_total = flow_total + delta
    return dict(flow_final=flow_total[..., ph:, pw:], next_flow=next_flow, delta_flow=torch.stack([delta0, delta], 1),
                flow_preds=[flow_total])


def idnet_step(sd, batch_data, states, flow_init=None, warp_mode="forward", depth=5, gt_flow=None):
    """IDCistaNet.forward."""
    bf = idnet_forward(sd, batch_data["event_voxel"], flow_init)
    flow_final = bf["flow_final"] if gt_flow is None else gt_flow
    if not flow_final.any():
        warped_I = batch_data["rec_img0"]
    else:
        warped_I = warp(batch_data["rec_img0"], flow_final, warp_mode)
        if states is not None:
            H, W = flow_final.shape[-2:]
            dflow = interp_bilinear(flow_final, H // 2, W // 2, align_corners=True)
            states[1] = warp(states[1], dflow, warp_mode)
    I_rec, new_states = cista_forward(sd, batch_data["event_voxel"], warped_I, states, depth=depth)
    return I_rec, bf, new_states


# ----------------------------------------------------------------------------------------------
# f-1: events -> voxel grid   /root/reference/utils/event_process.py:15-72, 193-216
# ----------------------------------------------------------------------------------------------


def events_to_voxel(events, num_bins, width, height, normalize=True, filter_hot_pixel=False):
    """events: numpy [N,4] float64 (t, x, y, polarity).  Temporal-bilinear accumulation then, if `normalize`,
    the 'std' pre-processing: non-zero voxels to mean 0 / std 1, zeros stay zero.  filter_hot_pixel: voxels with
    |v| > 25 / num_bins are zeroed first (event_process.py:196-198)."""
    import numpy as np
    vox = np.zeros(num_bins * height * width, np.float32)
    if len(events):
        t = events[:, 0]
        dT = t[-1] - t[0]
        if dT == 0:
            dT = 1.0
        ts = (num_bins - 1) * (t - t[0]) / dT
        xs = events[:, 1].astype(np.int64)
        ys = events[:, 2].astype(np.int64)
        pol = np.where(events[:, 3] == 0, -1.0, events[:, 3])
        ti = ts.astype(np.int64)
        dt = ts - ti
        for k in range(len(events)):          # sequential like np.add.at
            if ti[k] < num_bins:
                vox[xs[k] + ys[k] * width + ti[k] * width * height] += pol[k] * (1.0 - dt[k])
            if ti[k] + 1 < num_bins:
                vox[xs[k] + ys[k] * width + (ti[k] + 1) * width * height] += pol[k] * dt[k]
    vox = vox.reshape(num_bins, height, width)
    if filter_hot_pixel:
        vox[np.abs(vox) > np.float32(25.0 / num_bins)] = 0
    if normalize:
        nz = vox != 0
        n = nz.sum()
        if n > 0:
            mean = vox.sum() / n
            sd = np.sqrt((vox ** 2).sum() / n - mean ** 2)
            vox = (nz.astype(np.float32) * (vox - mean) / (sd + 1e-8)).astype(np.float32)
    return vox


# ----------------------------------------------------------------------------------------------
# f-3: evaluation metrics -- restated from /root/reference/loss.py
# ----------------------------------------------------------------------------------------------
def recon_metrics(rec, tgt):
    """nn.MSELoss + PSNR(data_range=1)   loss.py:15-24, 316-328 (ReconLoss.evaluate without ssim / lpips)."""
    mse = ((rec.double() - tgt.double()) ** 2).mean()
    psnr = 100.0 if mse < 1.0e-10 else 20.0 * math.log10(1.0 / math.sqrt(float(mse)))
    return float(mse), float(psnr)


def flow_metrics(flow_final, gt_flow, gt_img0, gt_img1, flow_valid=None, warp_mode="forward", max_flow=400.0):
    """FlowL1LossDict.evaluate   loss.py:237-265 -> (photo_loss, epe, 1px, 3px, 5px, out).  Per-pixel gt magnitude
    (the reference's epe [B,H,W] / mag [B,1,H,W] broadcast only works at batch 1, where it means the same)."""
    if flow_valid is None:
        flow_valid = torch.exp(-50 * (warp(gt_img0, gt_flow, warp_mode) - gt_img1) ** 2)       # :241
    mag = torch.sum(gt_flow ** 2, dim=1, keepdim=True).sqrt()                                    # :243
    valid = flow_valid * (mag < max_flow).float()                                                # :244
    photo = (warp(gt_img0, flow_final, warp_mode) - gt_img1).abs().double().mean()              # :247
    epe = torch.sum(valid * (flow_final - gt_flow) ** 2, dim=1, keepdim=True).sqrt()            # :248
    out = ((epe > 3.0) & ((epe / mag) > 0.05)).double()                                          # :250
    sel = valid > 0
    e = epe[sel].double()
    return (float(photo), float(e.mean()), float((e > 1).double().mean()), float((e > 3).double().mean()),
            float((e > 5).double().mean()), float(out[sel].mean() * 100))


def fwl_sample_zeros(plane, ix, iy):
    """grid_sample(bilinear, align_corners=True, padding_mode='zeros') of [B,H,W] planes at pixel coords."""
    B, H, W = plane.shape
    x0, y0 = torch.floor(ix), torch.floor(iy)
    tx, ty = ix - x0, iy - y0
    out = torch.zeros_like(ix)
    bidx = torch.arange(B).view(B, 1, 1).expand_as(ix)
    for dy, wy in ((0, 1 - ty), (1, ty)):
        for dx, wx in ((0, 1 - tx), (1, tx)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
            v = plane[bidx, yy.clamp(0, H - 1).long(), xx.clamp(0, W - 1).long()]
            out = out + torch.where(ok, v * (wy * wx), torch.zeros_like(v))
    return out


def voxel_warping_flow_loss(voxel, displacement):
    """loss.py:27-83: channel i of the voxel grid sampled at (x + dx*i/(C-1), y + dy*i/(C-1)), grid normalised by W (not
    W-1) and sampled with align_corners=True / zeros padding, summed over i; the loss is the variance of that image."""
    B, C, H, W = voxel.shape
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xx, yy = xx.float(), yy.float()
    inc = 1.0 / (C - 1.0)
    acc = torch.zeros(B, H, W)
    for i in range(C):
        r = i * inc
        gx = (2.0 * (xx + displacement[:, 0] * r)) / W - 1.0
        gy = (2.0 * (yy + displacement[:, 1] * r)) / H - 1.0
        ix = ((gx + 1.0) / 2.0) * (W - 1)
        iy = ((gy + 1.0) / 2.0) * (H - 1)
        acc = acc + fwl_sample_zeros(voxel[:, i], ix, iy)
    return float(acc.double().var())


def ssim(X, Y, data_range=1.0, win_size=11, win_sigma=1.5, K=(0.01, 0.03), dtype=torch.float32):
    """pytorch_msssim.ssim / SSIM(data_range=1, size_average=True, nonnegative_ssim=False) as ReconLoss builds it
    (loss.py:314,319).  The package (third-party dependency of the reference, no version pinned by it; algorithm of
    pytorch_msssim >= 0.2: ssim.py::_fspecial_gauss_1d, gaussian_filter, _ssim) is NOT installed offline, so this is a
    restatement of its published algorithm: 1-D gaussian window, separable 'valid' convolution (H direction first, then W),
    C1 = (K1 L)^2, C2 = (K2 L)^2, mean of the SSIM map over the valid positions of every (b, c) plane, then over planes.
    Parity for this metric is therefore unpinned by the reference.  Returns (ssim, cs) as Python floats."""
    X, Y = X.to(dtype), Y.to(dtype)
    coords = torch.arange(win_size, dtype=torch.float32) - win_size // 2
    g = torch.exp(-(coords ** 2) / (2 * win_sigma ** 2))
    g = (g / g.sum()).to(dtype)
    C = X.shape[1]

    def gauss(t):
        t = F.conv2d(t, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
        return F.conv2d(t, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)

    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gauss(X), gauss(Y)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1, s2, s12 = gauss(X * X) - mu1_sq, gauss(Y * Y) - mu2_sq, gauss(X * Y) - mu1_mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return float(ssim_map.flatten(2).mean(-1).mean()), float(cs_map.flatten(2).mean(-1).mean())


def merge_optical_flow(flow):
    """FlowWriter's colour coding, utils/data_io.py:9-29, restated in numpy float32 from OpenCV's PUBLISHED arithmetic (UNPINNED: cv2 is
    absent, so no reference-run vector exists): cartToPolar -> angle in [0, 2 pi) and magnitude; H = uint8(angle * 180 / pi / 2) and
    V = uint8(255 * magnitude / magnitude.max()) with numpy's truncating casts, S = 255; cvtColor(HSV2BGR) for 8-bit images:
    h6 = H / 30, sector = floor(h6), f = h6 - sector, tab = [v, v (1 - s), v (1 - s f), v (1 - s (1 - f))] with s, v in [0, 1], channels from the
    sector table, rounded half-to-even to uint8.  flow: [2, H, W] array -> [H, W, 3] uint8 (B, G, R)."""
    import numpy as np
    fx = np.asarray(flow[0], dtype=np.float32)
    fy = np.asarray(flow[1], dtype=np.float32)
    mag = np.sqrt(fx * fx + fy * fy).astype(np.float32)
    ang = np.arctan2(fy, fx).astype(np.float32)
    ang = np.where(ang < 0, ang + np.float32(2 * np.pi), ang).astype(np.float32)
    H = (ang * np.float32(180.0) / np.float32(np.pi) / np.float32(2.0)).astype(np.int32) & 255
    mx = mag.max()
    V = (np.float32(255.0) * mag / mx).astype(np.int32) if mx > 0 else np.zeros_like(H)
    v = V.astype(np.float32) * np.float32(1.0 / 255.0)
    h6 = H.astype(np.float32) * np.float32(6.0 / 180.0)
    sector = np.floor(h6).astype(np.int32)
    f = h6 - sector.astype(np.float32)
    sector = sector % 6
    tab = np.stack([v, np.zeros_like(v), v * (np.float32(1.0) - f), v * f], axis=0)        # s = 1
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    out = np.zeros(H.shape + (3,), dtype=np.uint8)
    yy, xx = np.indices(H.shape)
    for c in range(3):
        t = tab[sd[sector, c], yy, xx] * np.float32(255.0)
        out[..., c] = np.rint(np.clip(t, 0, 255)).astype(np.uint8)
    return out
