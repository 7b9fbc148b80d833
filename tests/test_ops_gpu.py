"""Kernel-level parity: every HIP operator of libcistaflow against a plain PyTorch fp32 CPU
reference of the same op (tolerance 1e-4 absolute on O(1) data unless stated)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from cista_flow_amd import lib
    return lib.load(), lib


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def ref_conv(x, w, b, stride, padT, padL, pad_mode):
    if pad_mode == 1:
        xp = F.pad(x, (padL, padL, padT, padT), mode="reflect") if (padT or padL) else x
    else:
        xp = F.pad(x, (padL, padL, padT, padT))
    return F.conv2d(xp, w, b, stride=stride)


def run_conv(gpu, x_in, w, b, stride, padT, padL, pad_mode, a_mode, epi, tile, Ho, Wo):
    L, lib = _lib()
    B = x_in.shape[0]
    Cout, Cin, KH, KW = w.shape
    if a_mode == 2:
        xin = x_in.contiguous().to(gpu)
        H, W = x_in.shape[2], x_in.shape[3]
    else:
        xin = nhwc(x_in).to(gpu)
        H, W = x_in.shape[2], x_in.shape[3]
    wg, bg = w.contiguous().to(gpu), (b.contiguous().to(gpu) if b is not None else None)
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=gpu)
    rc = L.cf_op_conv2d(lib.ptr(xin), B, Cin, H, W, lib.ptr(wg), lib.ptr(bg), Cout, KH, KW, stride, padT, padL,
                        pad_mode, a_mode, epi, tile, lib.ptr(out), lib.current_stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return nchw(out.cpu())


CONV_CASES = [
    # Cin, Cout, KH, KW, stride, padT, padL, pad_mode, tile
    (64, 64, 3, 3, 1, 1, 1, 1, 0),
    (64, 64, 3, 3, 2, 1, 1, 1, 0),
    (192, 256, 3, 3, 1, 1, 1, 1, 0),
    (128, 64, 3, 3, 1, 1, 1, 1, 0),
    (64, 128, 3, 3, 1, 1, 1, 1, 0),
    (256, 128, 3, 3, 1, 1, 1, 1, 0),
    (16, 96, 3, 3, 1, 1, 1, 0, 0),
    (64, 96, 1, 1, 2, 0, 0, 0, 0),
    (96, 96, 3, 3, 2, 1, 1, 0, 0),
    (384, 128, 1, 5, 1, 0, 2, 0, 0),
    (384, 128, 5, 1, 1, 2, 0, 0, 0),
    (128, 256, 3, 3, 1, 1, 1, 0, 0),
    (256, 2, 3, 3, 1, 1, 1, 0, 0),
    (64, 1, 3, 3, 1, 1, 1, 1, 0),
    (320, 126, 3, 3, 1, 1, 1, 0, 0),
    (336, 256, 1, 1, 1, 0, 0, 0, 0),
    (128, 576, 1, 1, 1, 0, 0, 0, 0),
    (64, 128, 3, 3, 1, 1, 1, 1, 1),
    (64, 128, 3, 3, 1, 1, 1, 1, 2),
    (64, 128, 3, 3, 1, 1, 1, 1, 3),
    (64, 128, 3, 3, 1, 1, 1, 1, 4),
    (64, 128, 3, 3, 1, 1, 1, 1, 5),
    (64, 128, 3, 3, 1, 1, 1, 1, 6),
    (256, 2, 3, 3, 1, 1, 1, 0, 6),     # small-cout layers forced through the MFMA path
    (64, 1, 3, 3, 1, 1, 1, 1, 6),
    (256, 2, 3, 3, 1, 1, 1, 0, 7),     # ... and through the VALU small-N kernel explicitly
    (64, 1, 3, 3, 1, 1, 1, 1, 7),
    (128, 2, 3, 3, 1, 1, 1, 0, 7),
    (256, 2, 3, 3, 1, 1, 1, 0, 15),    # ... and its 3x3 register-window form (weights + 3 input rows in registers)
    (64, 1, 3, 3, 1, 1, 1, 1, 15),
    (128, 2, 3, 3, 1, 1, 1, 1, 15),
    (64, 2, 3, 3, 1, 1, 1, 0, 15),
    (128, 128, 3, 3, 1, 1, 1, 0, 8),   # intra-workgroup split-K tiles (32x32xK4, 32x64xK2)
    (64, 96, 3, 3, 1, 1, 1, 1, 8),
    (256, 126, 1, 5, 1, 0, 2, 0, 8),
    (128, 256, 3, 3, 1, 1, 1, 0, 9),
    (96, 96, 3, 3, 2, 1, 1, 0, 9),
    (384, 128, 5, 1, 1, 2, 0, 0, 9),
    (128, 256, 3, 3, 1, 1, 1, 0, 10),  # wide-stage (32 k-columns per wave) variants
    (256, 128, 1, 5, 1, 0, 2, 0, 11),
    (96, 96, 3, 3, 1, 1, 1, 0, 12),
    (192, 256, 3, 3, 1, 1, 1, 1, 13),
    (128, 64, 3, 3, 2, 1, 1, 1, 14),
    (128, 256, 3, 3, 1, 1, 1, 0, 20),  # LDS-DMA staged variants (3-deep ring, swizzled rows)
    (96, 96, 3, 3, 2, 1, 1, 0, 20),
    (384, 126, 5, 1, 1, 2, 0, 0, 20),
    (128, 256, 3, 3, 1, 1, 1, 1, 21),
    (256, 128, 1, 5, 1, 0, 2, 0, 22),
    (64, 96, 3, 3, 1, 1, 1, 1, 22),
    (96, 96, 3, 3, 1, 1, 1, 0, 23),
    (16, 96, 3, 3, 1, 1, 1, 0, 23),
    (64, 128, 3, 3, 2, 1, 1, 1, 24),
    (192, 256, 3, 3, 1, 1, 1, 1, 25),
    (336, 256, 1, 1, 1, 0, 0, 0, 25),
    (128, 64, 3, 3, 1, 1, 1, 1, 26),
    (64, 128, 3, 3, 1, 1, 1, 0, 27),
    (96, 256, 3, 3, 1, 1, 1, 0, 28),
    (256, 576, 1, 1, 1, 0, 0, 0, 29),
    (64, 96, 3, 3, 1, 1, 1, 1, 29),
    (32, 32, 3, 3, 1, 1, 1, 0, 30),
    (64, 32, 3, 3, 2, 1, 1, 1, 30),
    (384, 256, 1, 5, 1, 0, 2, 0, 31),  # 4-deep ring (three stages in flight)
    (32, 64, 3, 3, 1, 1, 1, 1, 31),    # fewer stages than the ring is deep
    (96, 96, 3, 3, 1, 1, 1, 0, 32),
    (16, 64, 1, 1, 1, 0, 0, 0, 32),    # a single stage
    (128, 64, 3, 3, 2, 1, 1, 1, 33),
    (96, 96, 3, 3, 1, 1, 1, 0, 34),    # 32x96 split-K tile (three sub-tiles per wave) for the 96-channel stage
    (64, 96, 3, 3, 2, 1, 1, 0, 34),
    (96, 192, 1, 1, 1, 0, 0, 1, 34),
]


STATS_CASES = [
    # Cin, Cout, KH, KW, stride, pad, a_mode, tile, H, W   (the encoder convs that feed an InstanceNorm)
    (64, 64, 3, 3, 1, 1, 0, 0, 48, 64),
    (64, 96, 3, 3, 2, 1, 0, 0, 48, 64),
    (64, 96, 1, 1, 2, 0, 0, 0, 48, 64),
    (96, 128, 3, 3, 1, 1, 0, 0, 13, 19),     # ragged: last 32-pixel patch partly outside the image
    (128, 128, 3, 3, 1, 1, 0, 20, 24, 32),
    (128, 128, 3, 3, 1, 1, 0, 22, 24, 32),
    (64, 64, 3, 3, 1, 1, 0, 23, 21, 37),
    (64, 64, 3, 3, 1, 1, 0, 26, 40, 40),
    (64, 64, 3, 3, 1, 1, 0, 2, 21, 37),      # register-staged kernel
    (64, 96, 3, 3, 1, 1, 0, 9, 21, 37),
    (5, 64, 7, 7, 2, 3, 2, 0, 64, 80),       # conv1 of the encoders: planar gather mode
    (96, 96, 3, 3, 1, 1, 0, 34, 24, 32),     # 32x96 split-K tile (three sub-tiles per wave), register reduction path
    (64, 96, 3, 3, 2, 1, 0, 0, 96, 128),     # ... as the launcher picks it for the 96-channel stage
    (64, 64, 3, 3, 1, 1, 0, 40, 48, 64),     # Winograd tile: one partial per tile row of an 8 x 16 region
    (96, 96, 3, 3, 1, 1, 0, 40, 21, 37),     # ... ragged regions
    (96, 96, 3, 3, 1, 1, 0, 44, 21, 37),     # split-K Winograd (2 / 4 wave groups)
    (128, 128, 3, 3, 1, 1, 0, 45, 24, 32),
    (64, 64, 3, 3, 1, 1, 0, 42, 48, 64),     # F(4x4,3x3): sixteen partials per 32-tile region
    (96, 96, 3, 3, 1, 1, 0, 42, 21, 37),     # ... ragged regions
    (64, 160, 3, 3, 1, 1, 0, 42, 17, 17),    # ... a small ragged map with B * Cout > 128 (partial-buffer sizing)
    (64, 160, 3, 3, 1, 1, 0, 40, 17, 17),
    (64, 64, 3, 3, 1, 1, 0, 48, 48, 64),     # persistent Winograd workgroups: the partials of every item a walker carries
    (96, 96, 3, 3, 1, 1, 0, 48, 21, 37),
    (64, 64, 3, 3, 1, 1, 0, 49, 48, 64),     # ... software-pipelined variant
    (96, 96, 3, 3, 1, 1, 0, 49, 21, 37),
    (96, 128, 3, 3, 1, 1, 0, 47, 13, 19),    # conv_wino16_kernel: two statistics patches per 8 x 8 region, ragged
    (64, 64, 3, 3, 1, 1, 0, 47, 24, 32),
    (96, 128, 3, 3, 1, 1, 0, 50, 13, 19),    # ... deep-prefetch instantiation
    (64, 64, 3, 3, 1, 1, 0, 50, 24, 32),
]


@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_fused_inorm_stats(gpu, case):
    """conv epilogue statistics == mean / biased var of the conv output (InstanceNorm2d, raft_encoder.py:32-36)."""
    Cin, Cout, KH, KW, stride, pad, a_mode, tile, H, W = case
    L, lib = _lib()
    g = torch.Generator().manual_seed(sum(case))
    B = 3
    x = torch.randn(B, Cin, H, W, generator=g) + 0.5
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g) * 2.0
    ref = ref_conv(x, w, b, stride, pad, pad, 0)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xin = (x.contiguous() if a_mode == 2 else nhwc(x)).to(gpu)
    wg, bg = w.to(gpu), b.to(gpu)
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=gpu)
    stats = torch.full((B, Cout, 2), float("nan"), device=gpu)
    eps = 1e-5
    rc = L.cf_op_conv2d_inorm_stats(lib.ptr(xin), B, Cin, H, W, lib.ptr(wg), lib.ptr(bg), Cout, KH, KW, stride, pad, pad, 0,
                                    a_mode, tile, lib.ptr(out), lib.ptr(stats), eps, lib.current_stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = nchw(out.cpu())
    assert (got - ref).abs().max().item() < 1e-4
    # statistics of the values the kernel actually stored
    g64 = got.double()
    mean = g64.mean(dim=(2, 3))
    var = g64.var(dim=(2, 3), unbiased=False)
    st = stats.cpu().double()
    assert (st[:, :, 0] - mean).abs().max().item() < 1e-6
    assert ((st[:, :, 1] - 1.0 / torch.sqrt(var + eps)) / (1.0 / torch.sqrt(var + eps))).abs().max().item() < 1e-6


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_nhwc(gpu, case):
    Cin, Cout, KH, KW, stride, padT, padL, pad_mode, tile = case
    g = torch.Generator().manual_seed(hash(case) % 100000)
    B, H, W = 2, 20, 28
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = ref_conv(x, w, b, stride, padT, padL, pad_mode)
    got = run_conv(gpu, x, w, b, stride, padT, padL, pad_mode, 0, 0, tile, ref.shape[2], ref.shape[3])
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < 1e-4, err


WINO_CASES = [(64, 64, 1, 20, 28, 0), (128, 64, 0, 20, 28, 1), (16, 32, 1, 13, 21, 2), (48, 96, 0, 24, 40, 3),
              (256, 126, 0, 24, 32, 1), (192, 256, 1, 17, 33, 0), (64, 128, 1, 90, 120, 0), (48, 128, 1, 23, 30, 2),
              (256, 128, 0, 26, 19, 3)]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd(gpu, case):
    """conv_wino_kernel (tile 40): Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1, reflect and zero padding, partial
    regions (sizes that are not multiples of 8 x 16), a cout that is not a multiple of 32 and the fused activations --
    against F.conv2d, and against the direct kernel to ~1e-6 of tensor scale (the transforms' rounding)."""
    Cin, Cout, pad_mode, H, W, epi = case
    g = torch.Generator().manual_seed(1000 + Cin + Cout + H)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = ref_conv(x, w, b, 1, 1, 1, pad_mode)
    ref = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[epi](ref)
    got = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 40, H, W)
    direct = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 0, H, W)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-4
    assert (got - direct).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # persistent workgroups (tile 48, conv_wino_p_kernel): the same arithmetic, bit for bit -- with the default grid (one item per
    # walker at these sizes) and with the grid shrunk to one walker per (XCD, n-block), which then walks its XCD's whole run of regions
    # ... and the same with the software-pipelined chunk loop (tile 49: operands of chunk k + 1 built between the MFMAs of chunk k)
    import os
    for ptile in (48, 49):
        gp = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, ptile, H, W)
        assert torch.equal(gp, got), ptile
        os.environ["CF_WINOP_SLOTS"] = "8"
        try:
            gp1 = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, ptile, H, W)
        finally:
            del os.environ["CF_WINOP_SLOTS"]
        assert torch.equal(gp1, got), ptile
    for sk_tile in (44, 45):                   # chunks split over 2 / 4 wave groups of a workgroup (odd chunk counts: dead steps)
        gsk = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, sk_tile, H, W)
        assert (gsk - ref).abs().max().item() < 1e-4, sk_tile
        assert (gsk - got).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), sk_tile
        assert torch.equal(gsk, run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, sk_tile, H, W))
    g16 = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 47, H, W)      # half-size workgroups on the 16x16x4 MFMA (conv_wino16_kernel)
    assert (g16 - ref).abs().max().item() < 1e-4
    assert (g16 - got).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(g16, run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 47, H, W))
    # ... and its deep-prefetch / software-pipelined instantiation for launches of at most one workgroup per CU (tile 50): bit for bit
    assert torch.equal(run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 50, H, W), g16)


WINO4_CASES = WINO_CASES + [(64, 64, 1, 96, 128, 1), (128, 64, 1, 37, 50, 0), (16, 40, 0, 12, 12, 0), (80, 32, 1, 16, 32, 0),
                            (64, 128, 0, 45, 16, 0), (64, 128, 0, 16, 45, 0)]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_conv_winograd_f4x4(gpu, case):
    """conv_wino4_kernel (tile 42): Winograd F(4x4,3x3), both region orientations (16 x 32 and 32 x 16 output pixels), ragged
    regions, reflect and zero padding, partial output-channel blocks, channel counts that are multiples of 8 only, fused
    activations -- against F.conv2d (fp32) and against the direct kernel.  F(4x4,3x3)'s transforms round more than F(2x2,3x3)'s
    (interpolation points +-2: coefficients up to 8 and 1/24): 1e-5 of the tensor scale per layer is what it costs, 4e-5 asserted."""
    Cin, Cout, pad_mode, H, W, epi = case
    g = torch.Generator().manual_seed(2000 + Cin + Cout + H)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref64 = ref_conv(x.double(), w.double(), b.double(), 1, 1, 1, pad_mode)
    act = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[epi]
    scale = max(1.0, ref64.abs().max().item())
    ref = act(ref64).float()
    got = run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 42, H, W)
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 4e-5 * scale, (got - ref).abs().max().item()
    assert torch.equal(got, run_conv(gpu, x, w, b, 1, 1, 1, pad_mode, 0, epi, 42, H, W))     # deterministic


WINO1D_CASES = [
    # Cin, Cout, horizontal, H, W, pad_mode, epi
    (256, 256, True, 23, 30, 0, 0), (256, 128, False, 23, 30, 0, 0), (64, 96, True, 9, 13, 0, 3), (64, 96, False, 9, 13, 0, 2),
    (16, 40, True, 5, 7, 0, 0), (16, 40, False, 7, 5, 0, 1), (128, 32, True, 60, 80, 1, 0), (128, 32, False, 60, 80, 1, 0),
    (48, 64, True, 1, 4, 0, 0), (48, 64, False, 4, 1, 0, 0), (32, 32, True, 3, 65, 0, 0),
]


@pytest.mark.parametrize("case", WINO1D_CASES)
def test_conv_winograd_1d(gpu, case):
    """conv_wino1d_kernel (tile 46): one-dimensional Winograd F(2,5) for the separable GRU's 1x5 / 5x1 convolutions -- odd and even line
    lengths, lines shorter than a window, tile groups that straddle lines and end ragged, zero and reflect padding, partial
    output-channel blocks, fused activations -- against an fp64 convolution and the direct kernel; bit-reproducible."""
    Cin, Cout, horiz, H, W, pad_mode, epi = case
    g = torch.Generator().manual_seed(4600 + Cin + Cout + H + W)
    B = 3
    KH, KW, pT, pL = (1, 5, 0, 2) if horiz else (5, 1, 2, 0)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * 5) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref64 = ref_conv(x.double(), w.double(), b.double(), 1, pT, pL, pad_mode)
    act = {0: lambda t: t, 1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[epi]
    scale = max(1.0, ref64.abs().max().item())
    ref = act(ref64).float()
    got = run_conv(gpu, x, w, b, 1, pT, pL, pad_mode, 0, epi, 46, H, W)
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 1e-5 * scale, (got - ref).abs().max().item()
    direct = run_conv(gpu, x, w, b, 1, pT, pL, pad_mode, 0, epi, 4, H, W)       # conv_igemm_kernel<64,64>
    assert (got - direct).abs().max().item() < 1e-5 * scale
    assert torch.equal(got, run_conv(gpu, x, w, b, 1, pT, pL, pad_mode, 0, epi, 46, H, W))     # deterministic


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_conv_epilogue_activation(gpu, epi):
    g = torch.Generator().manual_seed(7 + epi)
    x = torch.randn(1, 32, 9, 13, generator=g)
    w = torch.randn(48, 32, 3, 3, generator=g) / 17.0
    b = torch.randn(48, generator=g)
    ref = ref_conv(x, w, b, 1, 1, 1, 1)
    ref = {1: torch.relu, 2: torch.sigmoid, 3: torch.tanh}[epi](ref)
    got = run_conv(gpu, x, w, b, 1, 1, 1, 1, 0, epi, 0, 9, 13)
    assert (got - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("case", [(5, 32, 3, 1, 1, 1), (1, 32, 3, 1, 1, 1), (1, 64, 7, 2, 3, 0), (5, 64, 7, 2, 3, 0),
                                  (2, 128, 7, 1, 3, 0)])
def test_conv_gather_small_cin(gpu, case):
    """Planar small-Cin inputs (encoder stems, We / Wi, convf1): the per-element gather of conv_igemm_kernel (what the launcher
    picks, and explicit tile 2) against F.conv2d -- ragged sizes, one tile and
    many, fused ReLU."""
    Cin, Cout, K, stride, pad, pad_mode = case
    g = torch.Generator().manual_seed(11 + Cin + K)
    for (B, H, W, epi) in ((2, 22, 30, 0), (3, 37, 53, 1), (1, 9, 16, 0), (2, 64, 80, 0)):
        if pad_mode == 1 and min(H, W) <= pad:
            continue
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
        b = torch.randn(Cout, generator=g)
        ref = ref_conv(x, w, b, stride, pad, pad, pad_mode)
        if epi == 1:
            ref = torch.relu(ref)
        for tile in (0, 2):
            got = run_conv(gpu, x, w, b, stride, pad, pad, pad_mode, 2, epi, tile, ref.shape[2], ref.shape[3])
            assert got.shape == ref.shape and (got - ref).abs().max().item() < 1e-4, (B, H, W, tile)


def test_conv_fused_upsample(gpu):
    """UpsampleConvLayer (e2v/base_layers.py:195-212): interpolate x2 (align_corners=False) ->
    ReflectionPad2d(1) -> conv 3x3 (padding 0) -> relu, with the upsample fused into the A read."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 11, 15, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    up = F.interpolate(x, size=[22, 30], mode="bilinear", align_corners=False)
    ref = torch.relu(F.conv2d(F.pad(up, (1, 1, 1, 1), mode="reflect"), w, b))
    got = run_conv(gpu, x, w, b, 1, 1, 1, 1, 1, 1, 0, 22, 30)
    assert (got - ref).abs().max().item() < 1e-4


def ref_warp(img, flow, backward):
    """utils/flow_utils.py:83-120 / 153-190 restated with F.grid_sample."""
    B, C, H, W = img.shape
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    u, v = flow[:, 0], flow[:, 1]
    x = gx[None].float() + u if backward else gx[None].float() - u
    y = gy[None].float() + v if backward else gy[None].float() - v
    x = 2 * (x / W - 0.5)
    y = 2 * (y / H - 0.5)
    grid = torch.stack((x, y), dim=3)
    return F.grid_sample(img, grid, align_corners=True, padding_mode="reflection")


@pytest.mark.parametrize("C_,backward", [(1, 0), (1, 1), (128, 0), (6, 0)])
def test_warp(gpu, C_, backward):
    from cista_flow_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(3 + C_)
    B, H, W = 2, 18, 26
    img = torch.randn(B, C_, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 6.0   # big enough to hit the reflection branch
    flow[0, :, :3] = 0.0
    ref = ref_warp(img, flow, backward)
    h = lib.Handle(lib.CF_MODE_CISTA, B, 2 * H, 2 * W)
    xin = nhwc(img).to(gpu)
    fl = flow.to(gpu)
    out = torch.full_like(xin, float("nan"))
    h.check(L.cf_warp(h.h, lib.ptr(xin), lib.ptr(fl), lib.ptr(out), B, C_, H, W, H, W, backward,
                      lib.current_stream_ptr()), "cf_warp")
    torch.cuda.synchronize()
    got = nchw(out.cpu())
    assert (got - ref).abs().max().item() < 2e-5


def test_warp_with_downsampled_flow(gpu):
    """states[1] warp: interpolate(flow, 0.5, bilinear, align_corners=True) without halving values
    (e2v/e2v_model.py:190-191)."""
    from cista_flow_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(17)
    B, H, W, C_ = 2, 20, 28, 128
    z = torch.randn(B, C_, H // 2, W // 2, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 3.0
    dflow = F.interpolate(flow, scale_factor=0.5, mode="bilinear", align_corners=True)
    ref = ref_warp(z, dflow, 0)
    h = lib.Handle(lib.CF_MODE_CISTA, B, H, W)
    xin = nhwc(z).to(gpu)
    fl = flow.to(gpu)
    out = torch.full_like(xin, float("nan"))
    h.check(L.cf_warp(h.h, lib.ptr(xin), lib.ptr(fl), lib.ptr(out), B, C_, H // 2, W // 2, H, W, 0,
                      lib.current_stream_ptr()), "cf_warp")
    torch.cuda.synchronize()
    # tolerance: ~1e-5 px of coordinate rounding (two chained interpolations) x O(1) image gradient
    assert (nchw(out.cpu()) - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("C_", [64, 96, 128])
def test_instance_norm_relu(gpu, C_):
    L, lib = _lib()
    g = torch.Generator().manual_seed(C_)
    B, H, W = 2, 24, 40
    x = torch.randn(B, C_, H, W, generator=g) * 3.0 + 1.5
    ref = torch.relu(F.instance_norm(x, eps=1e-5))
    xin = nhwc(x).to(gpu)
    out = torch.full_like(xin, float("nan"))
    assert L.cf_op_instance_norm_relu(lib.ptr(xin), lib.ptr(out), B, C_, H, W, 1e-5, lib.current_stream_ptr()) == 0
    torch.cuda.synchronize()
    assert (nchw(out.cpu()) - ref).abs().max().item() < 2e-5


def ref_corr_lookup(fmap1, fmap2, coords, radius=4, levels=4):
    """DCEIFlow/core/corr/raft_corr.py:15-65 + DCEIFlow/utils/sample_utils.py:38-52 restated."""
    B, D, h, w = fmap1.shape
    corr = torch.matmul(fmap1.view(B, D, h * w).transpose(1, 2), fmap2.view(B, D, h * w))
    corr = corr.view(B * h * w, 1, h, w) / torch.sqrt(torch.tensor(D).float())
    pyr = [corr]
    for _ in range(levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    r = radius
    c = coords.permute(0, 2, 3, 1)
    outs = []
    for i in range(levels):
        cr = pyr[i]
        dx = torch.linspace(-r, r, 2 * r + 1)
        dy = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(dy, dx, indexing="ij"), axis=-1)
        cl = c.reshape(B * h * w, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        Hh, Ww = cr.shape[-2:]
        xg, yg = cl.split([1, 1], dim=-1)
        xg = 2 * xg / (Ww - 1) - 1
        yg = 2 * yg / (Hh - 1) - 1
        s = F.grid_sample(cr, torch.cat([xg, yg], dim=-1), align_corners=True)
        outs.append(s.view(B, h, w, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous()


def test_corr_lookup(gpu):
    L, lib = _lib()
    g = torch.Generator().manual_seed(23)
    B, D, h, w = 2, 64, 16, 24
    f1 = torch.randn(B, D, h, w, generator=g)
    f2 = torch.randn(B, D, h, w, generator=g)
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = torch.stack([gx, gy], 0)[None].float().repeat(B, 1, 1, 1) + torch.randn(B, 2, h, w, generator=g) * 2.5
    ref = ref_corr_lookup(f1, f2, coords)
    out = torch.full((B, h, w, 324), float("nan"), device=gpu)
    f1g, f2g, cg = nhwc(f1).to(gpu), nhwc(f2).to(gpu), coords.to(gpu)   # keep alive across the async call
    rc = L.cf_op_corr_lookup(lib.ptr(f1g), lib.ptr(f2g), lib.ptr(cg), lib.ptr(out), B, D, h, w, lib.current_stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    err = (nchw(out.cpu()) - ref).abs().max().item()
    assert err < 2e-4, err


def test_layout_roundtrip(gpu):
    L, lib = _lib()
    x = torch.randn(3, 37, 5, 11)
    xg = x.to(gpu)
    y = torch.empty(3, 5, 11, 37, device=gpu)
    assert L.cf_op_nchw_to_nhwc(lib.ptr(xg), lib.ptr(y), 3, 37, 5, 11, lib.current_stream_ptr()) == 0
    z = torch.empty_like(xg)
    assert L.cf_op_nhwc_to_nchw(lib.ptr(y), lib.ptr(z), 3, 37, 5, 11, lib.current_stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), nhwc(x))
    assert torch.equal(z.cpu(), x)


def test_events_to_voxel_gpu(gpu):
    """f-1: GPU scatter + normalisation vs the reference's numpy functions (golden) for a ragged batch."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as gu
    from cista_flow_amd.utils.event_process import events_to_voxel_grid_batch
    g = gu.load("events.npz")
    for i in (0, 1, 3):
        H, W = [int(v) for v in g["dims_%d" % i]]
        ev = torch.from_numpy(g["ev_%d" % i]).to(gpu)
        raw = events_to_voxel_grid_batch([ev], 5, W, H, normalize=False)[0].cpu()
        nrm = events_to_voxel_grid_batch([ev], 5, W, H, normalize=True)[0].cpu()
        assert gu.rel_err(raw, g["raw_%d" % i]) < 1e-5, i
        assert gu.rel_err(nrm, g["norm_%d" % i]) < 1e-5, i
    # ragged batch incl. an empty sequence: each entry must equal its single-sequence result
    evs = [torch.from_numpy(g["ev_0"]).to(gpu), torch.from_numpy(g["ev_2"]).to(gpu).reshape(0, 4), torch.from_numpy(g["ev_0"][:100]).to(gpu)]
    H, W = [int(v) for v in g["dims_0"]]
    out = events_to_voxel_grid_batch(evs, 5, W, H).cpu()
    assert gu.rel_err(out[0], g["norm_0"]) < 1e-5
    assert out[1].abs().max() == 0
    one = events_to_voxel_grid_batch([evs[2]], 5, W, H).cpu()
    assert gu.rel_err(out[2], one[0]) < 1e-6


@pytest.mark.parametrize("prec,tol", [(3, 2e-5), (1, 4e-3)])
@pytest.mark.parametrize("case", [(192, 256, 3, 1, 1, 0), (128, 64, 3, 1, 1, 4), (384, 128, 3, 1, 1, 9), (64, 64, 3, 2, 1, 2),
                                  (128, 128, 3, 1, 0, 8), (128, 256, 3, 1, 0, 10), (256, 128, 3, 1, 0, 11), (64, 128, 3, 1, 1, 13)])
def test_conv_f16_split_modes(gpu, case, prec, tol):
    """precision 3 (f16x3 split MFMA) must stay fp32-grade; precision 1 (plain f16 products) is the reduced mode."""
    L, lib = _lib()
    Cin, Cout, K, stride, pad_mode, tile = case
    g = torch.Generator().manual_seed(Cin + Cout + tile)
    B, H, W = 2, 20, 28
    x = torch.randn(B, Cin, H, W, generator=g) * 3.0
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = ref_conv(x.double(), w.double(), b.double(), stride, 1, 1, pad_mode).float()
    xin, wg, bg = nhwc(x).to(gpu), w.to(gpu), b.to(gpu)
    out = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), device=gpu)
    ms = C.c_float(0)
    rc = L.cf_op_conv2d_bench(lib.ptr(xin), B, Cin, H, W, lib.ptr(wg), lib.ptr(bg), Cout, K, K, stride, 1, 1, pad_mode, 0, 0,
                              tile, lib.ptr(out), lib.current_stream_ptr(), 1, C.byref(ms), prec)
    assert rc == 0
    torch.cuda.synchronize()
    err = ((nchw(out.cpu()) - ref).abs().max() / ref.abs().max()).item()
    assert err < tol, err


def test_quantize_u8_matches_numpy(gpu):
    """f-2: np.uint8(pred * 255.) (test_with_flow.py:174) -- bit-exact, including the k/255 boundaries, 0 and 1."""
    import numpy as np
    from cista_flow_amd.utils.image_process import to_uint8
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 1, 37, 53, generator=g)
    k = torch.arange(256, dtype=torch.float32)
    edge = torch.cat([k / 255.0, torch.nextafter(k / 255.0, torch.tensor(2.0)).clamp(max=1.0),
                      torch.nextafter(k / 255.0, torch.tensor(-1.0)).clamp(min=0.0), torch.tensor([0.0, 1.0, 0.5, 0.999999])])
    for t in (x, edge.reshape(1, 1, 1, -1), torch.sigmoid(torch.randn(2, 1, 180, 240, generator=g) * 6)):
        want = np.uint8(t.numpy() * 255.)
        got = to_uint8(t.to(gpu)).cpu().numpy()
        assert got.dtype == np.uint8 and got.shape == want.shape
        assert (got == want).all()
