// pointwise.hip -- the HBM-bound kernels of the CISTA-Flow hot path (gfx950).
//
// Everything here moves each byte once: flow warp (bilinear gather), instance-norm
// statistics / apply, ConvLSTM cell, correlation pyramid + lookup, flow up-sampling and the
// NCHW<->NHWC boundary shuffles.  NHWC tensors are addressed as  base + b*bs + pixel*ld + c.
#include "cf_device.h"
#include "cf_kernels.h"

#include <cstring>

namespace cf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// a4: flow warp.  utils/flow_utils.py:153-190 (forward: x - u) / :83-120 (backward: x + u); tap geometry in
// cf_device.h::warp_taps.  HBM-bound: 4*(2C+2) bytes per pixel (every source row is read ~once through L2: each is a
// tap of ~4 neighbouring output pixels), so the kernel is built around bytes in flight and L2 locality:
//   * vector path (C % 4 == 0, C/4 a power of two <= 64): a workgroup owns 64..256 consecutive pixels.  Phase 1: one
//     thread per pixel resamples the flow, reflects / clips and leaves the four tap offsets + weights in LDS (done
//     once per pixel instead of once per 16-byte channel quad).  Phase 2: C/4 adjacent lanes own a pixel row (one
//     512-byte row per half-wave at C = 128), read the taps as LDS broadcasts and keep 16 16-byte gathers in flight
//     per lane before blending -- no index arithmetic in the loop;
//   * workgroup -> pixel-strip map is XCD-aware (consecutive block ids are dealt round-robin over the 8 XCDs; each XCD
//     gets one contiguous band of strips): vertically adjacent output rows share source rows, and without this every
//     XCD's L2 fetched them again (PMC, round 1: 2.3x the algorithmic read bytes);
//   * scalar path for any other C (C = 1: the image warp), one thread per (pixel, channel);
//   * two tensors (image at full resolution + sparse code at half resolution, e2v_model.py:186-191) go through ONE
//     launch: blocks [0, nblk_a) serve descriptor a, the rest descriptor b.
// flag (nullable): device int; *flag == 0 (`flow_final.any()` false) turns the warp into a copy.
// ---------------------------------------------------------------------------
struct WarpDesc {
    const float* img; int ld; long bs;
    float* out; int out_ld; long out_bs;
    int C, H, W;
    int vec;       // 1: vector path
    int nblk;      // workgroups serving this tensor
};

__device__ __forceinline__ int xcd_band(int bid, int n) {      // block id -> logical strip (contiguous per XCD)
    const int q8 = n >> 3, r8 = n & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

struct WarpSlot {            // 64 bytes per pixel in LDS
    long o[4];               // element offsets of the four taps (image base included); o[0] < 0: pixel past the end
    long oo;                 // element offset of the output row
    float w[4];
    int pad[2];
};

__device__ __forceinline__ void warp_vec_path(const WarpDesc& d, const float* __restrict__ flow, int Hf, int Wf, int B,
                                              int backward, bool pass, int bid, WarpSlot* slots) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lpp = d.C >> 2;                               // lanes per pixel (power of two)
    const int lsh = __builtin_ctz(lpp);
    const int ppi = 64 >> lsh;                              // pixels per wave pass
    const int ppw = ppi * 32 < 256 ? ppi * 32 : 256;        // pixels per workgroup
    const int iters = ppw / (4 * ppi);                      // passes per wave (8 for C >= 32)
    const int HW = d.H * d.W;
    const long total = (long)B * HW;
    const long base = (long)xcd_band(bid, d.nblk) * ppw;
    if (tid < ppw) {
        const long gp = base + tid;
        WarpSlot s;
        s.o[0] = -1; s.o[1] = s.o[2] = s.o[3] = 0; s.oo = 0;
        s.w[0] = s.w[1] = s.w[2] = s.w[3] = 0.f;
        if (gp < total) {
            const int b = (int)(gp / HW);
            const int p = (int)(gp - (long)b * HW);
            const int y = p / d.W, x = p - y * d.W;
            const long ib = (long)b * d.bs;
            s.oo = (long)b * d.out_bs + (long)p * d.out_ld;
            if (pass) {
                s.o[0] = ib + (long)p * d.ld;
            } else {
                const float* f = flow + (long)b * 2 * Hf * Wf;
                const float u = flow_at(f, y, x, d.H, d.W, Hf, Wf);
                const float v = flow_at(f + (long)Hf * Wf, y, x, d.H, d.W, Hf, Wf);
                const WarpTaps t = warp_taps(u, v, x, y, d.H, d.W, backward);
                s.o[0] = ib + (long)t.p00 * d.ld; s.o[1] = ib + (long)t.p01 * d.ld;
                s.o[2] = ib + (long)t.p10 * d.ld; s.o[3] = ib + (long)t.p11 * d.ld;
                s.w[0] = t.w00; s.w[1] = t.w01; s.w[2] = t.w10; s.w[3] = t.w11;
            }
        }
        slots[tid] = s;
    }
    __syncthreads();
    const int q4 = (lane & (lpp - 1)) * 4;
    const int sub = lane >> lsh;
    constexpr int BATCH = 4;
    // The loads of one batch are in flight together.  Neighbouring output pixels share source rows (the right tap of
    // pixel x is the left tap of x + 1); requested together, both requests miss and BOTH go out to the fabric (PMC, r02:
    // 93 MB fetched for 47 MB of algorithmic reads).  So a batch takes every nb-th pixel of the wave's strip and the
    // pixels in between come with the next batch, when the shared rows are L2 hits.
    const int nb = (iters + BATCH - 1) / BATCH;
    auto slot_of = [&](int it) { return wave * iters * ppi + nb * ((it % BATCH) * ppi + sub) + it / BATCH; };
    for (int it0 = 0; it0 < iters; it0 += BATCH) {
        f32x4 v[BATCH][4];
        bool ok[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const bool live = it0 + j < iters;
            const WarpSlot& s = slots[live ? slot_of(it0 + j) : 0];     // live: index < ppw <= 256
            ok[j] = live && s.o[0] >= 0;
            if (ok[j]) {
                v[j][0] = *reinterpret_cast<const f32x4*>(d.img + s.o[0] + q4);
                if (!pass) {
                    v[j][1] = *reinterpret_cast<const f32x4*>(d.img + s.o[1] + q4);
                    v[j][2] = *reinterpret_cast<const f32x4*>(d.img + s.o[2] + q4);
                    v[j][3] = *reinterpret_cast<const f32x4*>(d.img + s.o[3] + q4);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (!ok[j]) continue;
            const WarpSlot& s = slots[slot_of(it0 + j)];
            f32x4 r = v[j][0];
            if (!pass) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    r[e] = v[j][0][e] * s.w[0] + v[j][1][e] * s.w[1] + v[j][2][e] * s.w[2] + v[j][3][e] * s.w[3];
            }
            *reinterpret_cast<f32x4*>(d.out + s.oo + q4) = r;
        }
    }
}

__device__ __forceinline__ void warp_scalar_path(const WarpDesc& d, const float* __restrict__ flow, int Hf, int Wf, int B,
                                                 int backward, bool pass, int bid) {
    // blocks per image = ceil(HW*C / 256); one thread per (pixel, channel), channel fastest
    const unsigned per = (unsigned)d.H * d.W * d.C;
    const unsigned bpi = (per + 255u) / 256u;
    const int b = (int)((unsigned)bid / bpi);
    const unsigned idx = ((unsigned)bid - (unsigned)b * bpi) * 256u + threadIdx.x;
    if (b >= B || idx >= per) return;
    const int p = (int)(idx / (unsigned)d.C);
    const int c = (int)(idx - (unsigned)p * d.C);
    const float* ib = d.img + (long)b * d.bs + c;
    float* o = d.out + (long)b * d.out_bs + (long)p * d.out_ld + c;
    if (pass) {
        *o = ib[(long)p * d.ld];
        return;
    }
    const int y = p / d.W, x = p - y * d.W;
    const float* f = flow + (long)b * 2 * Hf * Wf;
    const float u = flow_at(f, y, x, d.H, d.W, Hf, Wf);
    const float v = flow_at(f + (long)Hf * Wf, y, x, d.H, d.W, Hf, Wf);
    const WarpTaps t = warp_taps(u, v, x, y, d.H, d.W, backward);
    *o = ib[(long)t.p00 * d.ld] * t.w00 + ib[(long)t.p01 * d.ld] * t.w01 + ib[(long)t.p10 * d.ld] * t.w10 +
         ib[(long)t.p11 * d.ld] * t.w11;
}

__global__ __launch_bounds__(256) void warp_kernel(const WarpDesc a, const WarpDesc b2, const float* __restrict__ flow, int Hf,
                                                   int Wf, int B, int backward, const int* flag) {
    __shared__ WarpSlot slots[256];
    const bool pass = flag && (*flag == 0);
    const int bid = blockIdx.x;
    if (bid < a.nblk) {
        if (a.vec) warp_vec_path(a, flow, Hf, Wf, B, backward, pass, bid, slots);
        else warp_scalar_path(a, flow, Hf, Wf, B, backward, pass, bid);
    } else {
        if (b2.vec) warp_vec_path(b2, flow, Hf, Wf, B, backward, pass, bid - a.nblk, slots);
        else warp_scalar_path(b2, flow, Hf, Wf, B, backward, pass, bid - a.nblk);
    }
}

static bool warp_desc(WarpDesc& d, const float* img, int img_ld, long img_bs, float* out, int out_ld, long out_bs, int B, int C,
                      int H, int W) {
    if (!img || !out || C <= 0 || H <= 0 || W <= 0 || img_ld < C || out_ld < C) return false;
    d.img = img; d.ld = img_ld; d.bs = img_bs; d.out = out; d.out_ld = out_ld; d.out_bs = out_bs; d.C = C; d.H = H; d.W = W;
    const int lpp = C / 4;
    d.vec = ((C % 4) == 0 && lpp <= 64 && (lpp & (lpp - 1)) == 0 && (img_ld % 4) == 0 && (out_ld % 4) == 0 && (img_bs % 4) == 0 &&
             (out_bs % 4) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
                ? 1 : 0;
    const long total = (long)B * H * W;
    if (d.vec) {
        const int ppi = 64 / lpp;
        const int ppw = ppi * 32 < 256 ? ppi * 32 : 256;
        d.nblk = (int)((total + ppw - 1) / ppw);
    } else {
        const long per = (long)H * W * C;
        if (per >= 0x7FFFFFFFL) return false;
        d.nblk = (int)(((per + 255) / 256) * B);
    }
    return total * C < (1L << 40);
}

// one or two tensors warped by the same flow in one launch (img2 == nullptr: one)
hipError_t launch_warp2(const float* img, int img_ld, long img_bs, float* out, int out_ld, long out_bs, int C, int H, int W,
                        const float* img2, int img2_ld, long img2_bs, float* out2, int out2_ld, long out2_bs, int C2, int H2,
                        int W2, const float* flow, int Hf, int Wf, int B, int backward, const int* flag, hipStream_t s) {
    if (!flow || B <= 0 || Hf <= 0 || Wf <= 0) return hipErrorInvalidValue;
    WarpDesc a, b;
    memset(&a, 0, sizeof(a));
    memset(&b, 0, sizeof(b));
    if (!warp_desc(a, img, img_ld, img_bs, out, out_ld, out_bs, B, C, H, W)) return hipErrorInvalidValue;
    if (img2 && !warp_desc(b, img2, img2_ld, img2_bs, out2, out2_ld, out2_bs, B, C2, H2, W2)) return hipErrorInvalidValue;
    if (b.vec && !a.vec) {       // the XCD band map keys on blockIdx & 7: the vector tensor's blocks must start at 0
        const WarpDesc t = a;
        a = b;
        b = t;
    }
    const long blocks = (long)a.nblk + b.nblk;
    if (blocks <= 0 || blocks >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    note_launch("warp_kernel", dim3((unsigned)blocks), dim3(256));
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, flow, Hf, Wf, B, backward, flag);
    return hipGetLastError();
}

hipError_t launch_warp(const float* img, int img_ld, long img_bs, const float* flow, int Hf, int Wf, float* out,
                       int out_ld, long out_bs, int B, int C, int H, int W, int backward, const int* flag,
                       hipStream_t s) {
    return launch_warp2(img, img_ld, img_bs, out, out_ld, out_bs, C, H, W, nullptr, 0, 0, nullptr, 0, 0, 0, 0, 0, flow, Hf, Wf,
                        B, backward, flag, s);
}

// flag |= any(x != 0)   (NaN != 0 is true, like torch.Tensor.any on floats)
__global__ void any_nonzero_kernel(const float* __restrict__ x, long n, int* flag) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    bool nz = false;
    for (; i < n; i += stride) nz = nz || (x[i] != 0.0f);
    if (__any(nz)) {
        if ((threadIdx.x & 63) == 0) *flag = 1;
    }
}

hipError_t launch_any_nonzero(const float* x, long n, int* flag, hipStream_t s) {
    if (!x || !flag || n <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    note_launch("any_nonzero_kernel", dim3((unsigned)blocks), dim3(256));
    hipLaunchKernelGGL(any_nonzero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// InstanceNorm2d (affine=False, biased variance, eps) -- raft_encoder.py:32-36,136-137.
// stage 1: per (image, pixel-chunk) partial {sum, sum of squares} in fp64 for every channel
// stage 2: fold the chunks -> {mean, rstd}
// ---------------------------------------------------------------------------
static constexpr int IN_CHUNK = 256;   // pixels per stage-1 workgroup

// one thread = 4 consecutive channels (16-byte loads); the C/4 channel quads of a pixel sit in adjacent
// lanes so every wave load covers whole 128-byte rows; partial sums are kept in fp64
__global__ __launch_bounds__(256) void inorm_partial_kernel(const float* __restrict__ x, int ld, long bs, int HW, int C,
                                                            double* __restrict__ partial, int nchunk) {
    __shared__ double sh[256 * 8];
    const int b = blockIdx.y, ch = blockIdx.x;
    const int tid = threadIdx.x;
    const int cq = C >> 2;              // channel quads per pixel (16, 24, 32)
    const int ppp = 256 / cq;           // pixels per pass
    const int pl = tid / cq;
    const int q = tid - pl * cq;
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (pl < ppp) {
        const int p0 = ch * IN_CHUNK;
        const int p1 = min(HW, p0 + IN_CHUNK);
        const float* xb = x + (long)b * bs + q * 4;
        for (int p = p0 + pl; p < p1; p += ppp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)p * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double d = (double)v[e];
                s[e] += d;
                ss[e] += d * d;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[tid * 8 + e] = s[e];
        sh[tid * 8 + 4 + e] = ss[e];
    }
    __syncthreads();
    if (tid < C) {
        const int qq = tid >> 2, e = tid & 3;
        double a = 0.0, aa = 0.0;
        for (int k = 0; k < ppp; ++k) {
            a += sh[(k * cq + qq) * 8 + e];
            aa += sh[(k * cq + qq) * 8 + 4 + e];
        }
        double* dst = partial + (((long)b * nchunk + ch) * C + tid) * 2;
        dst[0] = a;
        dst[1] = aa;
    }
}

// one workgroup = (image, 4 channels): 64 thread rows walk the chunks in parallel with all of a thread's 16-byte
// loads in flight at once (the partials were just written by other XCDs, so every load is a fabric round trip and
// the fold is latency-bound); thread row 0 then adds the 64 row sums in a fixed order -- deterministic
__global__ __launch_bounds__(256) void inorm_final_kernel(const double* __restrict__ partial, int nchunk, int C, int HW,
                                                          float eps, float* __restrict__ stats) {
    __shared__ double sh[64][4][2];
    const int b = blockIdx.y;
    const int cl = threadIdx.x & 3, k = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + cl;
    double s = 0.0, ss = 0.0;
    if (c < C) {
        const double* src = partial + ((long)b * nchunk * C + c) * 2;
        int i = k;
        for (; i + 7 * 64 < nchunk; i += 8 * 64) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(src + (long)(i + u * 64) * C * 2);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += v[u].x;
                ss += v[u].y;
            }
        }
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u].x = 0.0;
            v[u].y = 0.0;
            if (i + u * 64 < nchunk) v[u] = *reinterpret_cast<const double2*>(src + (long)(i + u * 64) * C * 2);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s += v[u].x;
            ss += v[u].y;
        }
    }
    sh[k][cl][0] = s;
    sh[k][cl][1] = ss;
    __syncthreads();
    if (k == 0 && c < C) {
        double a = 0.0, aa = 0.0;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            a += sh[r][cl][0];
            aa += sh[r][cl][1];
        }
        const double mean = a / (double)HW;
        double var = aa / (double)HW - mean * mean;
        if (var < 0.0) var = 0.0;
        float* dst = stats + ((long)b * C + c) * 2;
        dst[0] = (float)mean;
        dst[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

hipError_t launch_inorm_final(const double* partial, int nchunk, int B, int HW, int C, float eps, float* stats,
                              hipStream_t s) {
    if (!partial || !stats || nchunk <= 0 || B <= 0 || HW <= 0 || C <= 0) return hipErrorInvalidValue;
    note_launch("inorm_final_kernel", dim3((C + 3) / 4, B), dim3(256));
    hipLaunchKernelGGL(inorm_final_kernel, dim3((C + 3) / 4, B), dim3(256), 0, s, partial, nchunk, C, HW, eps, stats);
    return hipGetLastError();
}

hipError_t launch_inorm_stats(const float* x, int ld, long bs, int B, int HW, int C, float eps, double* partial,
                              float* stats, hipStream_t s) {
    if (!x || !partial || !stats || C <= 0 || C > 256 || (C % 4) != 0 || (ld % 4) != 0 || (bs % 4) != 0 || B <= 0 || HW <= 0 || ld < C ||
        (reinterpret_cast<uintptr_t>(x) & 15) != 0)
        return hipErrorInvalidValue;
    const int nchunk = (HW + IN_CHUNK - 1) / IN_CHUNK;
    note_launch("inorm_partial_kernel", dim3(nchunk, B), dim3(256));
    hipLaunchKernelGGL(inorm_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, ld, bs, HW, C, partial, nchunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_inorm_final(partial, nchunk, B, HW, C, eps, stats, s);
}

// number of doubles launch_inorm_stats needs in `partial`
long inorm_partial_doubles(int B, int HW, int C) { return (long)B * ((HW + IN_CHUNK - 1) / IN_CHUNK) * C * 2; }
// ... and a convolution with fused statistics (ConvParams::st_partial, 32-pixel patches)
// per image max(32-pixel patches, 4 partials per Winograd region in either orientation of the 8 x 16 regions, 16 per F(4x4,3x3) region): on small ragged
// maps the regions outnumber the patches (17 x 17: 24 partials against 10), so the bound is taken from (Ho, Wo), not from Ho*Wo
long inorm_patch_doubles(int B, int Ho, int Wo, int C) {
    const long patches = ((long)Ho * Wo + 31) / 32;
    const long wide = 4L * ((Ho + 7) / 8) * ((Wo + 15) / 16), tall = 4L * ((Ho + 15) / 16) * ((Wo + 7) / 8);
    const long w4 = 16L * ((Ho + 15) / 16) * ((Wo + 31) / 32);          // conv_wino4_kernel: 16 partials per 16 x 32 region
    long chunks = patches > wide ? patches : wide;                      // (`wide` is also conv_patch_kernel's count: 4 per 8 x 16 tile)
    chunks = chunks > tall ? chunks : tall;
    chunks = chunks > w4 ? chunks : w4;
    return (long)B * chunks * C * 2 + 1024;
}

__global__ __launch_bounds__(256) void inorm_apply_kernel(const float* __restrict__ x, int ld, long bs,
                                                          const float* __restrict__ stats, const float* __restrict__ res,
                                                          int res_ld, long res_bs, const float* __restrict__ res_stats,
                                                          float* __restrict__ out, int out_ld, long out_bs, int B, int HW,
                                                          int C) {
    // grid: x over the quads of one image (HW * C/4 < 2^31), y = image -- 32-bit index math only
    const int cq = C / 4;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (unsigned)HW * (unsigned)cq) return;
    const int p = (int)(gid / (unsigned)cq);
    const int c0 = (int)(gid - (unsigned)p * (unsigned)cq) * 4;
    const int b = blockIdx.y;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (long)b * bs + (long)p * ld + c0);
    f32x4 r;
    const float* st = stats + ((long)b * C + c0) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf((v[e] - st[2 * e]) * st[2 * e + 1], 0.f);
    if (res) {
        f32x4 rv = *reinterpret_cast<const f32x4*>(res + (long)b * res_bs + (long)p * res_ld + c0);
        if (res_stats) {
            const float* rs = res_stats + ((long)b * C + c0) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) rv[e] = (rv[e] - rs[2 * e]) * rs[2 * e + 1];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(rv[e] + r[e], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (long)b * out_bs + (long)p * out_ld + c0) = r;
}

hipError_t launch_inorm_apply(const float* x, int ld, long bs, const float* stats, const float* res, int res_ld,
                              long res_bs, const float* res_stats, float* out, int out_ld, long out_bs, int B, int HW,
                              int C, hipStream_t s) {
    if (!x || !stats || !out || (C % 4) != 0 || (ld % 4) != 0 || (out_ld % 4) != 0 || (bs % 4) != 0 || (out_bs % 4) != 0)
        return hipErrorInvalidValue;
    if (res && ((res_ld % 4) != 0 || (res_bs % 4) != 0)) return hipErrorInvalidValue;
    const long per_image = (long)HW * (C / 4);
    if (per_image >= 0x7FFFFFFFL || B > 65535) return hipErrorInvalidValue;
    note_launch("inorm_apply_kernel", dim3((unsigned)((per_image + 255) / 256), B), dim3(256));
    hipLaunchKernelGGL(inorm_apply_kernel, dim3((unsigned)((per_image + 255) / 256), B), dim3(256), 0, s, x, ld, bs, stats, res,
                       res_ld, res_bs, res_stats, out, out_ld, out_bs, B, HW, C);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// correlation pyramid: F.avg_pool2d(corr, 2, stride=2) over the (h2,w2) plane of every
// (b, i) row -- DCEIFlow/core/corr/raft_corr.py:28-30.
// ---------------------------------------------------------------------------
__global__ void corr_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int Hs, int Ws) {
    const int Hd = Hs / 2, Wd = Ws / 2;
    const long total = rows * Hd * Wd;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % Wd);
    const int y = (int)((gid / Wd) % Hd);
    const long r = gid / ((long)Wd * Hd);
    const float* s = src + r * Hs * Ws + (long)(2 * y) * Ws + 2 * x;
    const float sum = ((s[0] + s[1]) + s[Ws]) + s[Ws + 1];
    dst[gid] = sum / 4.f;
}

hipError_t launch_corr_pool(const float* src, float* dst, long rows, int Hs, int Ws, hipStream_t s) {
    if (!src || !dst || rows <= 0 || Hs < 2 || Ws < 2) return hipErrorInvalidValue;
    const long total = rows * (Hs / 2) * (Ws / 2);
    note_launch("corr_pool_kernel", dim3((unsigned)((total + 255) / 256)), dim3(256));
    hipLaunchKernelGGL(corr_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, rows, Hs, Ws);
    return hipGetLastError();
}

// The three pooled levels in ONE launch (the cascade is three ~6.5 us launches of almost no work on the step's critical path): a wave
// takes one (b, i) row, stages its level-0 map in LDS and pools it three times there -- every level is computed from the rounded
// values of the level above with corr_pool_kernel's expression, so the pyramid is bit-identical to the cascade.  The trailing
// blocks initialise coords1 (coords_init_kernel's job: one more tiny launch in front of the first lookup).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_pyramid_kernel(const float* __restrict__ l0, float* __restrict__ l1, float* __restrict__ l2,
                                                           float* __restrict__ l3, long rows, int H0, int W0, int nblk_pool,
                                                           float* coords1, const float* flow_init, int B, int h8, int w8, int* flag) {
    extern __shared__ float pyr_smem[];
    if ((int)blockIdx.x >= nblk_pool) {
        const long N = (long)h8 * w8;
        const long total = (long)B * 2 * N;
        const long gid = (long)(blockIdx.x - nblk_pool) * blockDim.x + threadIdx.x;
        if (gid == 0 && flag) *flag = 0;                   // the "any flow" flag of this frame (was a memset node of its own)
        if (gid >= total) return;
        const int i = (int)(gid % N);
        const int ch = (int)((gid / N) % 2);
        float v = ch == 0 ? (float)(i % w8) : (float)(i / w8);
        if (flow_init) v = v + flow_init[gid];
        coords1[gid] = v;
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + wave;
    if (r >= rows) return;                                   // no workgroup-wide barrier below
    const int H1 = H0 / 2, W1 = W0 / 2, H2 = H1 / 2, W2 = W1 / 2, H3 = H2 / 2, W3 = W2 / 2;
    const int n0 = H0 * W0, n1 = H1 * W1, n2 = H2 * W2, n3 = H3 * W3;
    float* s0 = pyr_smem + wave * (n0 + n1 + n2);
    float* s1 = s0 + n0;
    float* s2 = s1 + n1;
    const float* src = l0 + r * n0;
    for (int t = lane; t < n0; t += 64) s0[t] = src[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = lane; t < n1; t += 64) {
        const int y = t / W1, x = t - y * W1;
        const float* q = s0 + (2 * y) * W0 + 2 * x;
        const float v = (((q[0] + q[1]) + q[W0]) + q[W0 + 1]) / 4.f;
        s1[t] = v;
        l1[r * n1 + t] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = lane; t < n2; t += 64) {
        const int y = t / W2, x = t - y * W2;
        const float* q = s1 + (2 * y) * W1 + 2 * x;
        const float v = (((q[0] + q[1]) + q[W1]) + q[W1 + 1]) / 4.f;
        s2[t] = v;
        l2[r * n2 + t] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int t = lane; t < n3; t += 64) {
        const int y = t / W3, x = t - y * W3;
        const float* q = s2 + (2 * y) * W2 + 2 * x;
        l3[r * n3 + t] = (((q[0] + q[1]) + q[W2]) + q[W2 + 1]) / 4.f;
    }
}

// LDS one workgroup (four rows) needs; above the limit the caller keeps the three-launch cascade
long corr_pyramid_lds_bytes(int H0, int W0) {
    const long n0 = (long)H0 * W0, n1 = (long)(H0 / 2) * (W0 / 2), n2 = (long)(H0 / 4) * (W0 / 4);
    return 4 * (n0 + n1 + n2) * 4;
}

hipError_t launch_corr_pyramid(const float* l0, float* l1, float* l2, float* l3, long rows, int H0, int W0, float* coords1,
                               const float* flow_init, int B, int h8, int w8, int* flag, hipStream_t s) {
    if (!l0 || !l1 || !l2 || !l3 || rows <= 0 || H0 < 8 || W0 < 8) return hipErrorInvalidValue;
    if (flag && !coords1) return hipErrorInvalidValue;      // the flag is cleared by the trailing coords-init blocks: none without coords1
    const long lds = corr_pyramid_lds_bytes(H0, W0);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const long nb_pool = (rows + 3) / 4;
    const long nb_init = coords1 ? ((long)B * 2 * h8 * w8 + 255) / 256 : 0;
    if (nb_pool + nb_init >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    note_launch("corr_pyramid_kernel", dim3((unsigned)(nb_pool + nb_init)), dim3(256));
    hipLaunchKernelGGL(corr_pyramid_kernel, dim3((unsigned)(nb_pool + nb_init)), dim3(256), (size_t)lds, s, l0, l1, l2, l3, rows, H0, W0,
                       (int)nb_pool, coords1, flow_init, B, h8, w8, flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// correlation lookup (a10): raft_corr.py:32-54 + sample_utils.py:38-52, with FlowHead.conv2 and `coords1 += delta`
// of the PREVIOUS refinement iteration fused in front of it (with_event_updater.py:13-14, DCEIFlow.py:218): both are
// per-query-pixel work on the 1/8-resolution grid and sat on the iteration's critical path as three tiny launches.
//
// channel lvl*81 + a*9 + bb = zero-padded bilinear sample of level lvl at (x/2^lvl + a - 4, y/2^lvl + bb - 4).  The 81
// samples of a level differ by INTEGER offsets, so they share one pair of fractional weights and one 10x10 grid of
// integer taps: a wave (= one query pixel) loads the 100 taps of a level into its LDS patch (2 loads per lane) and every
// output is a 4-term blend of neighbouring patch entries -- 8 global loads per lane and query instead of 24, no
// per-channel coordinate arithmetic.
// FlowHead.conv2 (3x3, 256 -> 2, zero pad): lanes split the 256 channels (16-byte loads), 9 taps, wave reduction.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_lookup_kernel(const LookupParams p) {
    __shared__ float patch[4][4][104];          // [wave][level][10x10 taps]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int N = p.h8 * p.w8;
    const long qid = (long)blockIdx.x * 4 + wave;
    if (qid >= (long)p.B * N) return;
    const int b = (int)(qid / N);
    const int i = (int)(qid - (long)b * N);
    const int qy = i / p.w8, qx = i - qy * p.w8;
    float cx = p.coords1[((long)b * 2 + 0) * N + i];
    float cy = p.coords1[((long)b * 2 + 1) * N + i];
    if (p.fh) {
        // delta_flow = FlowHead.conv2(relu(conv1(net))) at this pixel; coords1 += delta_flow
        float a0 = 0.f, a1 = 0.f;
        const int c0 = lane * 4;                 // 64 lanes x 4 channels = 256
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = qy + t / 3 - 1, xx = qx + t % 3 - 1;
            if (yy < 0 || yy >= p.h8 || xx < 0 || xx >= p.w8) continue;      // wave-uniform
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.fh + ((long)b * N + (long)yy * p.w8 + xx) * p.fh_ld + c0);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.fh_w + (long)t * p.fh_ld + c0);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(p.fh_w + (long)p.fh_ktot + (long)t * p.fh_ld + c0);
            a0 += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
            a1 += v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a0 += __shfl_xor(a0, off);
            a1 += __shfl_xor(a1, off);
        }
        cx = (a0 + p.fh_bias[0]) + cx;           // conv output (+ bias) + aux, as the stand-alone conv tail orders it
        cy = (a1 + p.fh_bias[1]) + cy;
        if (lane == 0) {
            p.coords_out[((long)b * 2 + 0) * N + i] = cx;
            p.coords_out[((long)b * 2 + 1) * N + i] = cy;
        }
    }
    if (!p.out) return;
    // ---- taps: level l, lane -> tap (ty, tx) of the 10x10 grid, two passes ----
    float tx[4], ty[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l >= p.nlevels) break;
        const int Hl = p.lh[l], Wl = p.lw[l];
        const float* plane = p.lvl[l] + ((long)b * N + i) * Hl * Wl;
        const float div = (float)(1 << l);
        // bilinear_sampler: g = 2*x/(W-1) - 1 ; grid_sample(align_corners=True): (g+1)*((W-1)/2) -- evaluated for the
        // window's first sample (offset -4); the other 80 are at exact integer distances from it
        const float x0s = cx / div + (float)(-p.radius), y0s = cy / div + (float)(-p.radius);
        const float gx = 2.f * x0s / (float)(Wl - 1) - 1.f, gy = 2.f * y0s / (float)(Hl - 1) - 1.f;
        const float ix = (gx + 1.f) * ((float)(Wl - 1) / 2.f), iy = (gy + 1.f) * ((float)(Hl - 1) / 2.f);
        const float fx = floorf(ix), fy = floorf(iy);
        tx[l] = ix - fx;
        ty[l] = iy - fy;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int t = lane + 64 * k;
            if (t < 100) {
                const int tr = t / 10, tc = t - tr * 10;
                const float xf = fx + (float)tc, yf = fy + (float)tr;
                // float compare before the int cast keeps huge / NaN coordinates out of range
                const bool ok = xf >= 0.f && xf <= (float)(Wl - 1) && yf >= 0.f && yf <= (float)(Hl - 1);
                patch[wave][l][t] = ok ? plane[(long)(int)yf * Wl + (int)xf] : 0.f;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int d = 2 * p.radius + 1;          // 9
    float* o = p.out + ((long)b * N + i) * p.out_ld;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l >= p.nlevels) break;
        const float w00 = (1.f - tx[l]) * (1.f - ty[l]), w01 = tx[l] * (1.f - ty[l]), w10 = (1.f - tx[l]) * ty[l], w11 = tx[l] * ty[l];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = lane + 64 * k;
            if (ch < d * d) {
                const int a = ch / d, bb = ch - a * d;          // a steps x, bb steps y (RAFT's transposed window)
                const float* t0 = &patch[wave][l][bb * 10 + a];
                o[l * d * d + ch] = t0[0] * w00 + t0[1] * w01 + t0[10] * w10 + t0[11] * w11;
            }
        }
    }
    for (int ch = p.nlevels * d * d + lane; ch < p.out_ld; ch += 64) o[ch] = 0.f;      // pad channels of the concat buffer
    if (p.motion && lane < 2) {
        // flow = coords1 - coords0 ; coords0 = pixel grid (sample_utils.py:55-58)
        const float f = lane == 0 ? (cx - (float)qx) : (cy - (float)qy);
        p.motion[((long)b * N + i) * p.mo_ld + p.mo_off + lane] = f;
    }
}

hipError_t launch_corr_lookup(const LookupParams& p, hipStream_t s) {
    if (p.nlevels < 1 || p.nlevels > 4 || p.radius != 4 || !p.coords1) return hipErrorInvalidValue;
    const int d = 2 * p.radius + 1;
    if (p.out) {
        if (p.out_ld < p.nlevels * d * d) return hipErrorInvalidValue;
        for (int l = 0; l < p.nlevels; ++l)
            if (!p.lvl[l] || p.lh[l] < 2 || p.lw[l] < 2) return hipErrorInvalidValue;
    }
    if (p.fh) {
        if (!p.fh_w || !p.fh_bias || !p.coords_out || p.fh_ld != 256 || p.fh_ktot != 9 * 256 ||
            ((reinterpret_cast<uintptr_t>(p.fh) | reinterpret_cast<uintptr_t>(p.fh_w)) & 15) != 0)
            return hipErrorInvalidValue;
    } else if (!p.out) {
        return hipErrorInvalidValue;
    }
    const long nq = (long)p.B * p.h8 * p.w8;
    note_launch("corr_lookup_kernel", dim3((unsigned)((nq + 3) / 4)), dim3(256));
    hipLaunchKernelGGL(corr_lookup_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// coords1 = coords_grid (+ flow_init)   DCEIFlow.py:96-104,198-201
// ---------------------------------------------------------------------------
__global__ void coords_init_kernel(float* coords1, const float* flow_init, int B, int h8, int w8) {
    const long N = (long)h8 * w8;
    const long total = (long)B * 2 * N;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int i = (int)(gid % N);
    const int ch = (int)((gid / N) % 2);
    float v = ch == 0 ? (float)(i % w8) : (float)(i / w8);
    if (flow_init) v = v + flow_init[gid];
    coords1[gid] = v;
}

hipError_t launch_coords_init(float* coords1, const float* flow_init, int B, int h8, int w8, hipStream_t s) {
    if (!coords1 || B <= 0 || h8 <= 0 || w8 <= 0) return hipErrorInvalidValue;
    const long total = (long)B * 2 * h8 * w8;
    note_launch("coords_init_kernel", dim3((unsigned)((total + 255) / 256)), dim3(256));
    hipLaunchKernelGGL(coords_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, coords1, flow_init, B,
                       h8, w8);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// upflow8 (sample_utils.py:66-68): ds * interpolate(flow, x ds, bilinear, align_corners=True),
// then ImagePadder.unpad (image_process.py:103-107).  Also raises the ".any()" flag.
// ---------------------------------------------------------------------------
__global__ void upflow_kernel(const float* __restrict__ coords1, int B, int h8, int w8, int ds, float* flow_up,
                              float* flow_final, int H, int W, int padH, int padW, int* flag, float* flow_low, int nblk_main) {
    const int Hp = h8 * ds, Wp = w8 * ds;
    const long total = (long)B * 2 * Hp * Wp;
    if ((int)blockIdx.x >= nblk_main) {
        // trailing blocks: flow_low = coords1 - coords0 on the 1/8 grid (what a ds = 1 launch of this kernel computes: the bilinear
        // weights are exactly 1 and 0 there), folded into the last iteration's launch
        const long n = (long)h8 * w8, t = (long)(blockIdx.x - nblk_main) * blockDim.x + threadIdx.x;
        if (t < (long)B * 2 * n) {
            const int i = (int)(t % n), ch = (int)((t / n) % 2);
            const float g = ch == 0 ? (float)(i % w8) : (float)(i / w8);
            flow_low[t] = coords1[t] - g;            // = 1 * (1 * (1 * v + 0 * v') + 0 * (..)) of the general path for finite values
        }
        return;
    }
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool nz = false;
    if (gid < total) {
        const int x = (int)(gid % Wp);
        const int y = (int)((gid / Wp) % Hp);
        const int ch = (int)((gid / ((long)Wp * Hp)) % 2);
        const int b = (int)(gid / ((long)Wp * Hp * 2));
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        ac_true_src(y, h8, Hp, y0, y1, ly0, ly1);
        ac_true_src(x, w8, Wp, x0, x1, lx0, lx1);
        const float* c = coords1 + ((long)b * 2 + ch) * h8 * w8;
        // flow = coords1 - coords0 at the four taps
        const float g00 = ch == 0 ? (float)x0 : (float)y0;
        const float g01 = ch == 0 ? (float)x1 : (float)y0;
        const float g10 = ch == 0 ? (float)x0 : (float)y1;
        const float g11 = ch == 0 ? (float)x1 : (float)y1;
        const float v00 = c[(long)y0 * w8 + x0] - g00, v01 = c[(long)y0 * w8 + x1] - g01;
        const float v10 = c[(long)y1 * w8 + x0] - g10, v11 = c[(long)y1 * w8 + x1] - g11;
        const float v = (float)ds * (ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11));
        if (flow_up) flow_up[gid] = v;
        if (flow_final && y >= padH && x >= padW) {
            flow_final[(((long)b * 2 + ch) * H + (y - padH)) * W + (x - padW)] = v;
            nz = v != 0.0f;
        }
    }
    if (flag && __any(nz)) {
        if ((threadIdx.x & 63) == 0) *flag = 1;
    }
}

hipError_t launch_upflow(const float* coords1, int B, int h8, int w8, int ds, float* flow_up, float* flow_final, int H,
                         int W, int padH, int padW, int* flag, hipStream_t s, float* flow_low) {
    if (!coords1 || B <= 0 || h8 <= 0 || w8 <= 0 || ds <= 0) return hipErrorInvalidValue;
    if (flow_final && (H + padH != h8 * ds || W + padW != w8 * ds)) return hipErrorInvalidValue;
    const long total = (long)B * 2 * h8 * ds * w8 * ds;
    const long nb_main = (total + 255) / 256, nb_low = flow_low ? ((long)B * 2 * h8 * w8 + 255) / 256 : 0;
    if (nb_main + nb_low >= 0x7FFFFFFFL) return hipErrorInvalidValue;      // nb_main travels as an int, the grid is unsigned
    note_launch("upflow_kernel", dim3((unsigned)(nb_main + nb_low)), dim3(256));
    hipLaunchKernelGGL(upflow_kernel, dim3((unsigned)(nb_main + nb_low)), dim3(256), 0, s, coords1, B, h8, w8, ds,
                       flow_up, flow_final, H, W, padH, padW, flag, flow_low, (int)nb_main);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// learned convex up-sampling: one wave = one 1/8-res pixel, lane = (i, j) of its 8x8 output patch
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ coords1, int coords_is_flow,
                                                              const float* __restrict__ mask, int mask_ld, int B, int h8,
                                                              int w8, float* flow_up, float* flow_final, int H, int W,
                                                              int padH, int padW, int* flag, const float* add,
                                                              float* total_out) {
    const int lane = threadIdx.x & 63;
    const long N = (long)h8 * w8;
    const long qid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    bool nz = false;
    if (qid < (long)B * N) {
        const int b = (int)(qid / N);
        const int pix = (int)(qid % N);
        const int y = pix / w8, x = pix % w8;
        const float* m = mask + ((long)b * N + pix) * mask_ld + lane;   // channel k*64 + lane
        float lg[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = m[k * 64];
            mx = fmaxf(mx, lg[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            lg[k] = expf(lg[k] - mx);
            den += lg[k];
        }
        const float* cx = coords1 + ((long)b * 2 + 0) * N;
        const float* cy = coords1 + ((long)b * 2 + 1) * N;
        float ux = 0.f, uy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            float fx = 0.f, fy = 0.f;      // F.unfold pads with zeros
            if (yy >= 0 && yy < h8 && xx >= 0 && xx < w8) {
                fx = cx[(long)yy * w8 + xx];
                fy = cy[(long)yy * w8 + xx];
                if (!coords_is_flow) {
                    fx -= (float)xx;
                    fy -= (float)yy;
                }
            }
            const float wgt = lg[k] / den;
            ux += wgt * (8.f * fx);
            uy += wgt * (8.f * fy);
        }
        const int i = lane >> 3, j = lane & 7;
        const int oy = 8 * y + i, ox = 8 * x + j;
        const int Hp = 8 * h8, Wp = 8 * w8;
        const long o0 = (((long)b * 2 + 0) * Hp + oy) * Wp + ox;
        const long o1 = (((long)b * 2 + 1) * Hp + oy) * Wp + ox;
        if (flow_up) {
            flow_up[o0] = ux;
            flow_up[o1] = uy;
        }
        float tx = ux, ty = uy;
        if (add) {     // flow_total = flow_total + delta_flow (idedeq.py:206-207)
            tx = add[o0] + ux;
            ty = add[o1] + uy;
        }
        if (total_out) {
            total_out[o0] = tx;
            total_out[o1] = ty;
        }
        if (flow_final && oy >= padH && ox >= padW) {
            flow_final[(((long)b * 2 + 0) * H + (oy - padH)) * W + (ox - padW)] = tx;
            flow_final[(((long)b * 2 + 1) * H + (oy - padH)) * W + (ox - padW)] = ty;
            nz = (tx != 0.0f) || (ty != 0.0f);
        }
    }
    if (flag && __any(nz)) {
        if (lane == 0) *flag = 1;
    }
}

hipError_t launch_convex_upsample(const float* coords1, int coords_is_flow, const float* mask, int mask_ld, int B, int h8,
                                  int w8, float* flow_up, float* flow_final, int H, int W, int padH, int padW, int* flag,
                                  const float* add, float* total_out, hipStream_t s) {
    if (!coords1 || !mask || mask_ld < 576 || B <= 0 || h8 <= 0 || w8 <= 0) return hipErrorInvalidValue;
    if (flow_final && (H + padH != h8 * 8 || W + padW != w8 * 8)) return hipErrorInvalidValue;
    const long nq = (long)B * h8 * w8;
    note_launch("convex_upsample_kernel", dim3((unsigned)((nq + 3) / 4)), dim3(256));
    hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, coords1, coords_is_flow,
                       mask, mask_ld, B, h8, w8, flow_up, flow_final, H, W, padH, padW, flag, add, total_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// IDNet deblur: per-bin bilinear gather along the flow (zeros padding, align_corners=False on a (W-1)-normalised
// grid -- ATen CPU: ix = (g + 1) * (W / 2) - 0.5)
// ---------------------------------------------------------------------------
__global__ void idn_deblur_kernel(const float* __restrict__ bins, const float* __restrict__ flow, float* __restrict__ out,
                                  int B, int T, int H, int W, int padH, int padW) {
    const int Hp = H + padH, Wp = W + padW;
    const long total = (long)B * T * Hp * Wp;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int x = (int)(gid % Wp);
    const int y = (int)((gid / Wp) % Hp);
    const int t = (int)((gid / ((long)Wp * Hp)) % T);
    const int b = (int)(gid / ((long)Wp * Hp * T));
    float fx = 0.f, fy = 0.f;
    if (flow) {
        fx = flow[(((long)b * 2 + 0) * Hp + y) * Wp + x];
        fy = flow[(((long)b * 2 + 1) * Hp + y) * Wp + x];
    }
    const float dx = fx * (float)t / (float)(T - 1);
    const float dy = fy * (float)t / (float)(T - 1);
    const float gx = ((float)x + dx) / (float)(Wp - 1) * 2.f - 1.f;
    const float gy = ((float)y + dy) / (float)(Hp - 1) * 2.f - 1.f;
    const float ix = (gx + 1.f) * ((float)Wp / 2.f) - 0.5f;
    const float iy = (gy + 1.f) * ((float)Hp / 2.f) - 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f;
    const float* src = bins + ((long)b * T + t) * H * W;
    // the source is ImagePadder-padded: padded pixel (py,px) = bins[py-padH][px-padW], zero in the pad
    auto tap = [&](float yf, float xf) -> float {
        if (!(yf >= 0.f && yf <= (float)(Hp - 1) && xf >= 0.f && xf <= (float)(Wp - 1))) return 0.f;
        const int py = (int)yf - padH, px = (int)xf - padW;
        if (py < 0 || px < 0) return 0.f;
        return src[(long)py * W + px];
    };
    const float v = tap(y0f, x0f) * ((1.f - tx) * (1.f - ty)) + tap(y0f, x0f + 1.f) * (tx * (1.f - ty)) +
                    tap(y0f + 1.f, x0f) * ((1.f - tx) * ty) + tap(y0f + 1.f, x0f + 1.f) * (tx * ty);
    out[gid] = v;
}

hipError_t launch_idn_deblur(const float* bins, const float* flow, float* out, int B, int T, int H, int W, int padH,
                             int padW, hipStream_t s) {
    if (!bins || !out || B <= 0 || T < 2 || H <= 1 || W <= 1) return hipErrorInvalidValue;
    const long total = (long)B * T * (H + padH) * (W + padW);
    note_launch("idn_deblur_kernel", dim3((unsigned)((total + 255) / 256)), dim3(256));
    hipLaunchKernelGGL(idn_deblur_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, bins, flow, out, B, T, H, W,
                       padH, padW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// f-1: events -> voxel grid (utils/event_process.py:15-72) and 'std' normalisation (:193-216)
// ---------------------------------------------------------------------------
__global__ void events_scatter_kernel(const double* __restrict__ ev, const long* __restrict__ offsets, int B, int bins,
                                      int H, int W, float* __restrict__ voxel) {
    const int b = blockIdx.y;
    const long e0 = offsets[b], e1 = offsets[b + 1];
    const long n = e1 - e0;
    if (n <= 0) return;
    const double first = ev[e0 * 4], last = ev[(e1 - 1) * 4];
    double deltaT = last - first;
    if (deltaT == 0) deltaT = 1.0;
    float* vox = voxel + (long)b * bins * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double* r = ev + (e0 + i) * 4;
        const double ts = (double)(bins - 1) * (r[0] - first) / deltaT;
        const unsigned long xs = (unsigned long)r[1], ys = (unsigned long)r[2];
        double pol = r[3];
        if (pol == 0) pol = -1;                       // polarity is +1 / -1
        const unsigned long ti = (unsigned long)ts;
        const double dt = ts - (double)ti;
        if (xs >= (unsigned long)W || ys >= (unsigned long)H) continue;   // numpy would index out of range
        const long base = (long)xs + (long)ys * W;
        if (ti < (unsigned long)bins) atomicAdd(vox + base + (long)ti * W * H, (float)(pol * (1.0 - dt)));
        if (ti + 1 < (unsigned long)bins) atomicAdd(vox + base + (long)(ti + 1) * W * H, (float)(pol * dt));
    }
}

// stats[b] = {count of non-zeros, sum, sum of squares} (fp64)
// hot > 0: `event_voxel_grid[abs(event_voxel_grid) > 25./num_bins] = 0` first (event_preprocess(filter_hot_pixel=True),
// event_process.py:196-198)
__global__ void voxel_stats_kernel(const float* __restrict__ voxel, long per_seq, double* __restrict__ stats, float hot) {
    __shared__ double sh[3][256];
    const int b = blockIdx.y;
    const float* v = voxel + (long)b * per_seq;
    double c = 0, s = 0, ss = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_seq; i += (long)gridDim.x * blockDim.x) {
        float xf = v[i];
        if (hot > 0.f && fabsf(xf) > hot) xf = 0.f;
        const double x = (double)xf;
        if (x != 0) c += 1.0;
        s += x;
        ss += x * x;
    }
    sh[0][threadIdx.x] = c; sh[1][threadIdx.x] = s; sh[2][threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) atomicAdd(stats + b * 3 + k, sh[k][0]);
}

__global__ void voxel_normalize_kernel(float* __restrict__ voxel, long per_seq, const double* __restrict__ stats, float hot) {
    const int b = blockIdx.y;
    const double cnt = stats ? stats[b * 3 + 0] : 0.0;      // stats == nullptr: the hot-pixel filter alone
    if (cnt <= 0 && !(hot > 0.f)) return;
    const double mean = cnt > 0 ? stats[b * 3 + 1] / cnt : 0.0;
    const double sd = cnt > 0 ? sqrt(stats[b * 3 + 2] / cnt - mean * mean) : 1.0;
    float* v = voxel + (long)b * per_seq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_seq; i += (long)gridDim.x * blockDim.x) {
        const float x = v[i];
        if (hot > 0.f && fabsf(x) > hot) v[i] = 0.f;
        else if (x != 0.f && cnt > 0) v[i] = (float)(((double)x - mean) / (sd + 1e-8));
    }
}

hipError_t launch_events_to_voxel(const double* events, const long* offsets, int B, int bins, int H, int W, float* voxel,
                                  double* stats, int normalize, hipStream_t s, float hot) {
    if (!events || !offsets || !voxel || B <= 0 || bins <= 0 || H <= 0 || W <= 0 || (normalize && !stats))
        return hipErrorInvalidValue;
    const long per_seq = (long)bins * H * W;
    hipError_t e = hipMemsetAsync(voxel, 0, sizeof(float) * per_seq * B, s);
    if (e != hipSuccess) return e;
    note_launch("events_scatter_kernel", dim3(64, B), dim3(256));
    hipLaunchKernelGGL(events_scatter_kernel, dim3(64, B), dim3(256), 0, s, events, offsets, B, bins, H, W, voxel);
    if (normalize) {
        e = hipMemsetAsync(stats, 0, sizeof(double) * 3 * B, s);
        if (e != hipSuccess) return e;
        note_launch("voxel_stats_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_stats_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, stats, hot);
        note_launch("voxel_normalize_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_normalize_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, stats, hot);
    } else if (hot > 0.f) {
        // event_preprocess applies the hot-pixel filter whatever the normalisation mode (event_process.py:196-198)
        note_launch("voxel_normalize_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_normalize_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, (const double*)nullptr, hot);
    }
    return hipGetLastError();
}

// event_preprocess(mode='std') of grids that already sit on the device (event_process.py:193-216), in place: optional
// hot-pixel filter, then mean 0 / std 1 over the non-zero voxels of each of the B grids
hipError_t launch_voxel_preprocess(float* voxel, int B, long per_seq, double* stats, int normalize, float hot, hipStream_t s) {
    if (!voxel || B <= 0 || per_seq <= 0 || (normalize && !stats)) return hipErrorInvalidValue;
    if (normalize) {
        hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 3 * B, s);
        if (e != hipSuccess) return e;
        note_launch("voxel_stats_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_stats_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, stats, hot);
        note_launch("voxel_normalize_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_normalize_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, stats, hot);
    } else if (hot > 0.f) {
        note_launch("voxel_normalize_kernel", dim3(64, B), dim3(256));
        hipLaunchKernelGGL(voxel_normalize_kernel, dim3(64, B), dim3(256), 0, s, voxel, per_seq, (const double*)nullptr, hot);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// boundary shuffles (recurrent states arrive / leave as whatever the caller holds)
// ---------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int dst_ld, int B, int C,
                                    int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        tile[k][tx] = (c < C && p < HW) ? src[((long)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        if (p < HW && c < C) dst[((long)b * HW + p) * dst_ld + c] = tile[tx][k];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int B, int C,
                                    int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int p = p0 + k, c = c0 + tx;
        tile[k][tx] = (c < C && p < HW) ? src[((long)b * HW + p) * src_ld + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, p = p0 + tx;
        if (p < HW && c < C) dst[((long)b * C + c) * HW + p] = tile[tx][k];
    }
}

hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int dst_ld, int B, int C, int HW, hipStream_t s) {
    if (!src || !dst || dst_ld < C) return hipErrorInvalidValue;
    note_launch("nchw_to_nhwc_kernel", dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256));
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, src, dst, dst_ld, B, C,
                       HW);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float* src, int src_ld, float* dst, int B, int C, int HW, hipStream_t s) {
    if (!src || !dst || src_ld < C) return hipErrorInvalidValue;
    note_launch("nhwc_to_nchw_kernel", dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256));
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, src, src_ld, dst, B, C,
                       HW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// F.interpolate(size = 2h x 2w, bilinear, align_corners=False) of an NHWC tensor (base_layers.py:200-202), written out
// so that the following reflect-padded 3x3 conv can read it through the LDS-DMA kernel.  Same tap / blend arithmetic
// as the fused A_UPS2X read of the convolution kernel.
// ---------------------------------------------------------------------------
// One thread = one channel quad of a 2x2 block of OUTPUT pixels {2k+1, 2k+2} x {2j+1, 2j+2}: with scale 2 /
// align_corners=False those four share the source pixels {k, k+1} x {j, j+1}, so 4 16-byte loads feed 4 16-byte
// stores (the one-thread-per-output form read 16).  Block rows / columns k, j = -1 and the last one are the image
// border: source indices are clamped for the loads while every output keeps the weights of its own row / column
// (same expressions as before: results are unchanged).
__global__ __launch_bounds__(256) void upsample2x_nhwc_kernel(const float* __restrict__ src, int s_ld, long s_bs, float* __restrict__ dst,
                                                              int d_ld, long d_bs, int B, int Hs, int Ws, int C) {
    const int cq = C >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (Ws + 1) * cq) return;
    const int jj = i / cq;
    const int c0 = (i - jj * cq) * 4;
    const int j = jj - 1, k = (int)blockIdx.y - 1;
    const int b = blockIdx.z;
    const int r0 = k < 0 ? 0 : k, r1 = k + 1 > Hs - 1 ? Hs - 1 : k + 1;
    const int q0 = j < 0 ? 0 : j, q1 = j + 1 > Ws - 1 ? Ws - 1 : j + 1;
    const float* sb = src + (long)b * s_bs + c0;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(sb + (long)(r0 * Ws + q0) * s_ld);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(sb + (long)(r0 * Ws + q1) * s_ld);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(sb + (long)(r1 * Ws + q0) * s_ld);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(sb + (long)(r1 * Ws + q1) * s_ld);
    const int Wo = 2 * Ws, Ho = 2 * Hs;
    float* db = dst + (long)b * d_bs + c0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = 2 * k + 1 + a;
        if (oy < 0 || oy >= Ho) continue;
        float sy = ((float)oy + 0.5f) * 0.5f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        // (y0, y1) of this output row is (r0, r1); on the clamped border rows both taps hold the same data or the
        // far tap has weight 0
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int ox = 2 * j + 1 + e2;
            if (ox < 0 || ox >= Wo) continue;
            float sx = ((float)ox + 0.5f) * 0.5f - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            const int x0 = (int)sx;
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(db + (long)(oy * Wo + ox) * d_ld));
        }
    }
}

hipError_t launch_upsample2x_nhwc(const float* src, int s_ld, long s_bs, float* dst, int d_ld, long d_bs, int B, int Hs, int Ws,
                                  int C, hipStream_t s) {
    if (!src || !dst || B <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || (C % 4) != 0 || (s_ld % 4) != 0 || (d_ld % 4) != 0 ||
        (s_bs % 4) != 0 || (d_bs % 4) != 0 || (reinterpret_cast<uintptr_t>(src) & 15) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15) != 0)
        return hipErrorInvalidValue;
    if (Hs + 1 > 65535 || B > 65535) return hipErrorInvalidValue;
    const int row = (Ws + 1) * (C / 4);
    note_launch("upsample2x_nhwc_kernel", dim3((unsigned)((row + 255) / 256), Hs + 1, B), dim3(256));
    hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)((row + 255) / 256), Hs + 1, B), dim3(256), 0, s, src, s_ld, s_bs, dst, d_ld,
                       d_bs, B, Hs, Ws, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// f-2 output stage: np.uint8(pred * 255.)  (test_with_flow.py:174) -- fp32 product, truncation toward zero
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(x + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out + i) & 3) == 0)) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        uchar4 o;
        o.x = (unsigned char)(v[0] * 255.f);
        o.y = (unsigned char)(v[1] * 255.f);
        o.z = (unsigned char)(v[2] * 255.f);
        o.w = (unsigned char)(v[3] * 255.f);
        *reinterpret_cast<uchar4*>(out + i) = o;
    } else {
        for (long k = i; k < n && k < i + 4; ++k) out[k] = (unsigned char)(x[k] * 255.f);
    }
}

// ---------------------------------------------------------------------------
// f-2 output stage: flow -> HSV -> BGR colour coding of FlowWriter (utils/data_io.py:9-29, merge_optical_flow):
//     magnitude, angle = cv2.cartToPolar(u, v); H = uint8(angle * 180 / pi / 2); S = 255; V = uint8(255 * magnitude / magnitude.max());
//     bgr = cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR)
// UNPINNED: cv2 is not installed in the build container, so no reference-run vector exists.  What is restated is OpenCV's PUBLISHED
// arithmetic: cartToPolar's angle in [0, 2 pi) (here atan2f, OpenCV uses a polynomial of ~0.01 degree accuracy: a pixel whose hue sits on an
// integer boundary can land in the neighbouring 2-degree bucket), the truncating uint8 casts numpy does, and the 8-bit HSV -> BGR of
// cvtColor: h6 = H / 30 (hue range 180), sector = floor(h6), f = h6 - sector, tab = {v, v (1 - s), v (1 - s f), v (1 - s (1 - f))} with s, v
// in [0, 1], channel = round-half-even(255 * tab[sector table]).  One image = one maximum: pass 1 folds max |flow| per image with an
// integer atomicMax on the float's bits (magnitudes are >= 0, so the orders agree; deterministic), pass 2 colours.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_maxmag_kernel(const float* __restrict__ flow, long HW, unsigned* __restrict__ maxbits) {
    const int b = blockIdx.y;
    const float* u = flow + (long)b * 2 * HW;
    const float* v = u + HW;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, sqrtf(u[i] * u[i] + v[i] * v[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(maxbits + b, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void flow_to_bgr_kernel(const float* __restrict__ flow, long HW, const unsigned* __restrict__ maxbits,
                                                          unsigned char* __restrict__ out) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float u = flow[(long)b * 2 * HW + i], v = flow[(long)b * 2 * HW + HW + i];
    const float mag = sqrtf(u * u + v * v);
    float ang = atan2f(v, u);
    if (ang < 0.f) ang += 6.283185307179586f;
    const float mx = __uint_as_float(maxbits[b]);
    // numpy: float32 arithmetic, then the C cast of the uint8 assignment (truncation); an all-zero flow gives 0 / 0 = NaN -> 0 here
    const int H = (int)(ang * 180.f / 3.14159265358979323846f / 2.f) & 255;
    const int V = mx > 0.f ? (int)(255.f * mag / mx) : 0;
    const float vv = (float)V * (1.f / 255.f);
    float h6 = (float)H * (6.f / 180.f);
    int sector = (int)floorf(h6);
    const float f = h6 - (float)sector;
    sector = sector % 6;                       // H < 180 after the cast above unless the angle rounds to 2 pi: hue 180 = sector 6 = sector 0
    const float tab[4] = {vv, 0.f, vv * (1.f - f), vv * f};       // s = 1
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};      // (b, g, r) per sector
    unsigned char* o = out + ((long)b * HW + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = tab[sd[sector][c]] * 255.f;
        t = fminf(fmaxf(t, 0.f), 255.f);
        o[c] = (unsigned char)__float2int_rn(t);
    }
}

hipError_t launch_flow_to_bgr(const float* flow, int B, int H, int W, unsigned char* out, unsigned* scratch, hipStream_t s) {
    if (!flow || !out || !scratch || B <= 0 || H <= 0 || W <= 0) return hipErrorInvalidValue;
    const long HW = (long)H * W;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(unsigned) * (size_t)B, s);
    if (e != hipSuccess) return e;
    const unsigned nb = (unsigned)((HW + 255) / 256);
    note_launch("flow_maxmag_kernel", dim3(nb < 64 ? nb : 64, (unsigned)B), dim3(256));
    hipLaunchKernelGGL(flow_maxmag_kernel, dim3(nb < 64 ? nb : 64, (unsigned)B), dim3(256), 0, s, flow, HW, scratch);
    note_launch("flow_to_bgr_kernel", dim3(nb, (unsigned)B), dim3(256));
    hipLaunchKernelGGL(flow_to_bgr_kernel, dim3(nb, (unsigned)B), dim3(256), 0, s, flow, HW, scratch, out);
    return hipGetLastError();
}

hipError_t launch_quantize_u8(const float* x, unsigned char* out, long n, hipStream_t s) {
    if (!x || !out || n <= 0) return hipErrorInvalidValue;
    const long threads = (n + 3) / 4;
    note_launch("quantize_u8_kernel", dim3((unsigned)((threads + 255) / 256)), dim3(256));
    hipLaunchKernelGGL(quantize_u8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, out, n);
    return hipGetLastError();
}

}  // namespace cf
