// cf_device.h -- device helpers shared by the HBM-class kernels (pointwise.hip) and the metric kernels (metrics.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// grid_sample coordinate helpers -- restated from ATen's CPU grid sampler
// (align_corners=True): unnormalize = (g + 1) * ((size-1)/2); reflection about
// [0, size-1]:  extra = |x| - trunc(|x| / (2*span)) * 2*span ; min(extra, 2*span - extra).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float reflect_coord(float x, int size) {
    if (size <= 1) return 0.f;
    const float twice_span = (float)(size - 1) * 2.f;
    const float a = fabsf(x);
    const float flips = truncf(a / twice_span);
    const float extra = a - flips * twice_span;
    return fminf(extra, twice_span - extra);
}

// interpolate(..., mode='bilinear', align_corners=True) source index for output index d
__device__ __forceinline__ void ac_true_src(int d, int in, int out, int& i0, int& i1, float& l0, float& l1) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)d;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// flow value at (ch, y, x) of a (H,W) grid, resampled from one image's planar [2][Hf][Wf] flow with
// interpolate(bilinear, align_corners=True) WITHOUT rescaling its values (e2v_model.py:190)
__device__ __forceinline__ float flow_at(const float* f, int y, int x, int H, int W, int Hf, int Wf) {
    if (Hf == H && Wf == W) return f[(long)y * Wf + x];
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    ac_true_src(y, Hf, H, y0, y1, ly0, ly1);
    ac_true_src(x, Wf, W, x0, x1, lx0, lx1);
    const float v00 = f[(long)y0 * Wf + x0], v01 = f[(long)y0 * Wf + x1];
    const float v10 = f[(long)y1 * Wf + x0], v11 = f[(long)y1 * Wf + x1];
    return ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

// The four bilinear taps of the flow warp at output pixel (x, y) for displacement (u, v):
// utils/flow_utils.py:153-190 (forward: x - u) / :83-120 (backward: x + u); grid = 2*(xs/W - 0.5) with W, not W-1,
// then grid_sample(bilinear, align_corners=True, padding_mode='reflection').  p = pixel index y*W + x of each tap (always
// inside the image), w = its weight (0 for a tap past the last row / column, exactly as ATen drops it).
struct WarpTaps {
    int p00, p01, p10, p11;
    float w00, w01, w10, w11;
};
__device__ __forceinline__ WarpTaps warp_taps(float u, float v, int x, int y, int H, int W, int backward) {
    float xs = backward ? ((float)x + u) : ((float)x - u);
    float ys = backward ? ((float)y + v) : ((float)y - v);
    xs = 2.f * (xs / (float)W - 0.5f);
    ys = 2.f * (ys / (float)H - 0.5f);
    float ix = (xs + 1.f) * ((float)(W - 1) / 2.f);
    float iy = (ys + 1.f) * ((float)(H - 1) / 2.f);
    ix = reflect_coord(ix, W);
    iy = reflect_coord(iy, H);
    // clip (ATen clips after reflecting); the !(>=) form also sends NaN coordinates to 0 instead of an int cast of NaN
    ix = !(ix >= 0.f) ? 0.f : fminf(ix, (float)(W - 1));
    iy = !(iy >= 0.f) ? 0.f : fminf(iy, (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;   // ATen: w = x - x_w ; e = 1 - w
    const float wx0 = 1.f - tx, wy0 = 1.f - ty;
    const bool x1ok = x0 + 1 <= W - 1, y1ok = y0 + 1 <= H - 1;   // x0, y0 always in range after the clip
    const int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
    WarpTaps t;
    t.p00 = y0 * W + x0; t.p01 = y0 * W + x1; t.p10 = y1 * W + x0; t.p11 = y1 * W + x1;
    t.w00 = wy0 * wx0;
    t.w01 = x1ok ? wy0 * tx : 0.f;
    t.w10 = y1ok ? ty * wx0 : 0.f;
    t.w11 = (x1ok && y1ok) ? ty * tx : 0.f;
    return t;
}

// block-wide sum of K doubles per thread (256 threads): result valid in thread 0; sh must hold 4*K doubles
template <int K>
__device__ __forceinline__ void block_sum_256(double (&v)[K], double* sh) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double a = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        v[k] = a;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ((sh[k] + sh[K + k]) + sh[2 * K + k]) + sh[3 * K + k];
    }
}

}  // namespace cf
