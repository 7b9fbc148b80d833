// metrics.hip -- f-3: the evaluation metrics the reference's drivers compute per frame, on the device,
// so that a frame's scores leave the GPU as a handful of doubles instead of the frames themselves.
//
//   cf_metrics_recon   ReconLoss.evaluate's pure-torch half: mse (nn.MSELoss) and PSNR      loss.py:15-24,316-328
//   cf_metrics_flow    FlowL1LossDict.evaluate: photo_loss, epe, 1/3/5 px outliers, `out`    loss.py:237-265
//   cf_metrics_fwl     voxel_warping_flow_loss (flow-warp loss, FWL) for the flow and for zero flow   loss.py:27-83,
//                      test_wo_flow.py:161
//   cf_metrics_ssim    ReconLoss.evaluate's 'ssim' = pytorch_msssim.SSIM(data_range=1, channel=1): 11-tap gaussian (sigma 1.5),
//                      valid convolution, K = (0.01, 0.03).  The package is absent offline; the kernel follows its published
//                      algorithm (restated in oracle/cista_oracle.py::ssim) -- parity UNPINNED by the reference for this metric.
// LPIPS (lpips + torchvision AlexNet weights from the network) is not restated.
//
// Every kernel writes one partial-sum row per workgroup and a one-workgroup fold kernel adds the rows in a fixed
// order in fp64: deterministic, and more accurate than the fp32 tree sums of torch.mean (results agree with the
// reference to fp32 rounding of its own reductions).
#include "cf_device.h"
#include "cf_kernels.h"

namespace cf {

static constexpr int MET_BLOCKS = 1024;     // partial rows (scratch = MET_BLOCKS * 8 doubles)

__global__ __launch_bounds__(256) void met_recon_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                        double* __restrict__ partial) {
    __shared__ double sh[4];
    double s[1] = {0.0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = a[i] / 1.0f - b[i] / 1.0f;        // loss.py:21
        s[0] += (double)(d * d);
    }
    block_sum_256<1>(s, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 8] = s[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) partial[blockIdx.x * 8 + k] = 0.0;
    }
}

// kind 0: out = {mse, psnr}; 1: flow metrics; 2: FWL
__global__ __launch_bounds__(256) void met_fold_kernel(const double* __restrict__ partial, int nrows, int kind, double n,
                                                       double* __restrict__ out) {
    __shared__ double sh[4 * 8];
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = threadIdx.x; r < nrows; r += 256)
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += partial[r * 8 + k];
    block_sum_256<8>(s, sh);
    if (threadIdx.x != 0) return;
    if (kind == 0) {
        const double mse = s[0] / n;
        out[0] = mse;
        out[1] = mse < 1.0e-10 ? 100.0 : 20.0 * log10(1.0 / sqrt(mse));      // PSNR(data_range=1), loss.py:20-24
    } else if (kind == 3) {
        out[0] = s[0] / n;          // mean of the SSIM map over every plane's valid window positions
        out[1] = s[1] / n;          // mean of the contrast-structure map (pytorch_msssim's `cs`)
    } else if (kind == 1) {
        // s: 0 photo sum, 1 #valid, 2 epe sum, 3 #(epe>1), 4 #(epe>3), 5 #(epe>5), 6 #out   (over valid > 0)
        const double nv = s[1];
        out[0] = s[0] / n;
        out[1] = s[2] / nv;
        out[2] = s[3] / nv;
        out[3] = s[4] / nv;
        out[4] = s[5] / nv;
        out[5] = s[6] / nv * 100.0;
    } else {
        // unbiased variance (torch.Tensor.var) of the warped event image: s0/s1 = sum, sum of squares with the flow,
        // s2/s3 with zero flow
        const double v1 = (s[1] - s[0] * s[0] / n) / (n - 1.0);
        const double v0 = (s[3] - s[2] * s[2] / n) / (n - 1.0);
        out[0] = v1;
        out[1] = v0;
        out[2] = v1 / v0;
    }
}

// one image, one channel: FrameWarp.warp_frame sample at pixel (x, y)
__device__ __forceinline__ float warp1(const float* img, float u, float v, int x, int y, int H, int W, int backward) {
    const WarpTaps t = warp_taps(u, v, x, y, H, W, backward);
    return img[t.p00] * t.w00 + img[t.p01] * t.w01 + img[t.p10] * t.w10 + img[t.p11] * t.w11;
}

__global__ __launch_bounds__(256) void met_flow_kernel(const float* __restrict__ flow, const float* __restrict__ gt,
                                                       const float* __restrict__ img0, const float* __restrict__ img1,
                                                       const float* __restrict__ valid_in, int B, int H, int W, int backward,
                                                       float max_flow, double* __restrict__ partial) {
    __shared__ double sh[4 * 8];
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long HW = (long)H * W, total = (long)B * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (long)b * HW);
        const int y = p / W, x = p - y * W;
        const float* fb = flow + (long)b * 2 * HW;
        const float* gb = gt + (long)b * 2 * HW;
        const float* i0 = img0 + (long)b * HW;
        const float fu = fb[p], fv = fb[HW + p], gu = gb[p], gv = gb[HW + p], t1 = img1[i];
        float vo;
        if (valid_in) {
            vo = valid_in[i];
        } else {   // exp(-50 * mse(warp(gt_img0, gt_flow), gt_img1, reduction='none'))   loss.py:241
            const float d = warp1(i0, gu, gv, x, y, H, W, backward) - t1;
            vo = expf(-50.f * (d * d));
        }
        const float mag = sqrtf(gu * gu + gv * gv);
        const float valid = vo * (mag < max_flow ? 1.f : 0.f);                       // :244
        const float photo = fabsf(warp1(i0, fu, fv, x, y, H, W, backward) - t1);     // :247
        const float du = fu - gu, dv = fv - gv;
        const float epe = sqrtf(valid * (du * du) + valid * (dv * dv));              // :248 (valid inside the root)
        s[0] += (double)photo;
        if (valid > 0.f) {
            s[1] += 1.0;
            s[2] += (double)epe;
            s[3] += epe > 1.f ? 1.0 : 0.0;
            s[4] += epe > 3.f ? 1.0 : 0.0;
            s[5] += epe > 5.f ? 1.0 : 0.0;
            s[6] += (epe > 3.0f && (epe / mag) > 0.05f) ? 1.0 : 0.0;                 // :250
        }
    }
    block_sum_256<8>(s, sh);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) partial[blockIdx.x * 8 + k] = s[k];
}

// grid_sample(bilinear, align_corners=True, zeros) of one plane at grid (gx, gy) given in pixels BEFORE the 2x/W - 1
// normalisation of loss.py:63-64 (W, not W-1: the sampled pixel is x*(W-1)/W)
__device__ __forceinline__ float fwl_sample(const float* plane, float gx, float gy, int H, int W) {
    const float nx = (2.0f * gx) / (float)W - 1.0f;
    const float ny = (2.0f * gy) / (float)H - 1.0f;
    const float ix = ((nx + 1.f) / 2.f) * (float)(W - 1);
    const float iy = ((ny + 1.f) / 2.f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const bool inx0 = fx >= 0.f && fx <= (float)(W - 1), inx1 = fx + 1.f >= 0.f && fx + 1.f <= (float)(W - 1);
    const bool iny0 = fy >= 0.f && fy <= (float)(H - 1), iny1 = fy + 1.f >= 0.f && fy + 1.f <= (float)(H - 1);
    const int x0 = inx0 ? (int)fx : 0, x1 = inx1 ? (int)fx + 1 : 0;
    const int y0 = iny0 ? (int)fy : 0, y1 = iny1 ? (int)fy + 1 : 0;
    const float v00 = (inx0 && iny0) ? plane[(long)y0 * W + x0] : 0.f;
    const float v01 = (inx1 && iny0) ? plane[(long)y0 * W + x1] : 0.f;
    const float v10 = (inx0 && iny1) ? plane[(long)y1 * W + x0] : 0.f;
    const float v11 = (inx1 && iny1) ? plane[(long)y1 * W + x1] : 0.f;
    return v00 * ((1.f - tx) * (1.f - ty)) + v01 * (tx * (1.f - ty)) + v10 * ((1.f - tx) * ty) + v11 * (tx * ty);
}

__global__ __launch_bounds__(256) void met_fwl_kernel(const float* __restrict__ voxel, const float* __restrict__ flow, int B,
                                                      int C, int H, int W, double* __restrict__ partial) {
    __shared__ double sh[4 * 4];
    double s[4] = {0, 0, 0, 0};
    const long HW = (long)H * W, total = (long)B * HW;
    const double inc = 1.0 / ((double)C - 1.0);                 // displacement_increment, a Python float (:49)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / HW);
        const int p = (int)(i - (long)b * HW);
        const int y = p / W, x = p - y * W;
        const float dx = flow[(long)b * 2 * HW + p], dy = flow[(long)b * 2 * HW + HW + p];
        float acc1 = 0.f, acc0 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float ratio = (float)((double)c * inc);       // tensor * python scalar: the scalar is cast to fp32
            const float* plane = voxel + ((long)b * C + c) * HW;
            acc1 += fwl_sample(plane, (float)x + dx * ratio, (float)y + dy * ratio, H, W);
            acc0 += fwl_sample(plane, (float)x + 0.f * ratio, (float)y + 0.f * ratio, H, W);
        }
        s[0] += (double)acc1;
        s[1] += (double)acc1 * (double)acc1;
        s[2] += (double)acc0;
        s[3] += (double)acc0 * (double)acc0;
    }
    block_sum_256<4>(s, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 8 + 0] = s[0];
        partial[blockIdx.x * 8 + 1] = s[1];
        partial[blockIdx.x * 8 + 2] = s[2];
        partial[blockIdx.x * 8 + 3] = s[3];
#pragma unroll
        for (int k = 4; k < 8; ++k) partial[blockIdx.x * 8 + k] = 0.0;
    }
}

// ---- SSIM (pytorch_msssim._ssim): separable 11-tap gaussian, H direction first, then W, 'valid' (no padding) ----
struct SsimWin { float g[11]; };
static constexpr int SS_TY = 16, SS_TX = 64;                  // output tile of a workgroup
static constexpr int SS_IY = SS_TY + 10, SS_IX = SS_TX + 10;  // input tile

__global__ __launch_bounds__(256) void met_ssim_kernel(const float* __restrict__ X, const float* __restrict__ Y, int planes, int H,
                                                       int W, SsimWin win, float C1, float C2, double* __restrict__ partial) {
    __shared__ float sX[SS_IY][SS_IX], sY[SS_IY][SS_IX];
    __shared__ float sV[5][SS_TY][SS_IX + 1];
    __shared__ double sh[4 * 2];
    const int Ho = H - 10, Wo = W - 10;
    const int nty = (Ho + SS_TY - 1) / SS_TY, ntx = (Wo + SS_TX - 1) / SS_TX;
    const long ntiles = (long)planes * nty * ntx;
    double s[2] = {0.0, 0.0};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int pl = (int)(t / (nty * ntx));
        const int r = (int)(t - (long)pl * nty * ntx);
        const int oy0 = (r / ntx) * SS_TY, ox0 = (r % ntx) * SS_TX;
        const float* xp = X + (long)pl * H * W;
        const float* yp = Y + (long)pl * H * W;
        __syncthreads();                                      // the previous tile's passes are done with the buffers
        for (int i = threadIdx.x; i < SS_IY * SS_IX; i += 256) {
            const int iy = i / SS_IX, ix = i - iy * SS_IX;
            const int gy = oy0 + iy, gx = ox0 + ix;
            const bool ok = gy < H && gx < W;
            sX[iy][ix] = ok ? xp[(long)gy * W + gx] : 0.f;
            sY[iy][ix] = ok ? yp[(long)gy * W + gx] : 0.f;
        }
        __syncthreads();
        // H direction: the five filtered quantities x, y, x*x, y*y, x*y at (output row, input column)
        for (int i = threadIdx.x; i < SS_TY * SS_IX; i += 256) {
            const int oy = i / SS_IX, ix = i - oy * SS_IX;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float x = sX[oy + k][ix], y = sY[oy + k][ix], g = win.g[k];
                a0 += g * x; a1 += g * y; a2 += g * (x * x); a3 += g * (y * y); a4 += g * (x * y);
            }
            sV[0][oy][ix] = a0; sV[1][oy][ix] = a1; sV[2][oy][ix] = a2; sV[3][oy][ix] = a3; sV[4][oy][ix] = a4;
        }
        __syncthreads();
        // W direction + the SSIM expression
        for (int i = threadIdx.x; i < SS_TY * SS_TX; i += 256) {
            const int oy = i / SS_TX, ox = i - oy * SS_TX;
            if (oy0 + oy >= Ho || ox0 + ox >= Wo) continue;
            float m1 = 0.f, m2 = 0.f, xx = 0.f, yy = 0.f, xy = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = win.g[k];
                m1 += g * sV[0][oy][ox + k]; m2 += g * sV[1][oy][ox + k]; xx += g * sV[2][oy][ox + k];
                yy += g * sV[3][oy][ox + k]; xy += g * sV[4][oy][ox + k];
            }
            const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
            const float s1 = xx - m1s, s2 = yy - m2s, s12 = xy - m12;
            const float cs = (2.f * s12 + C2) / (s1 + s2 + C2);
            const float v = ((2.f * m12 + C1) / (m1s + m2s + C1)) * cs;
            s[0] += (double)v;
            s[1] += (double)cs;
        }
    }
    block_sum_256<2>(s, sh);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 8 + 0] = s[0];
        partial[blockIdx.x * 8 + 1] = s[1];
#pragma unroll
        for (int k = 2; k < 8; ++k) partial[blockIdx.x * 8 + k] = 0.0;
    }
}

static int met_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > MET_BLOCKS ? MET_BLOCKS : b));
}

long metrics_scratch_doubles() { return (long)MET_BLOCKS * 8; }

hipError_t launch_metrics_recon(const float* rec, const float* tgt, long n, double* out, double* scratch, hipStream_t s) {
    if (!rec || !tgt || !out || !scratch || n <= 0) return hipErrorInvalidValue;
    const int nb = met_blocks(n);
    note_launch("met_recon_kernel", dim3(nb), dim3(256));
    hipLaunchKernelGGL(met_recon_kernel, dim3(nb), dim3(256), 0, s, rec, tgt, n, scratch);
    hipLaunchKernelGGL(met_fold_kernel, dim3(1), dim3(256), 0, s, scratch, nb, 0, (double)n, out);
    return hipGetLastError();
}

hipError_t launch_metrics_flow(const float* flow, const float* gt, const float* img0, const float* img1, const float* valid,
                               int B, int H, int W, int backward, float max_flow, double* out, double* scratch, hipStream_t s) {
    if (!flow || !gt || !img0 || !img1 || !out || !scratch || B <= 0 || H < 2 || W < 2) return hipErrorInvalidValue;
    const long n = (long)B * H * W;
    const int nb = met_blocks(n);
    note_launch("met_flow_kernel", dim3(nb), dim3(256));
    hipLaunchKernelGGL(met_flow_kernel, dim3(nb), dim3(256), 0, s, flow, gt, img0, img1, valid, B, H, W, backward, max_flow, scratch);
    hipLaunchKernelGGL(met_fold_kernel, dim3(1), dim3(256), 0, s, scratch, nb, 1, (double)n, out);
    return hipGetLastError();
}

hipError_t launch_metrics_fwl(const float* voxel, const float* flow, int B, int C, int H, int W, double* out, double* scratch,
                              hipStream_t s) {
    if (!voxel || !flow || !out || !scratch || B <= 0 || C < 2 || H < 2 || W < 2) return hipErrorInvalidValue;
    const long n = (long)B * H * W;
    const int nb = met_blocks(n);
    note_launch("met_fwl_kernel", dim3(nb), dim3(256));
    hipLaunchKernelGGL(met_fwl_kernel, dim3(nb), dim3(256), 0, s, voxel, flow, B, C, H, W, scratch);
    hipLaunchKernelGGL(met_fold_kernel, dim3(1), dim3(256), 0, s, scratch, nb, 2, (double)n, out);
    return hipGetLastError();
}

hipError_t launch_metrics_ssim(const float* x, const float* y, int planes, int H, int W, double* out, double* scratch, hipStream_t s) {
    if (!x || !y || !out || !scratch || planes <= 0 || H < 11 || W < 11) return hipErrorInvalidValue;
    SsimWin win;
    float sum = 0.f;
    for (int k = 0; k < 11; ++k) {          // pytorch_msssim._fspecial_gauss_1d(11, 1.5) in fp32
        const float c = (float)(k - 5);
        win.g[k] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        sum += win.g[k];
    }
    for (int k = 0; k < 11; ++k) win.g[k] /= sum;
    const long tiles = (long)planes * ((H - 10 + SS_TY - 1) / SS_TY) * ((W - 10 + SS_TX - 1) / SS_TX);
    const int nb = (int)(tiles < MET_BLOCKS ? tiles : MET_BLOCKS);
    const float C1 = (0.01f * 1.f) * (0.01f * 1.f), C2 = (0.03f * 1.f) * (0.03f * 1.f);
    note_launch("met_ssim_kernel", dim3(nb), dim3(256));
    hipLaunchKernelGGL(met_ssim_kernel, dim3(nb), dim3(256), 0, s, x, y, planes, H, W, win, C1, C2, scratch);
    hipLaunchKernelGGL(met_fold_kernel, dim3(1), dim3(256), 0, s, scratch, nb, 3, (double)planes * (H - 10) * (W - 10), out);
    return hipGetLastError();
}

}  // namespace cf
