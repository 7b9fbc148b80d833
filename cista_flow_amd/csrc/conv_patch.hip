// conv_patch.hip -- convolutions over planar boundary tensors with a handful of input channels (round 3): the 7x7 / stride-2 stems
// of the three encoders (1 or 5 -> 64), We / Wi of CISTA-LSTC (5 / 1 -> 32, 3x3 reflect), the flow branch of the motion encoder
// (7x7, 2 -> 128) and IDNet's stem.  Launched through launch_conv (conv_igemm.hip) as tile 43.  gfx950 only.
//
// STATUS: parity-green (test_conv_gather_small_cin, test_conv_fused_inorm_stats) and OPT-IN (CF_PATCH=1 / explicit tile 43): it ties
// with conv_igemm_kernel's gather mode on the 7x7 stems and loses on We / Wi (see launch_conv), i.e. VERDICT r2's "-170 us" for the
// stems was not there to take -- the 5-channel stem's 3.1 GFLOP cost 24 us of fp32 MFMA time in any kernel.
//
// These layers have K = taps x Cin <= 256 and an A operand that must be COMPUTED per element (ImagePadder's zero pad on top / left,
// `2 I - 1`, `flow = coords1 - coords0`, reflect padding).  conv_igemm_kernel's gather mode does that per A element while staging --
// a table lookup, the padding arithmetic and a scalar buffer load for every (pixel, k) -- and runs at 7-46 TFLOP/s (MFMA 3-9 % busy).
// Here a workgroup resolves all of that ONCE for the input patch of its 8 x 16 output pixels (at most 5 x 21 x 37 floats, 15 KB of
// LDS) and then runs a plain MFMA loop whose A operand is two ds_read_b32 per step through a k -> patch-offset table:
//   workgroup   256 threads = 4 waves; 8 x 16 output pixels x NT*32 output channels; wave w owns tile rows 2w, 2w+1 (32 pixels)
//   K loop      super-groups of 16 k: lane (pixel m, half h) feeds k = 4g + 2h + {0, 1} to two v_mfma_f32_32x32x2_f32 per n-tile; the
//               [NT*32] x [16] weight block of a super-group is staged through LDS (coalesced 16-byte loads, double-buffered)
//   tail        per n-tile a 32-pixel x 32-channel patch through the common fused epilogue (patch_tail with a row -> pixel table);
//               InstanceNorm partials per (workgroup, wave) patch
#include "conv_common.h"

namespace cf {

static constexpr int CP_TH = 8, CP_TW = 16;
static constexpr int CP_PATCH_MAX = 5 * 21 * 37;            // floats of the largest input patch (5 channels, 7x7, stride 2)
static constexpr int CP_KMAX = 256;

__host__ __device__ inline int patch_floats(const ConvParams& p) {
    return p.g_cin * ((CP_TH - 1) * p.stride + p.KH) * ((CP_TW - 1) * p.stride + p.KW);
}

template <int NT>
__device__ __forceinline__ void conv_patch_body(const ConvParams& p, float* smem, int* ktab) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = p.Ho, Wo = p.Wo, s = p.stride;
    const int PH = (CP_TH - 1) * s + p.KH, PW = (CP_TW - 1) * s + p.KW;
    const int ntx = (Wo + CP_TW - 1) / CP_TW, nty = (Ho + CP_TH - 1) / CP_TH;
    const int ntile = ntx * nty;
    const int nt = (p.cout + 32 * NT - 1) / (32 * NT);
    int tile_id = blockIdx.x;
    if (p.sched == 1) {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = tile_id & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    const int nblk = tile_id % nt;
    const int rest = tile_id / nt;
    const int tl = rest % ntile;
    const int b = rest / ntile;
    const int oy0 = (tl / ntx) * CP_TH, ox0 = (tl % ntx) * CP_TW;
    const int n0 = nblk * 32 * NT;

    // ---- k -> patch offset table: k = tap * Cin + c (the packed matrix's K order), offset = (c * PH + ky) * PW + kx; k past the
    // real taps meet zero weights (pack_weight_kernel pads with zeros) and read offset 0 ----
    const int ntap = p.KH * p.KW;
    for (int k = tid; k < p.Ktot; k += 256) {
        const int tap = k / p.g_cin, c = k - tap * p.g_cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        ktab[k] = tap < ntap ? (c * PH + ky) * PW + kx : 0;
    }
    // ---- the input patch, every boundary rule resolved once: virtual input (Hin, Win) = source image shifted by (g_offy, g_offx)
    // (ImagePadder's zero pad on top / left), v' = g_scale v + g_shift inside it, minus the pixel grid for g_subgrid, conv padding
    // zero or reflect around the VIRTUAL input ----
    // Wave w takes patch rows w, w + 4, ... of the flattened (channel, row) list: everything about a row is wave-uniform (scalar
    // arithmetic), a lane is a column, and the loads of NB rows are in flight together -- the first version walked a flat index with
    // two divisions and one dependent load per element, and the 16 serial round trips of a 5-channel 7x7 patch were the kernel.
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.in[0] + (long)b * p.seg_bs[0]);
        const int iy0 = oy0 * s - p.padT, ix0 = ox0 * s - p.padL;
        const int nrow = p.g_cin * PH;
        int ix = ix0 + lane;
        bool okx = lane < PW;
        if (p.pad_mode == 1) {
            okx = okx && ix >= -p.padL && ix < p.Win + p.KW;
            ix = reflect_idx(ix, p.Win);
        }
        okx = okx && ix >= 0 && ix < p.Win;
        const int sx = ix - p.g_offx;
        okx = okx && sx >= 0;
        constexpr int NB = 8;
        for (int r0 = wave; r0 < nrow; r0 += 4 * NB) {
            float x[NB];
            bool ok[NB];
            int sy[NB], cc[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int rr = r0 + 4 * u;                      // wave-uniform
                const int c = rr / PH, py = rr - c * PH;
                int iy = iy0 + py;
                bool oky = rr < nrow;
                if (p.pad_mode == 1) {
                    oky = oky && iy >= -p.padT && iy < p.Hin + p.KH;
                    iy = reflect_idx(iy, p.Hin);
                }
                oky = oky && iy >= 0 && iy < p.Hin;
                sy[u] = iy - p.g_offy;
                cc[u] = c;
                ok[u] = oky && sy[u] >= 0 && okx;
                x[u] = buf_load1(rs, ok[u] ? (unsigned)((c * p.Hsrc + sy[u]) * p.Wsrc + sx) * 4u : BUF_OOB, 0u);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int rr = r0 + 4 * u;
                float v = x[u] * p.g_scale + p.g_shift;
                if (p.g_subgrid) v -= (cc[u] == 0) ? (float)sx : (float)sy[u];
                if (rr < nrow && lane < PW) smem[rr * PW + lane] = ok[u] ? v : 0.f;
            }
        }
    }
    const int lr = lane & 31, lh = lane >> 5;
    const int ty = 2 * wave + (lr >> 4), tx = lr & 15;
    const int mbase = (ty * s) * PW + tx * s;
    // B: the [NT*32 rows] x [16 k] block of a super-group is staged through LDS (16-byte loads, four lanes per row = 64 contiguous
    // bytes; register-staged, double-buffered, one barrier per super-group).  The first version let every lane fetch its own 8-byte
    // fragments: 32 different cache lines per load instruction, the same 64 KB of weights pulled through every wave's L1 path --
    // the texture-address unit, not the matrix pipe, set the time (68 us for the 5-channel stem, the same as the gather kernel).
    constexpr int BROW = 18;                                    // floats per staged row (16 + 2: conflict-free 8-byte fragment reads)
    constexpr int BL = (NT * 32 * 4 + 255) / 256;               // 16-byte loads per thread and super-group
    float* const sB = smem + CP_PATCH_MAX;                      // [2][NT*32][BROW]
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w + (long)wgroup(p, b) * p.w_bs);
    unsigned boff[BL];
    int bdst[BL];
#pragma unroll
    for (int u = 0; u < BL; ++u) {
        const int sl = tid + 256 * u;
        const int row = sl >> 2, qd = sl & 3;
        const bool live = row < NT * 32;
        boff[u] = (live && n0 + row < p.w_rows) ? ((unsigned)(n0 + row) * (unsigned)p.Ktot + 4u * qd) * 4u : BUF_OOB;
        bdst[u] = live ? row * BROW + 4 * qd : -1;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int nsg = p.Ktot >> 4;
    f32x4 breg[BL];
#pragma unroll
    for (int u = 0; u < BL; ++u) breg[u] = buf_load4(w_rsrc, boff[u], 0u);
#pragma unroll
    for (int u = 0; u < BL; ++u)
        if (bdst[u] >= 0) {
            float* d = sB + bdst[u];
            d[0] = breg[u][0]; d[1] = breg[u][1]; d[2] = breg[u][2]; d[3] = breg[u][3];
        }
    __syncthreads();                                // patch, k table and B(0) are in place
    for (int G = 0; G < nsg; ++G) {
        const bool more = G + 1 < nsg;
        if (more) {
#pragma unroll
            for (int u = 0; u < BL; ++u) breg[u] = buf_load4(w_rsrc, boff[u], (unsigned)(G + 1) * 64u);
        }
        const float* bb = sB + (G & 1) * (NT * 32 * BROW);
        int2 ko[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ko[q] = *reinterpret_cast<const int2*>(ktab + 16 * G + 4 * q + 2 * lh);
        float a0[4], a1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0[q] = smem[mbase + ko[q].x];
            a1[q] = smem[mbase + ko[q].y];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float* bf = bb + (32 * j + lr) * BROW + 4 * q + 2 * lh;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], bf[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], bf[1], acc[j], 0, 0, 0);
            }
        if (more) {
            float* nb = sB + ((G + 1) & 1) * (NT * 32 * BROW);
#pragma unroll
            for (int u = 0; u < BL; ++u)
                if (bdst[u] >= 0) {
                    float* d = nb + bdst[u];
                    d[0] = breg[u][0]; d[1] = breg[u][1]; d[2] = breg[u][2]; d[3] = breg[u][3];
                }
        }
        __syncthreads();
    }

    // (the loop's last barrier: everybody is done with the input patch and B: the epilogue patches go on top of them)
    float* const sW = smem + wave * (32 * EPI_S);
    int* const mtab = ktab + wave * 32;             // the k table is dead as well
    if (lane < 32) {
        const int oy = oy0 + 2 * wave + (lane >> 4), ox = ox0 + (lane & 15);
        mtab[lane] = (oy < Ho && ox < Wo) ? oy * Wo + ox : -1;
    }
#pragma unroll 1
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = 0.f;
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) v = jj == j ? acc[jj][r] : v;
            sW[row * EPI_S + lr] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (n0 + 32 * j < p.cout) {
            patch_tail(p, sW, b, 0, n0 + 32 * j, lane, Ho * Wo, 0, 4, 1, 0, mtab);
            if (p.st_partial) patch_stats(p, sW, b, 0, n0 + 32 * j, lane, Ho * Wo, mtab, tl * 4 + wave, ntile * 4);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// floats: input patch + the double-buffered B blocks (4 n-tiles x 32 rows x 18), later the four epilogue patches on top
static constexpr int CP_SMEM = CP_PATCH_MAX + 2 * 4 * 32 * 18;
static_assert(CP_SMEM >= 4 * 32 * EPI_S, "epilogue patches overlay the loop's buffers");

template <int NT>
__global__ __launch_bounds__(256, NT <= 2 ? 4 : 3) void conv_patch_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[CP_SMEM + CP_KMAX];      // ONE __shared__ object: see conv_wino_kernel
    conv_patch_body<NT>(p, smem, reinterpret_cast<int*>(smem + CP_SMEM));
}

int patch_tiles(int Ho, int Wo) { return ((Ho + CP_TH - 1) / CP_TH) * ((Wo + CP_TW - 1) / CP_TW); }

bool patch_ok(const ConvParams& p) {
    if (p.a_mode != A_GATHER || p.prec != 0 || p.w_bs != 0) return false;
    if (p.Ktot > CP_KMAX || (p.Ktot & 15) || p.KH * p.KW * p.g_cin > p.Ktot) return false;
    if (p.stride < 1 || p.stride > 2 || p.g_cin < 1) return false;
    if (patch_floats(p) > CP_PATCH_MAX) return false;
    if ((long)p.g_cin * p.Hsrc * p.Wsrc * 4L >= 0x7FFFFF00L) return false;        // 32-bit offsets inside one image
    return true;
}

hipError_t launch_patch(const ConvParams& p, int batch, hipStream_t s) {
    if (!patch_ok(p)) return hipErrorInvalidValue;
    const int NT = p.cout <= 32 ? 1 : (p.cout <= 64 ? 2 : 4);
    const long wgs = (long)patch_tiles(p.Ho, p.Wo) * ((p.cout + 32 * NT - 1) / (32 * NT)) * batch;
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256;
    if (NT == 1) hipLaunchKernelGGL(conv_patch_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, p);
    else if (NT == 2) hipLaunchKernelGGL(conv_patch_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_patch_kernel<4>, dim3((unsigned)wgs), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
