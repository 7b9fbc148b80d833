// conv_igemm.hip -- implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// products and accumulation, so results differ from PyTorch only by summation order).
//
// One workgroup (4 waves, WAVES_M x WAVES_N x WK) owns a BM x BN tile of  out[pixel][cout]  for one image; K runs
// over (tap, cin) in stages of 16*WK (or 32*WK) columns.  Two kernels share the tail and the tile scheduler:
//   conv_dma_kernel    plain NHWC reads in fp32 (almost all of the flops): A / B stage tiles go straight into LDS by
//                      buffer_load ... lds into a swizzled 3-deep ring, counted vmcnt, DMA pieces issued between the
//                      MFMA groups (see the comment block in front of the kernel);
//   conv_igemm_kernel  A operands that are computed while staging (fused bilinear x2 read, planar tiny-Cin gather,
//                      fp32 activations as B, the f16 / f16x3 operand split): global -> registers -> LDS, double
//                      buffered, rows 80 B apart so ds_read_b128 is conflict free.
// A lane reads its row's k = 4h..4h+3 (h = lane>>5) of an 8-deep k group with one ds_read_b128; MFMA step s of a
// group therefore multiplies k = 8g + 4h + s on both operands -- a permutation of k, which a dot product does not see.
// The tail (bias, activations, ISTA / LSTC / LSTM / GRU updates, InstanceNorm statistics) is fused: patch_tail*.
//
// Reference semantics covered (each cited where it is used in cf_api.hip):
//   nn.Conv2d(padding_mode='reflect'|'zeros', stride 1|2, kernels 1x1,3x3,7x7,1x5,5x1),
//   F.interpolate(x2, bilinear, align_corners=False)+ReflectionPad2d fused into the A read
//   (e2v/base_layers.py:195-212), ImagePadder zero pad (utils/image_process.py:87-101) and
//   `2*image-1` (DCEIFlow/DCEIFlow.py:146) fused into the small-Cin gather read.
#include "conv_common.h"

namespace cf {

thread_local LaunchInfo g_last_launch = {"", 0};
long g_wino4_min = 0;
long wino16_max() {
    static const long v = getenv("CF_WINO16_MAX") ? atol(getenv("CF_WINO16_MAX")) : 640;
    return v;
}

// waves per SIMD the register allocation must leave room for (see conv_dma_kernel): what the main loop needs
// (5 / 4 / 3 / 3 for 1 / 2 / 3 / 4 accumulator sub-tiles), capped by the LDS footprint of the two stage buffers
constexpr int igemm_waves(int BM, int BN, int WM, int WN, int WK, int KCW) {
    const int t = (BM / (32 * WM)) * (BN / (32 * WN));
    const int by_regs = t == 1 ? 5 : (t == 2 ? 4 : 3);
    const int stage2 = 2 * (BM + BN) * (KCW * WK + 4);
    const int tail = (WK - 1) * WM * WN * t * 1024 + WM * WN * 32 * 36;
    const int by_lds = (160 * 1024) / (4 * (stage2 > tail ? stage2 : tail) + 1024);
    return by_regs < by_lds ? by_regs : by_lds;
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int WK, int AMODE, int PREC, int KCW = 16>
__global__ __launch_bounds__(256, igemm_waves(BM, BN, WAVES_M, WAVES_N, WK, KCW)) void conv_igemm_kernel(const ConvParams p) {
    static_assert(KCW == 16 || KCW == 32, "KCW");
    static_assert(KCW == 16 || AMODE == A_NHWC, "wide stages only for the plain NHWC read");
    static_assert(WAVES_M * WAVES_N * WK == 4, "4 waves per workgroup");
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tiles are multiples of the 32x32 MFMA");
    static_assert(AMODE == A_NHWC || WK == 1, "split-K tiles only for the plain NHWC read");
    constexpr int KS = KCW * WK;         // k-columns per stage
    constexpr int QPR = KS / 4;          // 16-byte quads per row
    constexpr int LS = KS + 4;           // LDS row stride (floats): 16-byte slot (LS/4)*r mod 16 is a permutation
    constexpr int WMN = WAVES_M * WAVES_N;
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int A_IT = (BM * QPR + 255) / 256;
    constexpr int B_IT = (BN * QPR + 255) / 256;
    constexpr int STAGE = (BM + BN) * LS;
    constexpr int GTAB = (AMODE == A_GATHER) ? 256 : 4;
    constexpr bool A_FULL = (BM * QPR) % 256 == 0;   // every staging slot of every thread is inside the tile
    constexpr bool B_FULL = (BN * QPR) % 256 == 0;
    constexpr int RED = (WK - 1) * WMN * TM * TN * 1024;            // split-K reduction area (floats)
    constexpr int SMEM = (2 * STAGE > RED + WMN * 32 * EPI_S) ? 2 * STAGE : RED + WMN * 32 * EPI_S;

    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    __shared__ __attribute__((aligned(16))) int gtab[GTAB];   // A_GATHER: k -> (ky | kx<<8 | c<<16), -1 = pad

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / WMN;
    const int wmn = wave % WMN;
    const int wm = wmn / WAVES_N;
    const int wn = wmn % WAVES_N;
    const int M = p.Ho * p.Wo;
    // ---- workgroup -> tile map.  1-D grid; hardware deals consecutive block ids round-robin over the 8 XCDs
    // (ids b and b+8 share an XCD and its L2).  sched 1 hands every XCD one contiguous run of logical tiles
    // ordered (image, m-tile, n-tile): the n-tiles of an m-tile (same A rows) and vertically adjacent m-tiles
    // (shared halo rows) then hit in the same L2 instead of being re-fetched through the fabric.  Speed only --
    // any placement computes the same tiles.
    const int mt = (M + BM - 1) / BM;
    const int nt = (p.cout + BN - 1) / BN;
    int tile_id = blockIdx.x;
    if (p.sched == 1) {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = tile_id & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    int b, m0, n0;
    if (p.sched == 1) {
        const int ny = tile_id % nt;
        const int rest = tile_id / nt;
        n0 = ny * BN;
        m0 = (rest % mt) * BM;
        b = rest / mt;
    } else {
        m0 = (tile_id % mt) * BM;
        const int rest = tile_id / mt;
        n0 = (rest % nt) * BN;
        b = rest / nt;
    }

    // per-thread A slots (row, quad) -- fixed for the whole K loop
    int a_oy[A_IT], a_ox[A_IT], a_row[A_IT], a_q[A_IT];
    bool a_ok[A_IT], a_live[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int slot = tid + 256 * j;
        const int row = slot / QPR;
        a_row[j] = row;
        a_q[j] = slot - row * QPR;
        a_live[j] = row < BM;
        const int m = m0 + row;
        a_ok[j] = a_live[j] && m < M;
        const int oy = m / p.Wo;
        a_oy[j] = oy * p.stride - p.padT;
        a_ox[j] = (m - oy * p.Wo) * p.stride - p.padL;
    }
    // per-thread B slots: byte offset of (row, quad) inside the packed matrix, BUF_OOB past its last row
    unsigned b_off[B_IT];
    bool b_live[B_IT];
    int b_row[B_IT], b_q[B_IT];
    const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(p.w + (long)wgroup(p, b) * p.w_bs);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int slot = tid + 256 * it;
        const int row = slot / QPR;
        b_row[it] = row;
        b_q[it] = slot - row * QPR;
        b_live[it] = row < BN;
        const bool ok = b_live[it] && (n0 + row) < p.w_rows;
        b_off[it] = ok ? (unsigned)(n0 + row) * (unsigned)p.Ktot * 4u + (unsigned)b_q[it] * 16u : BUF_OOB;
    }
    if (AMODE == A_GATHER) {
        const int ntap = p.KH * p.KW;
        for (int k = tid; k < GTAB; k += 256) {
            const int tap = k / p.g_cin;
            const int c = k - tap * p.g_cin;
            const int ky = tap / p.KW;
            gtab[k] = (tap < ntap && k < p.Ktot) ? (ky | ((tap - ky * p.KW) << 8) | (c << 16)) : -1;
        }
        __syncthreads();
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nck = p.Ktot / KS;

    // ---- chunk iterator (wave-uniform): tap (ky,kx), channel segment, channel offset inside it ----
    int it_ky = 0, it_kx = 0, it_seg = 0, it_cs = 0, it_k = 0;
    const int nseg = p.nseg, kw_n = p.KW;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];

    // per-tap, per-slot source pixel (element index into the NHWC plane, -1 = zero padding / out of range)
    int a_pix[A_IT];
    // A_UPS2X: the four bilinear taps and weights of each row
    int u_p01[A_IT], u_p10[A_IT], u_p11[A_IT];
    float u_ly1[A_IT], u_lx1[A_IT];

    const int g_hin = p.Hin, g_win = p.Win, g_padm = p.pad_mode;
    auto tap_setup = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int iy = a_oy[j] + it_ky;
            int ix = a_ox[j] + it_kx;
            bool ok = a_ok[j];
            if (g_padm == 1) {
                iy = reflect_idx(iy, g_hin);
                ix = reflect_idx(ix, g_win);
            } else {
                ok = ok && iy >= 0 && iy < g_hin && ix >= 0 && ix < g_win;
            }
            if (AMODE == A_NHWC) {
                a_pix[j] = ok ? iy * g_win + ix : -1;
            } else {   // A_UPS2X: bilinear x2 (align_corners=False) source taps of the (Hsrc,Wsrc) plane
                float sy = ((float)iy + 0.5f) * 0.5f - 0.5f;
                float sx = ((float)ix + 0.5f) * 0.5f - 0.5f;
                sy = sy < 0.f ? 0.f : sy;
                sx = sx < 0.f ? 0.f : sx;
                const int y0 = (int)sy, x0 = (int)sx;
                const int y1 = y0 + (y0 < p.Hsrc - 1 ? 1 : 0);
                const int x1 = x0 + (x0 < p.Wsrc - 1 ? 1 : 0);
                u_ly1[j] = sy - (float)y0;
                u_lx1[j] = sx - (float)x0;
                a_pix[j] = ok ? y0 * p.Wsrc + x0 : -1;
                u_p01[j] = y0 * p.Wsrc + x1;
                u_p10[j] = y1 * p.Wsrc + x0;
                u_p11[j] = y1 * p.Wsrc + x1;
            }
        }
    };
    if (AMODE != A_GATHER) tap_setup();

    f32x4 a_reg[A_IT], b_reg[B_IT];

    auto load_chunk = [&]() __attribute__((always_inline)) {
        // ---- B: packed weights, K contiguous ----
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            // unconditional load (rows past the matrix were redirected to row 0) + select: a branch around the
            // load would serialise the wave and drain vmcnt per element
            b_reg[it] = buf_load4(b_rsrc, b_off[it], (unsigned)it_k * 4u);
        }
        // ---- A ----
        if (AMODE == A_GATHER) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                const int4 tb = *reinterpret_cast<const int4*>(&gtab[it_k + a_q[j] * 4]);
                const int te[4] = {tb.x, tb.y, tb.z, tb.w};
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = te[e];
                    const int ky = t & 0xff, kx = (t >> 8) & 0xff, c = t >> 16;
                    int iy = a_oy[j] + ky;
                    int ix = a_ox[j] + kx;
                    bool ok = a_ok[j] && t >= 0;
                    if (p.pad_mode == 1) {
                        iy = reflect_idx(iy, p.Hin);
                        ix = reflect_idx(ix, p.Win);
                    } else {
                        ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                    }
                    const int sy = iy - p.g_offy;
                    const int sx = ix - p.g_offx;
                    ok = ok && sy >= 0 && sx >= 0;
                    float x = buf_load1(rs, ok ? (unsigned)((c * p.Hsrc + sy) * p.Wsrc + sx) * 4u : BUF_OOB, 0u);
                    x = x * p.g_scale + p.g_shift;
                    if (p.g_subgrid) x -= (c == 0) ? (float)sx : (float)sy;
                    v[e] = ok ? x : 0.f;
                }
                a_reg[j] = v;
            }
        } else if (AMODE == A_NHWC) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
            const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                // padding taps / rows past the image use an out-of-range offset: the load returns zeros
                const unsigned off = a_pix[j] < 0 ? BUF_OOB : (unsigned)a_pix[j] * ld4 + (unsigned)a_q[j] * 16u;
                a_reg[j] = buf_load4(rs, off, so);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
            const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                const bool inv = a_pix[j] < 0;          // all four taps out of range -> zeros -> lerp gives 0
                const unsigned qo = (unsigned)a_q[j] * 16u;
                const f32x4 v00 = buf_load4(rs, inv ? BUF_OOB : (unsigned)a_pix[j] * ld4 + qo, so);
                const f32x4 v01 = buf_load4(rs, inv ? BUF_OOB : (unsigned)u_p01[j] * ld4 + qo, so);
                const f32x4 v10 = buf_load4(rs, inv ? BUF_OOB : (unsigned)u_p10[j] * ld4 + qo, so);
                const f32x4 v11 = buf_load4(rs, inv ? BUF_OOB : (unsigned)u_p11[j] * ld4 + qo, so);
                const float ly1 = u_ly1[j], lx1 = u_lx1[j];
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    v[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
                a_reg[j] = v;
            }
        }
        // ---- advance the iterator to the next chunk ----
        it_k += KS;
        if (AMODE != A_GATHER) {
            it_cs += KS;
            if (it_cs >= seg_cn) {
                it_cs = 0;
                ++it_seg;
                if (it_seg >= nseg) {
                    it_seg = 0;
                    ++it_kx;
                    if (it_kx >= kw_n) {
                        it_kx = 0;
                        ++it_ky;
                    }
                    tap_setup();
                }
                // (keeping the three descriptors in extra locals instead of re-reading kernarg memory here makes
                // hipcc spill the whole loop state to scratch -- measured, ROCm 7.2 -- and a segment switch only
                // happens every few stages)
                seg_base = sel3(p.in, it_seg) + (long)b * (it_seg == 0 ? p.seg_bs[0] : (it_seg == 1 ? p.seg_bs[1] : p.seg_bs[2]));
                seg_ld = it_seg == 0 ? p.seg_ld[0] : (it_seg == 1 ? p.seg_ld[1] : p.seg_ld[2]);
                seg_cn = it_seg == 0 ? p.seg_c[0] : (it_seg == 1 ? p.seg_c[1] : p.seg_c[2]);
            }
        }
    };

    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + BM * LS;
        if (PREC == 0) {
#pragma unroll
            for (int j = 0; j < A_IT; ++j)
                if (A_FULL || a_live[j]) *reinterpret_cast<f32x4*>(sA + a_row[j] * LS + a_q[j] * 4) = a_reg[j];
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                if (B_FULL || b_live[it]) *reinterpret_cast<f32x4*>(sB + b_row[it] * LS + b_q[it] * 4) = b_reg[it];
        } else {
            // quad q of a row = k 4q..4q+3 -> chunk q/4: hi at chunk*64 + (q%4)*8 bytes, lo 32 bytes further
#pragma unroll
            for (int j = 0; j < A_IT; ++j) {
                if (!A_FULL && !a_live[j]) continue;
                f16x4 hi, lo;
                split_f16(a_reg[j], hi, lo);
                char* dst = reinterpret_cast<char*>(sA + a_row[j] * LS) + (a_q[j] >> 2) * 64 + (a_q[j] & 3) * 8;
                *reinterpret_cast<f16x4*>(dst) = hi;
                if (PREC == 3) *reinterpret_cast<f16x4*>(dst + 32) = lo;
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                if (!B_FULL && !b_live[it]) continue;
                if (p.b_f32) {   // B operand is an activation (all-pairs correlation): split it here as well
                    f16x4 hi, lo;
                    split_f16(b_reg[it], hi, lo);
                    char* dst = reinterpret_cast<char*>(sB + b_row[it] * LS) + (b_q[it] >> 2) * 64 + (b_q[it] & 3) * 8;
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    if (PREC == 3) *reinterpret_cast<f16x4*>(dst + 32) = lo;
                } else {         // weights were split at pack time: HBM rows already have the LDS format
                    *reinterpret_cast<f32x4*>(sB + b_row[it] * LS + b_q[it] * 4) = b_reg[it];
                }
            }
        }
    };

    load_chunk();
    store_chunk(0);
    __syncthreads();

    const int lr = lane & 31;
    const int lh = lane >> 5;
    for (int ck = 0; ck < nck; ++ck) {
        const int buf = ck & 1;
        if (ck + 1 < nck) load_chunk();
        const float* sA = smem + buf * STAGE + wk * KCW;
        const float* sB = smem + buf * STAGE + BM * LS + wk * KCW;
        if (PREC != 0) {
#pragma unroll
          for (int cc = 0; cc < KCW / 16; ++cc) {
            // one v_mfma_f32_32x32x16_f16 spans a whole 16-column chunk: lane (r, h) holds k = 8h..8h+7
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char* r = reinterpret_cast<const char*>(sA + ((wm * TM + i) * 32 + lr) * LS) + cc * 64 + lh * 16;
                ah[i] = *reinterpret_cast<const f16x8*>(r);
                if (PREC == 3) al[i] = *reinterpret_cast<const f16x8*>(r + 32);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const char* r = reinterpret_cast<const char*>(sB + ((wn * TN + j) * 32 + lr) * LS) + cc * 64 + lh * 16;
                bh[j] = *reinterpret_cast<const f16x8*>(r);
                if (PREC == 3) bl[j] = *reinterpret_cast<const f16x8*>(r + 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (PREC == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
          }
        } else
#pragma unroll
        for (int ks = 0; ks < KCW / 8; ++ks) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(sA + ((wm * TM + i) * 32 + lr) * LS + ks * 8 + lh * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(sB + ((wn * TN + j) * 32 + lr) * LS + ks * 8 + lh * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (ck + 1 < nck) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- split-K: fold the partial accumulators of the wk > 0 waves into the wk == 0 wave ----
    if (WK > 1 && !p.st_partial) {
        // Split-K tail shared by all waves: every wave parks its partial 32x32 tile in patch layout, and after the barrier
        // finishes 4/WK of the patch's row groups (summing the WK partials on the way, in the order k = 0..WK-1).
        // Before, the wk > 0 waves left and one wave per sub-tile did the whole tail.
        static_assert(WK == 1 || (TM == 1 && TN == 1), "split-K tiles hold one sub-tile per wave");
        static_assert(WK * WMN * 32 * EPI_S <= SMEM, "partial patches must fit the stage buffers");
        constexpr int PSTRIDE = WMN * 32 * EPI_S;
        float* part = smem + wk * PSTRIDE + wmn * (32 * EPI_S);
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_S + lr] = acc[0][0][r];
        __syncthreads();
        patch_tail(p, smem + wmn * (32 * EPI_S), b, m0 + wm * 32, n0 + wn * 32, lane, M, wk * (4 / WK), (wk + 1) * (4 / WK), WK, PSTRIDE);
        return;
    }
    if (WK > 1) {
        float* red = smem;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((wk - 1) * WMN + wmn) * TM + i) * TN + j) * 1024 + r * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int k = 1; k < WK; ++k)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += red[((((k - 1) * WMN + wmn) * TM + i) * TN + j) * 1024 + r * 64 + lane];
    }

    // ---- epilogue ----
    // C/D map of the 32x32 MFMA: col = lane&31 (cout), row = (r&3)+8*(r>>2)+4*(lane>>5).  Each
    // 32x32 sub-tile goes through a per-wave LDS patch so that a lane ends up with 4 consecutive
    // couts of one pixel: aux reads and the store are then 16-byte accesses on 128-byte rows.
    float* sW = smem + RED + wmn * (32 * EPI_S);
    // One run-time loop over the wave's TM x TN sub-tiles: the (large) tail code exists once, and the compiler
    // cannot hoist every sub-tile's loads / address arithmetic to the top (which cost 60-130 VGPRs when the loop
    // was unrolled).  Only the copy of the accumulators into the patch is selected per sub-tile.
#pragma unroll 1
    for (int pi = 0; pi < TM * TN; ++pi) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (pi == i * TN + j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sW[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_S + lr] = acc[i][j][r];
                }
        const int pi_i = pi / TN, pi_j = pi - pi_i * TN;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        patch_tail(p, sW, b, m0 + (wm * TM + pi_i) * 32, n0 + (wn * TN + pi_j) * 32, lane, M);
        if (p.st_partial) patch_stats(p, sW, b, m0 + (wm * TM + pi_i) * 32, n0 + (wn * TN + pi_j) * 32, lane, M);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int BM, int BN, int WM, int WN, int WK, int PREC, int KCW = 16>
static hipError_t launch_tp(const ConvParams& p, int batch, hipStream_t s) {
    const int M = p.Ho * p.Wo;
    dim3 grid(((M + BM - 1) / BM) * ((p.cout + BN - 1) / BN) * batch);
    g_last_launch.threads = (long)grid.x * 256;
    if (p.a_mode == A_NHWC) {
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, WK, A_NHWC, PREC, KCW>), grid, dim3(256), 0, s, p);
    } else if constexpr (WK == 1 && KCW == 16) {
        if (p.a_mode == A_UPS2X) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, A_UPS2X, PREC>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, A_GATHER, PREC>), grid, dim3(256), 0, s, p);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int WK, int KCW = 16>
static hipError_t launch_t(const ConvParams& p, int batch, hipStream_t s) {
    if (p.prec == 3) return launch_tp<BM, BN, WM, WN, WK, 3, KCW>(p, batch, s);
    if (p.prec == 1) return launch_tp<BM, BN, WM, WN, WK, 1, KCW>(p, batch, s);
    return launch_tp<BM, BN, WM, WN, WK, 0, KCW>(p, batch, s);
}

// stages wider than 16 columns need every channel segment to be a whole number of stages
static bool stage_ok(const ConvParams& p, int ks) {
    if (p.a_mode != A_NHWC) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg_c[i] % ks) return false;
    return true;
}
static bool splitk_ok(const ConvParams& p, int wk) { return stage_ok(p, KC * wk); }

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant (plain NHWC read, fp32 MFMA): the A / B stage tiles are written straight into LDS by
// `buffer_load_dwordx4 ... lds` -- no staging registers, no ds_write pass, no per-stage vmcnt(0) -- through a
// 3-deep ring with a COUNTED vmcnt (the DMA of stages k+1 / k+2 stays in flight across the one barrier per
// stage).  A wave-instruction lands 64 x 16 bytes LINEARLY in LDS (lane i -> base + 16 i), so rows are unpadded
// and bank conflicts are avoided by an XOR swizzle applied on the SOURCE side (which 16-byte quad of its row a
// lane fetches) and again when the fragments are read: slot(row, q) = row*QPR + (q ^ (row / RPB) % QPR), with
// RPB = rows per 256-byte bank row.  Out-of-range offsets (zero padding, rows past the tile) make the DMA write
// zeros.  Hand-off rule per stage k: wait vmcnt(loads of the later stages) -> s_barrier -> ds_read + MFMA on stage k,
// with the DMA pieces of stage k+2 (into the buffer the barrier just freed) issued between those MFMA groups.
// ---------------------------------------------------------------------------------------------------------
// Second launch bound = waves per SIMD the register allocation must leave room for: 5 / 4 / 3 for 1 / 2 / 4
// accumulator sub-tiles per wave, capped by what the LDS ring allows anyway (one wave per SIMD per workgroup).
// Without it hipcc lets the tail code take registers the main loop does not need and halves the occupancy.
constexpr int dma_stage_floats(int BM, int BN, int WK, int KCW) { return (BM + BN) * KCW * WK; }
constexpr int dma_smem_floats(int BM, int BN, int WM, int WN, int WK, int KCW, int NBUF) {
    const int ring = NBUF * dma_stage_floats(BM, BN, WK, KCW);
    const int wmn = WM * WN, tm = BM / (32 * WM), tn = BN / (32 * WN);
    const int tail = (WK - 1) * wmn * tm * tn * 1024 + wmn * 32 * EPI_S;
    return ring > tail ? ring : tail;
}
constexpr int dma_waves(int BM, int BN, int WM, int WN, int WK, int KCW, int NBUF) {
    const int t = (BM / (32 * WM)) * (BN / (32 * WN));
    const int by_regs = t == 1 ? 5 : (t == 2 ? WPE2 : 3);
    const int by_lds = (160 * 1024) / (4 * dma_smem_floats(BM, BN, WM, WN, WK, KCW, NBUF));
    return by_regs < by_lds ? by_regs : by_lds;
}
// NBUF: ring depth; NBUF - 1 stages are in flight ahead of the one being multiplied (3: default; 4: the small tiles,
// whose ~12 KB stages otherwise leave too few bytes in flight per CU to cover the L2 latency at their load rate)
template <int BM, int BN, int WAVES_M, int WAVES_N, int WK, int KCW, int NBUF = 3>
__global__ __launch_bounds__(256, dma_waves(BM, BN, WAVES_M, WAVES_N, WK, KCW, NBUF))
void conv_dma_kernel(const ConvParams p) {
    static_assert(WAVES_M * WAVES_N * WK == 4, "4 waves per workgroup");
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();     // 100 MHz wall clock
    long long st_wait = 0, st_bar = 0, st_issue = 0;
#endif
    constexpr int KS = KCW * WK;
    constexpr int QPR = KS / 4;                        // 16-byte quads per row
    constexpr int RPB = (16 / QPR) > 0 ? (16 / QPR) : 1;   // rows per 256-byte bank row
    constexpr int WMN = WAVES_M * WAVES_N;
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int A_SLOTS = BM * QPR, B_SLOTS = BN * QPR;
    static_assert(A_SLOTS % 64 == 0 && B_SLOTS % 64 == 0, "whole wave-instructions");
    constexpr int A_IT = (A_SLOTS + 255) / 256;
    constexpr int B_IT = (B_SLOTS + 255) / 256;
    constexpr int STAGE = (BM + BN) * KS;              // floats per ring slot
    constexpr int RED = (WK - 1) * WMN * TM * TN * 1024;
    constexpr int SMEM = (NBUF * STAGE > RED + WMN * 32 * EPI_S) ? NBUF * STAGE : RED + WMN * 32 * EPI_S;

    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / WMN;
    const int wmn = wave % WMN;
    const int wm = wmn / WAVES_N;
    const int wn = wmn % WAVES_N;
    const int M = p.Ho * p.Wo;
    const int mt = (M + BM - 1) / BM;
    const int nt = (p.cout + BN - 1) / BN;
    int tile_id = blockIdx.x;
    if (p.sched == 1) {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = tile_id & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    int b, m0, n0;
    if (p.sched == 1) {
        const int ny = tile_id % nt;
        const int rest = tile_id / nt;
        n0 = ny * BN;
        m0 = (rest % mt) * BM;
        b = rest / mt;
    } else {
        m0 = (tile_id % mt) * BM;
        const int rest = tile_id / mt;
        n0 = (rest % nt) * BN;
        b = rest / nt;
    }

    // ---- per-thread DMA slots: slot -> (row, physical quad); the lane fetches logical quad qp ^ swz(row) ----
    int a_oy[A_IT], a_ox[A_IT];
    unsigned a_qoff[A_IT];          // logical quad * 16 bytes
    bool a_ok[A_IT], a_use[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int slot = tid + 256 * j;
        const int row = slot / QPR;
        const int qp = slot - row * QPR;
        a_use[j] = slot < A_SLOTS;
        a_qoff[j] = (unsigned)(qp ^ ((row / RPB) & (QPR - 1))) * 16u;
        const int m = m0 + row;
        a_ok[j] = a_use[j] && m < M;
        const int oy = m / p.Wo;
        a_oy[j] = oy * p.stride - p.padT;
        a_ox[j] = (m - oy * p.Wo) * p.stride - p.padL;
    }
    unsigned b_off[B_IT];
    bool b_use[B_IT];
    const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(p.w + (long)wgroup(p, b) * p.w_bs);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int slot = tid + 256 * it;
        const int row = slot / QPR;
        const int qp = slot - row * QPR;
        b_use[it] = slot < B_SLOTS;
        const bool ok = b_use[it] && (n0 + row) < p.w_rows;
        b_off[it] = ok ? (unsigned)(n0 + row) * (unsigned)p.Ktot * 4u + (unsigned)(qp ^ ((row / RPB) & (QPR - 1))) * 16u : BUF_OOB;
    }
    // ---- fragment read addresses (bytes inside a ring slot), fixed for the whole loop ----
    const int lr = lane & 31, lh = lane >> 5;
    unsigned fa[TM][KCW / 8], fb[TN][KCW / 8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int R = (wm * TM + i) * 32 + lr;
#pragma unroll
        for (int ks = 0; ks < KCW / 8; ++ks) {
            const int q = wk * (KCW / 4) + ks * 2 + lh;
            fa[i][ks] = (unsigned)(R * QPR + (q ^ ((R / RPB) & (QPR - 1)))) * 16u;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int R = (wn * TN + j) * 32 + lr;
#pragma unroll
        for (int ks = 0; ks < KCW / 8; ++ks) {
            const int q = wk * (KCW / 4) + ks * 2 + lh;
            fb[j][ks] = (unsigned)(BM * KS * 4) + (unsigned)(R * QPR + (q ^ ((R / RPB) & (QPR - 1)))) * 16u;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nck = p.Ktot / KS;
    int it_ky = 0, it_kx = 0, it_seg = 0, it_cs = 0, it_k = 0;
    const int nseg = p.nseg, kw_n = p.KW;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    const int g_hin = p.Hin, g_win = p.Win, g_padm = p.pad_mode;
    int a_pix[A_IT];
    auto tap_setup = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int iy = a_oy[j] + it_ky;
            int ix = a_ox[j] + it_kx;
            bool ok = a_ok[j];
            if (g_padm == 1) {
                iy = reflect_idx(iy, g_hin);
                ix = reflect_idx(ix, g_win);
            } else {
                ok = ok && iy >= 0 && iy < g_hin && ix >= 0 && ix < g_win;
            }
            a_pix[j] = ok ? iy * g_win + ix : -1;
        }
    };
    tap_setup();

    // one stage: A_IT + B_IT wave-instructions per wave (a wave whose 64 slots lie past the tile issues nothing)
    // A stage is fetched in NP = A_IT + B_IT pieces (one wave-instruction each); issue_piece(buf, q) issues piece q and
    // the last piece also advances the (wave-uniform) chunk iterator.  In the main loop the pieces are spread between
    // the MFMA groups of the stage being multiplied: a DMA piece holds the wave's instruction stream for 60-185
    // cycles, which then passes under MFMAs already queued instead of in front of them (+2-6 % per layer).
    constexpr int NP = A_IT + B_IT;
    auto issue_piece = [&](int buf, int q) __attribute__((always_inline)) {
        float* base = smem + buf * STAGE;
        if (q < A_IT) {
            const int j = q;
            if ((256 * j + 64 * wave) < A_SLOTS) {      // wave-uniform
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
                const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
                const unsigned off = a_pix[j] < 0 ? BUF_OOB : (unsigned)a_pix[j] * ld4 + a_qoff[j];
#ifndef CF_EXP_NOLOAD
                dma16_to_lds(rs, base + (256 * j + 64 * wave) * 4, off, so);
#else
                asm volatile("" ::"v"(off), "s"(so));
#endif
            }
        } else {
            const int it = q - A_IT;
            if ((256 * it + 64 * wave) < B_SLOTS) {
#ifndef CF_EXP_NOLOAD
                dma16_to_lds(b_rsrc, base + BM * KS + (256 * it + 64 * wave) * 4, b_off[it], (unsigned)it_k * 4u);
#else
                asm volatile("" ::"v"(b_off[it]));
#endif
            }
        }
        if (q == NP - 1) {
            // ---- advance the (wave-uniform) iterator ----
            it_k += KS;
            it_cs += KS;
            if (it_cs >= seg_cn) {
                it_cs = 0;
                ++it_seg;
                if (it_seg >= nseg) {
                    it_seg = 0;
                    ++it_kx;
                    if (it_kx >= kw_n) {
                        it_kx = 0;
                        ++it_ky;
                    }
                    tap_setup();
                }
                seg_base = sel3(p.in, it_seg) + (long)b * (it_seg == 0 ? p.seg_bs[0] : (it_seg == 1 ? p.seg_bs[1] : p.seg_bs[2]));
                seg_ld = it_seg == 0 ? p.seg_ld[0] : (it_seg == 1 ? p.seg_ld[1] : p.seg_ld[2]);
                seg_cn = it_seg == 0 ? p.seg_c[0] : (it_seg == 1 ? p.seg_c[1] : p.seg_c[2]);
            }
        }
    };
    auto issue_stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(buf, q);
    };
    // DMA wave-instructions THIS wave issues per stage (what the counted vmcnt leaves in flight)
    int nl = 0;
#pragma unroll
    for (int j = 0; j < A_IT; ++j) nl += ((256 * j + 64 * wave) < A_SLOTS) ? 1 : 0;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) nl += ((256 * it + 64 * wave) < B_SLOTS) ? 1 : 0;

    constexpr int DIST = NBUF - 1;                 // prefetch distance in stages
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nck) issue_stage(d);

    int rbuf = 0, wbuf = DIST;
#ifdef CF_STAMP
    const long long t_loop_begin = __builtin_readcyclecounter();
#endif
    for (int ck = 0; ck < nck; ++ck) {
        // stage ck has landed once at most the loads of stage ck+1 are still in flight
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        {   // stage ck has landed once only the loads of the (up to DIST - 1) later stages are still in flight
            const int later = nck - 1 - ck;
            wait_vmcnt_le(nl * (later < DIST - 1 ? later : DIST - 1));
        }
#ifdef CF_STAMP
        const long long t1 = __builtin_readcyclecounter();
#endif
        raw_barrier();
#ifdef CF_STAMP
        const long long t2 = __builtin_readcyclecounter();
#endif
        const bool more = ck + DIST < nck;      // stage ck + DIST goes into the buffer freed by the barrier above
#ifdef CF_STAMP
        st_wait += t1 - t0;
        st_bar += t2 - t1;
        st_issue += __builtin_readcyclecounter() - t2;
#endif
        const char* sbase = reinterpret_cast<const char*>(smem + rbuf * STAGE);
#pragma unroll
        for (int ks = 0; ks < KCW / 8; ++ks) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(sbase + fa[i][ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sbase + fb[j][ks]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#ifndef CF_EXP_NOMFMA
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
#else
                        acc[i][j][s] += af[i][s] * bf[j][s];
#endif
                    }
                // the DMA pieces of the stage being prefetched are spread evenly over the stage's MFMA groups
                constexpr int SLOTS = (KCW / 8) * 4, PSTEP = SLOTS / NP;
                if ((ks * 4 + s) % PSTEP == 0 && (ks * 4 + s) / PSTEP < NP) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) issue_piece(wbuf, (ks * 4 + s) / PSTEP);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        static_assert(NP <= (KCW / 8) * 4, "a stage's DMA pieces must fit between its MFMA groups");
        rbuf = rbuf == NBUF - 1 ? 0 : rbuf + 1;
        wbuf = wbuf == NBUF - 1 ? 0 : wbuf + 1;
    }
#ifdef CF_STAMP
    const long long t_loop_end = __builtin_readcyclecounter();
#endif
    wait_lgkm0();
    raw_barrier();          // every wave is done reading the ring before it is reused below

    if (WK > 1 && !p.st_partial) {
        // Split-K tail shared by all waves: every wave parks its partial 32x32 tile in patch layout, and after the barrier
        // finishes 4/WK of the patch's row groups (summing the WK partials on the way, in the order k = 0..WK-1).
        // Before, the wk > 0 waves left and one wave per sub-tile did the whole tail.
        constexpr int NSUB = TM * TN;                         // sub-tiles per wave (1, or 3 for the 32x96 tile)
        static_assert(WK == 1 || WK * WMN * 32 * EPI_S <= SMEM, "partial patches must fit the stage buffers");
        constexpr int PSTRIDE = WMN * 32 * EPI_S;             // between the WK partials of a sub-tile
        float* part = smem + wk * PSTRIDE + wmn * (32 * EPI_S);
#pragma unroll 1
        for (int pi = 0; pi < NSUB; ++pi) {                   // one sub-tile at a time: the patches stay 18 KB
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (pi == i * TN + j) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_S + lr] = acc[i][j][r];
                    }
            __syncthreads();
            const int pi_i = pi / TN, pi_j = pi - pi_i * TN;
            patch_tail(p, smem + wmn * (32 * EPI_S), b, m0 + (wm * TM + pi_i) * 32, n0 + (wn * TN + pi_j) * 32, lane, M,
                       wk * (4 / WK), (wk + 1) * (4 / WK), WK, PSTRIDE);
            if (pi + 1 < NSUB) __syncthreads();               // the patches are rewritten for the next sub-tile
        }
        return;
    }
    if (WK > 1) {
        float* red = smem;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((wk - 1) * WMN + wmn) * TM + i) * TN + j) * 1024 + r * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int k = 1; k < WK; ++k)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += red[((((k - 1) * WMN + wmn) * TM + i) * TN + j) * 1024 + r * 64 + lane];
    }

    float* sW = smem + RED + wmn * (32 * EPI_S);
    // One run-time loop over the wave's TM x TN sub-tiles: the (large) tail code exists once, and the compiler
    // cannot hoist every sub-tile's loads / address arithmetic to the top (which cost 60-130 VGPRs when the loop
    // was unrolled).  Only the copy of the accumulators into the patch is selected per sub-tile.
#pragma unroll 1
    for (int pi = 0; pi < TM * TN; ++pi) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (pi == i * TN + j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sW[((r & 3) + 8 * (r >> 2) + 4 * lh) * EPI_S + lr] = acc[i][j][r];
                }
        const int pi_i = pi / TN, pi_j = pi - pi_i * TN;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        patch_tail(p, sW, b, m0 + (wm * TM + pi_i) * 32, n0 + (wn * TN + pi_j) * 32, lane, M);
        if (p.st_partial) patch_stats(p, sW, b, m0 + (wm * TM + pi_i) * 32, n0 + (wn * TN + pi_j) * 32, lane, M);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef CF_STAMP
    if (p.stamp && lane == 0) {      // [wait, barrier, issue, prologue, stages, loop, tail, -] cycles of this wave
        long long* q = p.stamp + ((long)blockIdx.x * 4 + wave) * 8;
        q[0] = st_wait; q[1] = st_bar; q[2] = st_issue; q[3] = t_loop_begin - t_begin; q[4] = nck;
        q[5] = t_loop_end - t_loop_begin; q[6] = __builtin_readcyclecounter() - t_loop_end;
        // shader clock held over this wave's life in MHz: (cycle counter delta) / (100 MHz wall-clock delta) * 100
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions (reflect or zero pad), fp32:
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A          per 4x4 input tile d -> 2x2 output tile,
// i.e. 16 independent [tiles x Cin] x [Cin x Cout] products instead of 9 taps: 2.25x fewer matrix-core flops, the same
// fp32 products and accumulation (results differ from the direct kernel by the transforms' rounding, ~1e-6).
// The un-fused form (transform kernels + batched GEMM) moves 4x the input and 4x the output through memory and loses
// on this machine, so everything is fused in one workgroup (256 threads) per region of 4 x 8 tiles (8 x 16 output
// pixels) x 32 output channels, K = Cin in chunks of 8 channels:
//   raw     the (10 x 18 pixel) x 8-channel input patch of a chunk, LDS-DMA'd straight from the NHWC tensor (reflect /
//           zero padding resolved in the per-lane source offset; double buffered);
//   MFMA    wave w owns positions (w, 0..3) of the 4x4 grid = row w of B^T d B: 4 accumulators [32 tiles x 32 couts], 16 x
//           v_mfma_f32_32x32x2_f32 per chunk;
//   V       never stored: row w needs two of a tile's four patch rows, so lane (tile, channel quad) reads 2 x 4 pixels x 4
//           channels (8 ds_read_b128 of the raw patch) and computes exactly its own 16 A operands in registers (32 VALU
//           ops); the four waves together transform every (tile, channel) once; ONE barrier per chunk;
//   U       = G g G^T, transformed ONCE at weight-pack time into [n-block][chunk][pos][32 n][8 k] blocks (row 2 negated,
//           see wino_weight_kernel); nobody but wave w reads positions (w, .), so each lane loads its own B fragments from
//           global memory (L2) into registers, one chunk ahead, position by position behind that position's MFMAs;
//   tail    the A^T . A reduction is separable: along j inside the wave (registers), along i across the four waves
//           through LDS (32 KB exchange buffer over the raw ring); each wave then owns one tile row = 32 output pixels x
//           32 couts as a patch and runs the common fused epilogue (patch_tail) with a row -> pixel table.
// 118 VGPRs and 32.5 KB of LDS: four workgroups per CU.  History and measurements: DESIGN.md section 3.
// ---------------------------------------------------------------------------------------------------------
// tiles per region: 4 x 8 (8 x 16 output pixels, "wide") or 8 x 4 (16 x 8, "tall"), whichever wastes fewer pixels on the
// image's ragged edge (90 x 120: 96 x 128 = +13.8 % wide, 96 x 120 = +6.7 % tall); the patch is 10 x 18 or 18 x 10 pixels
#ifndef WG_ABL
#define WG_ABL 0
#endif
__global__ __launch_bounds__(256, 4) void conv_wino_kernel(const ConvParams p) {
#ifdef CF_CENSUS
    long long c0;
    census_begin(p, c0);
#endif
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long st_wait = 0, st_bar = 0, st_issue = 0;
#endif
    // ONE __shared__ object on purpose: with several, the compiler's LDS lowering tags each with an alias scope, its waitcnt
    // pass then tracks the LDS-DMA writes per scope and puts s_waitcnt vmcnt(0) in front of the first ds_read that follows a
    // DMA issue -- which serialises the prefetch of chunk k+1 with the transform of chunk k.  Without scopes the DMA hand-off
    // is left to the explicit vmcnt waits + barriers below (as in conv_dma_kernel).
    //   sRaw  [2][2 channel quads][192 cells]   raw patches of the current / next chunk (LDS-DMA ring of two), 12 KB
    //   after the loop: the cross-wave exchange X[4][2][32][32] (32 KB), then the epilogue patches; sMtab = their row -> pixel tables
    // No V buffer: wave w owns positions (w, 0..3), i.e. row w of B^T d B, which needs two of the four patch rows of each tile;
    // lane (tile lr, channel quad lh) reads those 2 x 4 pixels (4 channels each, ds_read_b128) and computes exactly the 16 values
    // it feeds to its MFMAs as A operands -- the four waves together do each (tile, channel) transform once, nothing is written
    // back to LDS, and one barrier per chunk (the raw hand-off) is left.  U never enters LDS either: nobody else reads a wave's
    // weights, so each lane fetches its own B fragments (16 bytes, L2 hits) straight into registers.
    constexpr int WG_A = 4 * 2 * 32 * 32;
    static_assert(2 * WG_RAW <= WG_A && 4 * 32 * EPI_S <= WG_A, "raw ring and epilogue patches overlay the exchange buffer");
#ifdef CF_LDS32K
    static_assert(4 * 32 * EPI_S + 4 * 32 <= WG_A, "the row -> pixel tables sit behind the epilogue patches, inside the dead exchange buffer");
    __shared__ __attribute__((aligned(16))) float smem[WG_A];            // 32,768 bytes exactly
    int* const sMtab = reinterpret_cast<int*>(smem + 4 * 32 * EPI_S);
#else
    __shared__ __attribute__((aligned(16))) float smem[WG_A + 4 * 32];
    int* const sMtab = reinterpret_cast<int*>(smem + WG_A);
#endif
    float* const sRaw = smem;
    float* const sPatch = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = p.Ho, Wo = p.Wo;
    const int tall = wino_tall(Ho, Wo);
    const int TWr = tall ? 4 : 8;                            // tile columns of a region (tile rows = 32 / TWr)
    const int RH = tall ? 16 : 8, RW = tall ? 8 : 16;        // region size in output pixels
    const int PC = RW + 2, PCh = PC >> 1;                    // patch columns (rows = RH + 2; PC * (RH + 2) = 180 either way)
    const int nrx = (Wo + RW - 1) / RW, nry = (Ho + RH - 1) / RH;
    const int nreg = nrx * nry;
    const int nt = (p.cout + 31) / 32;
    int tile_id = blockIdx.x;
    if (p.sched == 1) {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = tile_id & 7;
        tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile_id >> 3);
    }
    const int nblk = tile_id % nt;
    const int rest = tile_id / nt;
    const int reg = rest % nreg;
    const int b = rest / nreg;
    const int oy0 = (reg / nrx) * RH, ox0 = (reg % nrx) * RW;
    const int n0 = nblk * 32;

    // ---- raw patch DMA slots: slot s -> channel quad s / 192, patch cell s % 192 (180 live).  Cells of a patch row are stored
    // even columns first, then odd columns (cell = py * PC + (px >> 1) + (px & 1) * PC/2), so that the tiles of a tile row read
    // neighbouring 16-byte slots of their quad's plane ----
    int a_pix[2];
    unsigned a_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = tid + 256 * j;
        const int quad = sl / WG_PLANE, cell = sl - quad * WG_PLANE;      // quad 2: the dead slots 384..511
        a_q[j] = (unsigned)(quad & 1) * 16u;
        const int py = cell / PC, pc = cell - py * PC;
        const int px = pc < PCh ? 2 * pc : 2 * (pc - PCh) + 1;
        int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        bool ok = quad < 2 && cell < WG_PIX && iy <= p.Hin && ix <= p.Win;      // beyond the halo of the last row / column: unused
        if (p.pad_mode == 1) {
            iy = reflect_idx(iy, p.Hin);
            ix = reflect_idx(ix, p.Win);
        } else {
            ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        }
        a_pix[j] = ok ? iy * p.Win + ix : -1;
    }
    const int nchunk = p.cin_pad / WG_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino + (long)wgroup(p, b) * p.wino_gs + (long)nblk * nchunk * WG_UV);

    // chunk iterator over the channel segments (wave-uniform)
    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    // Every wave issues exactly two DMA instructions per chunk, unconditionally (dead slots and the chunk past the end fetch
    // out of range = zeros into unused LDS): with a fixed count the compiler's s_waitcnt vmcnt before each position's MFMAs
    // waits for that position's U registers only, not for the raw patch of the next chunk behind them in the queue.
    auto issue_raw = [&](int buf, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
        float* rbase = sRaw + buf * WG_RAW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = (a_pix[j] < 0 || !live) ? BUF_OOB : (unsigned)a_pix[j] * ld4 + a_q[j];
            dma16_to_lds(rs, rbase + (256 * j + 64 * wave) * 4, off, so);
        }
        // advance to the next chunk
        it_cs += WG_KC;
        if (it_cs >= seg_cn) {
            it_cs = 0;
            ++it_seg;
            if (it_seg < p.nseg) {
                seg_base = sel3(p.in, it_seg) + (long)b * (it_seg == 1 ? p.seg_bs[1] : p.seg_bs[2]);
                seg_ld = it_seg == 1 ? p.seg_ld[1] : p.seg_ld[2];
                seg_cn = it_seg == 1 ? p.seg_c[1] : p.seg_c[2];
            }
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // lane (lr, lh): tile lr = (ty, tx) of the region, channel quad lh; its patch rows: ra, rb of {0,2} {1,2} {1,2} {1,3} for wave 0..3
    const int lr = lane & 31, lh = lane >> 5;
    const int tty = lr / TWr, ttx = lr - tty * TWr;
    const int ra = wave == 0 ? 0 : 1, rb = wave == 0 ? 2 : wave == 3 ? 3 : 2;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const int cell0 = 2 * tty * PC + ttx;                                             // cell of patch pixel (2 ty, 2 tx)
    const int rd_a = (lh * WG_PLANE + cell0 + ra * PC) * 4, rd_b = (lh * WG_PLANE + cell0 + rb * PC) * 4;   // floats; + column cell * 4
    const unsigned uoff = (unsigned)((wave * 4) * 256 + lr * WG_KC + ((lh ^ ((lr >> 3) & 1)) << 2)) * 4u;  // + j KiB: position (wave, j) of a chunk's U block

    auto load_u = [&](int chunk, f32x4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = buf_load4(u_rsrc, uoff + 1024u * j, (unsigned)chunk * (WG_UV * 4u));
    };
    // one chunk: hand-off of raw(k); LDS-DMA of raw(k+1); this wave's row of the transform; 16 MFMAs, position by position, each
    // position's U registers refilled for chunk k+1 as soon as its four MFMAs are issued (a whole chunk of lead, no second buffer)
    f32x4 bu[4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    auto chunk_step = [&](int k) __attribute__((always_inline)) {
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        // in flight, oldest first: the two raw(k) pieces, then the four U(k) loads -- raw(k) has landed once at most four are outstanding
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        raw_barrier();                              // raw(k) has landed for everybody; everybody has read raw(k-1)
#ifdef CF_STAMP
        const long long t1 = __builtin_readcyclecounter();
#endif
        const bool more = k + 1 < nchunk;
#if !(WG_ABL & 8)
        issue_raw((k + 1) & 1, more);
#endif
        __builtin_amdgcn_sched_barrier(0);          // raw(k+1) before U(k+1) in issue order: the vmcnt(4) above counts on it
        f32x4 af[4];
#if !(WG_ABL & 1)
        {
            const float* r = sRaw + (k & 1) * WG_RAW;
            f32x4 da[4], db[4], t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = ((c >> 1) + (c & 1) * PCh) * 4;
                da[c] = *reinterpret_cast<const f32x4*>(r + rd_a + col);
                db[c] = *reinterpret_cast<const f32x4*>(r + rd_b + col);
            }
            // row `wave` of B^T d: d0 - d2, d1 + d2, -(d2 - d1), d1 - d3  =  da + sgn * db (exact: sgn = +-1; row 2 negated, as its U is)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sgn, db[c][e], da[c][e]);
            af[0] = t[0] - t[2];
            af[1] = t[1] + t[2];
            af[2] = t[2] - t[1];
            af[3] = t[1] - t[3];
        }
#else
        for (int j = 0; j < 4; ++j) af[j] = bu[j];
#endif
#ifdef CF_STAMP
        wait_lgkm0();
        const long long t2 = __builtin_readcyclecounter();
        st_wait += t1 - t0;
        st_issue += t2 - t1;
#endif
        // ---- 16 MFMAs: positions (wave, 0..3) ----
        const unsigned u_next = (unsigned)(more ? k + 1 : k) * (WG_UV * 4u);      // past the end: a harmless re-load
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][s2], bu[j][s2], acc[j], 0, 0, 0);
#if !(WG_ABL & 2)
            __builtin_amdgcn_sched_barrier(0);
            bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, u_next);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };

    issue_raw(0, true);
    __builtin_amdgcn_sched_barrier(0);
#if !(WG_ABL & 2)
    load_u(0, bu);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifdef CF_STAMP
    const long long t_loop_begin = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < nchunk; ++k) chunk_step(k);
#ifdef CF_STAMP
    const long long t_loop_end = __builtin_readcyclecounter();
#endif
#if WG_ABL & 4
    {
        float sum = 0.f;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
        if (sum == 12345.f) p.out[tid] = sum;
        return;
    }
#endif
    // the last chunk step's dead past-the-end DMA (zeros into the ring) must have landed before anything overlays the ring: an explicit
    // wait, so that this does not rest on the compiler's LDS-DMA bookkeeping in front of the barrier
    wait_vmcnt0();
    __syncthreads();                                // every wave is done with the raw ring before it becomes the exchange buffer
    // ---- output transform, j direction (in registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3 ----
    float* X = smem;                                // X[i = wave][bcol][tile 32][cout 32], over the raw ring
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        X[((wave * 2 + 0) * 32 + trow) * 32 + lr] = (acc[0][r] + acc[1][r]) + acc[2][r];
        X[((wave * 2 + 1) * 32 + trow) * 32 + lr] = (acc[1][r] - acc[2][r]) - acc[3][r];
    }
    __syncthreads();
    // ---- i direction across the waves + patch of this wave's tile row: 8 tiles x (2 x 2) pixels x 32 couts ----
    float* sW = sPatch + wave * (32 * EPI_S);
    int* mtab = sMtab + wave * 32;
    float yv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int prow = lh * 16 + q;               // patch row = tl * 4 + a * 2 + bb
        const int tl = prow >> 2, a = (prow >> 1) & 1, bb = prow & 1;
        const int t = wave * 8 + tl;
        const float x0 = X[((0 * 2 + bb) * 32 + t) * 32 + lr], x1 = X[((1 * 2 + bb) * 32 + t) * 32 + lr];
        const float x2 = X[((2 * 2 + bb) * 32 + t) * 32 + lr], x3 = X[((3 * 2 + bb) * 32 + t) * 32 + lr];
        yv[q] = a == 0 ? (x0 + x1) + x2 : (x1 - x2) - x3;
    }
    __syncthreads();                                // everybody has read X: the patches go on top of it
#pragma unroll
    for (int q = 0; q < 16; ++q) sW[(lh * 16 + q) * EPI_S + lr] = yv[q];
    if (lane < 32) {
        const int tl = lane >> 2, a = (lane >> 1) & 1, bb = lane & 1;
        const int t = wave * 8 + tl, ty = t / TWr, tx = t - ty * TWr;
        const int oy = oy0 + 2 * ty + a, ox = ox0 + 2 * tx + bb;
        mtab[lane] = (oy < Ho && ox < Wo) ? oy * Wo + ox : -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    patch_tail(p, sW, b, 0, n0, lane, Ho * Wo, 0, 4, 1, 0, mtab);
    if (p.st_partial) patch_stats(p, sW, b, 0, n0, lane, Ho * Wo, mtab, reg * 4 + wave, nreg * 4);
#ifdef CF_CENSUS
    census_end(p, c0);
#endif
#ifdef CF_STAMP
    if (p.stamp && lane == 0) {      // [DMA wait + barrier, second barrier, issue + transform, prologue, chunks, loop, tail, MHz]
        long long* q = p.stamp + ((long)blockIdx.x * 4 + wave) * 8;
        q[0] = st_wait; q[1] = st_bar; q[2] = st_issue; q[3] = t_loop_begin - t_begin; q[4] = nchunk;
        q[5] = t_loop_end - t_loop_begin; q[6] = __builtin_readcyclecounter() - t_loop_end;
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
}

// U = G g G^T of a packed direct matrix w [rows][tap][cin_pad] (BatchNorm folds, stacking, interleaving already applied),
// stored [n-block][chunk][pos = i*4+j][32 n][8 k] with the 16-byte k quads of a row swapped for rows 8..15 and 24..31
// (the swizzle conv_wino_kernel reads with); rows past `rows` are zero
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int rows, int cin_pad, int nblk) {
    const int nchunk = cin_pad / WG_KC;
    const long total = (long)nblk * nchunk * WG_UV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kk = (int)(idx & 7);
    const int nl = (int)((idx >> 3) & 31);
    const int pos = (int)((idx >> 8) & 15);
    const long blk = idx >> 12;
    const int chunk = (int)(blk % nchunk);
    const int nb = (int)(blk / nchunk);
    const int n = nb * 32 + nl;
    const int kq = (kk >> 2) ^ ((nl >> 3) & 1);                 // un-swizzle: which logical k this slot holds
    const int c = chunk * WG_KC + kq * 4 + (kk & 3);
    float val = 0.f;
    if (n < rows) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
        const int i = pos >> 2, j = pos & 3;
        double acc = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int bq = 0; bq < 3; ++bq)
                acc += (double)G[i][a] * (double)w[(long)n * 9 * cin_pad + (long)(a * 3 + bq) * cin_pad + c] * (double)G[j][bq];
        // row i = 2 is stored NEGATED: the kernels compute -(row 2 of B^T d B) = (d1 - d2 ...), so that three of the four waves share
        // one formula (a - b) and the fourth (a + b) differs by a scalar sign only; (-V)(-U) = V U exactly
        val = i == 2 ? -(float)acc : (float)acc;
    }
    u[idx] = val;
}

hipError_t launch_wino_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s) {
    if (!w || !u || rows <= 0 || cin_pad <= 0 || (cin_pad % WG_KC) != 0) return hipErrorInvalidValue;
    const int nblk = (rows + 31) / 32;
    const long total = (long)nblk * (cin_pad / WG_KC) * WG_UV;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, rows, cin_pad, nblk);
    return hipGetLastError();
}
long wino_weight_floats(int rows, int cin_pad) { return (long)((rows + 31) / 32) * (cin_pad / WG_KC) * WG_UV; }
static int wino_regions(int Ho, int Wo) {
    return wino_tall(Ho, Wo) ? ((Ho + 15) / 16) * ((Wo + 7) / 8) : ((Ho + 7) / 8) * ((Wo + 15) / 16);
}

// statistics partials a convolution with st_partial writes per image: one per 32-pixel patch, or per Winograd tile row
int conv_stats_chunks(const ConvParams& p, int tile) {
    if (tile == 40 || tile == 44 || tile == 45 || tile == 48 || tile == 49) return wino_regions(p.Ho, p.Wo) * 4;
    if (tile == 42) return wino4_regions(p.Ho, p.Wo) * 16;
    if (tile == 47 || tile == 50) return wino16_regions(p.Ho, p.Wo) * 2;
    return (p.Ho * p.Wo + 31) / 32;
}

static bool wino_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || (p.prec != 0 && p.prec != 3) || !p.w_wino || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.padT != 1 || p.padL != 1) return false;
    if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin < 12 || p.Win < 12) return false;
    if (p.w_bs != 0 && p.w_div <= 1) return false;          // per-image matrices (correlation GEMM); weight groups are fine
    for (int i = 0; i < p.nseg; ++i) {
        if (p.seg_c[i] % WG_KC) return false;
    }
    // the raw patch is addressed with 32-bit byte offsets through a buffer resource: a larger image would read zeros past the
    // range check instead of failing, so such a launch falls back to the direct kernels
    return dma_range_ok(p);
}

static hipError_t launch_wino(const ConvParams& p, int batch, hipStream_t s) {
    if (!wino_ok(p)) return hipErrorInvalidValue;
    const long nreg = wino_regions(p.Ho, p.Wo);
    const long wgs = nreg * ((p.cout + 31) / 32) * batch;
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256;
    hipLaunchKernelGGL(conv_wino_kernel, dim3((unsigned)wgs), dim3(256), 0, s, p);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int WK, int KCW, int NBUF = 3>
static hipError_t launch_dma(const ConvParams& p, int batch, hipStream_t s) {
    if (p.a_mode != A_NHWC || p.prec != 0 || !stage_ok(p, KCW * WK)) return hipErrorInvalidValue;
    const int M = p.Ho * p.Wo;
    dim3 grid(((M + BM - 1) / BM) * ((p.cout + BN - 1) / BN) * batch);
    g_last_launch.threads = (long)grid.x * 256;
    hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, WK, KCW, NBUF>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// tile ids: 1 = 128x128, 2 = 128x64, 3 = 128x96, 4 = 64x64, 5 = 64x128, 6 = 128x32
// ---------------------------------------------------------------------------
// Convolutions with 1-2 output channels (FlowHead.conv2 256->2, final_conv 64->1): on the matrix
// cores they would waste 31/32 of every tile, and their M x K work is tiny, so they run on the vector
// ALUs instead: LPP = Cin/4 adjacent lanes share one output pixel (each lane owns 4 input channels and
// does 16-byte loads), every tap is one coalesced row read, and the lane group is folded with
// wave shuffles.  HBM-bound by construction (reads the input once per tap through L1/L2).
// ---------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv_smalln_kernel(const ConvParams p, int lpp) {
    const int tid = threadIdx.x;
    const int g = tid / lpp;                 // pixel slot inside the workgroup
    const int ppw = 256 / lpp;               // pixels per workgroup
    const int c0 = (tid - g * lpp) * 4;
    const int b = blockIdx.y;
    const int M = p.Ho * p.Wo;
    const int m = blockIdx.x * ppw + g;
    const bool live = m < M;
    const int oy = live ? m / p.Wo : 0;
    const int ox = live ? m - oy * p.Wo : 0;
    const float* src = p.in[0] + (long)b * p.seg_bs[0] + c0;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
        for (int kx = 0; kx < p.KW; ++kx) {
            int iy = oy * p.stride + ky - p.padT;
            int ix = ox * p.stride + kx - p.padL;
            bool ok = live;
            if (p.pad_mode == 1) {
                iy = reflect_idx(iy, p.Hin);
                ix = reflect_idx(ix, p.Win);
            } else {
                ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            }
            if (ok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((long)iy * p.Win + ix) * p.seg_ld[0]);
                const int koff = (ky * p.KW + kx) * p.cin_pad + c0;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (long)co * p.Ktot + koff);
                    acc[co] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float a = acc[co];
        for (int off = lpp >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off);
        acc[co] = a;
    }
    if (live && c0 == 0) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = acc[co] + (p.bias ? p.bias[co] : 0.f);
            const long ooff = (long)b * p.out_bs + (long)m * p.out_ld + (long)co * p.out_cs;
            if (p.epi == EPI_SIGMOID) v = sigmoidf_(v);
            else if (p.epi == EPI_RELU) v = fmaxf(v, 0.f);
            else if (p.epi == EPI_TANH) v = tanhf(v);
            else if (p.epi == EPI_ADD_AUX) v += p.aux0[(long)b * p.aux0_bs + (long)m * p.aux0_ld + (long)co * p.aux0_cs];
            p.out[ooff] = v;
        }
    }
}

// 3x3 / stride 1 / pad 1 specialisation of the small-N conv (final_conv 64->1, FlowHead.conv2 256->2): HBM-bound.
// The generic kernel above re-reads every input quad 9x and every weight quad per tap through the L1 (64 B/clk/CU),
// which is what bounded it (~1.3 TB/s).  Here a lane keeps its weight quads (COUT x 9) in registers and walks down
// a column of RY output rows with a 3-row register window, so each input quad is loaded 3x (the x taps) instead
// of 9x and no weight is reloaded.  lpp lanes share a pixel (4 channels each) and fold with xor-shuffles.
template <int COUT>
__global__ __launch_bounds__(256) void conv_small3x3_kernel(const ConvParams p, int lpp, int RY) {
    const int tid = threadIdx.x;
    const int g = tid / lpp;                 // column slot inside the workgroup
    const int ppw = 256 / lpp;               // columns per workgroup
    const int c0 = (tid - g * lpp) * 4;
    const int b = blockIdx.z;
    const int H = p.Hin, W = p.Win;
    const int x = blockIdx.x * ppw + g;
    const int y0 = blockIdx.y * RY;
    const bool live = x < W;
    const int ld = p.seg_ld[0];
    const float* src = p.in[0] + (long)b * p.seg_bs[0] + c0;
    f32x4 w[COUT][9];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) w[co][t] = *reinterpret_cast<const f32x4*>(p.w + (long)co * p.Ktot + t * p.cin_pad + c0);
    // column offsets of the three x taps (reflect / zero padding)
    int xo[3];
    bool xok[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        int ix = x + kx - 1;
        bool ok = live;
        if (p.pad_mode == 1) ix = reflect_idx(ix, W);
        else ok = ok && ix >= 0 && ix < W;
        xok[kx] = ok;
        xo[kx] = ok ? ix : 0;
    }
    auto load_row = [&](int y, f32x4 (&r)[3]) __attribute__((always_inline)) {
        int iy = y;
        bool ok = true;
        if (p.pad_mode == 1) iy = reflect_idx(iy, H);
        else ok = iy >= 0 && iy < H;
        const float* row = src + (long)(ok ? iy : 0) * W * ld;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && xok[kx]) v = *reinterpret_cast<const f32x4*>(row + (long)xo[kx] * ld);
            r[kx] = v;
        }
    };
    f32x4 r0[3], r1[3], r2[3], r3[3];
    load_row(y0 - 1, r0);
    load_row(y0, r1);
    load_row(y0 + 1, r2);
    const int yend = (y0 + RY < H) ? y0 + RY : H;
    for (int y = y0; y < yend; ++y) {
        load_row(y + 2, r3);                 // one row ahead of the one this iteration consumes
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float a = 0.f;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    a += r0[kx][e] * w[co][kx][e] + r1[kx][e] * w[co][3 + kx][e] + r2[kx][e] * w[co][6 + kx][e];
            }
            for (int off = lpp >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off);
            acc[co] = a;
        }
        if (live && c0 == 0) {
            const int m = y * W + x;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                float v = acc[co] + (p.bias ? p.bias[co] : 0.f);
                const long ooff = (long)b * p.out_bs + (long)m * p.out_ld + (long)co * p.out_cs;
                if (p.epi == EPI_SIGMOID) v = sigmoidf_(v);
                else if (p.epi == EPI_RELU) v = fmaxf(v, 0.f);
                else if (p.epi == EPI_TANH) v = tanhf(v);
                else if (p.epi == EPI_ADD_AUX) v += p.aux0[(long)b * p.aux0_bs + (long)m * p.aux0_ld + (long)co * p.aux0_cs];
                p.out[ooff] = v;
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            r0[kx] = r1[kx];
            r1[kx] = r2[kx];
            r2[kx] = r3[kx];
        }
    }
}

static int default_small3x3() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_SMALL3X3");
        v = e ? atoi(e) : 1;
    }
    return v;
}

static bool smalln_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || p.nseg != 1 || p.cout > 2 || p.w_bs != 0 || p.bias_gs != 0) return false;
    if (p.cin_pad != 64 && p.cin_pad != 128 && p.cin_pad != 256) return false;
    return p.epi == EPI_NONE || p.epi == EPI_SIGMOID || p.epi == EPI_RELU || p.epi == EPI_TANH || p.epi == EPI_ADD_AUX;
}

static bool small3x3_ok(const ConvParams& p) {
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.padT == 1 && p.padL == 1 && p.Ho == p.Hin && p.Wo == p.Win && p.Hin >= 3 &&
           p.Win >= 3;
}

// force3x3: tile 15 (tests); otherwise the window kernel is used where it pays: the full-resolution final conv
// (the 1/8-resolution FlowHead conv is launch-latency-bound either way)
static hipError_t launch_smalln(const ConvParams& p, int batch, hipStream_t s, bool force3x3) {
    const int lpp = p.cin_pad / 4;
    if (force3x3 && !small3x3_ok(p)) return hipErrorInvalidValue;
    if (small3x3_ok(p) && (force3x3 || (default_small3x3() && (long)p.Hin * p.Win * batch >= 100000))) {
        const int ppw = 256 / lpp;
        // rows per workgroup: long columns amortise the 2-row warm-up, but keep >= ~1000 workgroups in flight
        int RY = 12;
        while (RY > 3 && (long)((p.Win + ppw - 1) / ppw) * ((p.Hin + RY - 1) / RY) * batch < 1024) RY -= 3;
        dim3 grid((p.Win + ppw - 1) / ppw, (p.Hin + RY - 1) / RY, batch);
        note_launch(p.cout == 1 ? "conv_small3x3_kernel<1>" : "conv_small3x3_kernel<2>", grid, dim3(256));
        if (p.cout == 1) hipLaunchKernelGGL(conv_small3x3_kernel<1>, grid, dim3(256), 0, s, p, lpp, RY);
        else hipLaunchKernelGGL(conv_small3x3_kernel<2>, grid, dim3(256), 0, s, p, lpp, RY);
        return hipGetLastError();
    }
    const int ppw = 256 / lpp;
    const int M = p.Ho * p.Wo;
    dim3 grid((M + ppw - 1) / ppw, batch);
    note_launch(p.cout == 1 ? "conv_smalln_kernel<1>" : "conv_smalln_kernel<2>", grid, dim3(256));
    if (p.cout == 1) hipLaunchKernelGGL(conv_smalln_kernel<1>, grid, dim3(256), 0, s, p, lpp);
    else hipLaunchKernelGGL(conv_smalln_kernel<2>, grid, dim3(256), 0, s, p, lpp);
    return hipGetLastError();
}

const char* conv_tile_name(int tile) {
    switch (tile) {
        case 1: return "conv_igemm_kernel<128,128,2,2,1>";
        case 2: return "conv_igemm_kernel<128,64,2,2,1>";
        case 3: return "conv_igemm_kernel<128,96,4,1,1>";
        case 4: return "conv_igemm_kernel<64,64,2,2,1>";
        case 5: return "conv_igemm_kernel<64,128,2,2,1>";
        case 6: return "conv_igemm_kernel<128,32,4,1,1>";
        case 7: return "conv_smalln_kernel";
        case 8: return "conv_igemm_kernel<32,32,1,1,4>";
        case 9: return "conv_igemm_kernel<32,64,1,2,2>";
        case 10: return "conv_igemm_kernel<32,64,1,2,2,kcw32>";
        case 11: return "conv_igemm_kernel<32,32,1,1,4,kcw32>";
        case 12: return "conv_igemm_kernel<64,64,2,2,1,kcw32>";
        case 13: return "conv_igemm_kernel<128,128,2,2,1,kcw32>";
        case 14: return "conv_igemm_kernel<128,64,2,2,1,kcw32>";
        case 20: return "conv_dma_kernel<32,64,1,2,2,16>";
        case 21: return "conv_dma_kernel<32,64,1,2,2,32>";
        case 22: return "conv_dma_kernel<32,32,1,1,4,16>";
        case 23: return "conv_dma_kernel<64,64,2,2,1,16>";
        case 24: return "conv_dma_kernel<64,64,2,2,1,32>";
        case 25: return "conv_dma_kernel<128,128,2,2,1,16>";
        case 26: return "conv_dma_kernel<128,64,2,2,1,16>";
        case 27: return "conv_dma_kernel<128,64,2,2,1,32>";
        case 28: return "conv_dma_kernel<64,128,2,2,1,16>";
        case 29: return "conv_dma_kernel<128,96,4,1,1,16>";
        case 30: return "conv_dma_kernel<128,32,4,1,1,16>";
        case 31: return "conv_dma_kernel<32,64,1,2,2,16,nbuf4>";
        case 32: return "conv_dma_kernel<64,64,2,2,1,16,nbuf4>";
        case 33: return "conv_dma_kernel<128,64,2,2,1,16,nbuf4>";
        case 34: return "conv_dma_kernel<32,96,1,1,4,8>";
        case 40: return "conv_wino_kernel";
        case 42: return "conv_wino4_kernel";
        case 46: return "conv_wino1d_kernel";
        case 47: return "conv_wino16_kernel";
        case 50: return "conv_wino16_kernel<deep>";
        case 48: return "conv_wino_p_kernel<0>";
        case 49: return "conv_wino_p_kernel<1>";
        case 44: return "conv_wino_sk_kernel<2>";
        case 45: return "conv_wino_sk_kernel<4>";
        default: return "?";
    }
}

// Fraction of a convolution's ALGORITHMIC flops (2 M N taps Cin: what the reference computes) that the kernel of tile kind `tile`
// executes on the matrix cores: 16 / 36 for Winograd F(2x2,3x3), 6 / 10 for the one-dimensional F(2,5), 36 / 144 for F(4x4,3x3), 1 for the
// direct kernels.  bench.py prices the executed-MFMA roofline fraction with it (per launch-site row, not by kernel-name prefix).
double conv_tile_mfma_ratio(int tile) {
    switch (tile) {
        case 40: case 44: case 45: case 47: case 48: case 49: case 50: return 4.0 / 9.0;
        case 46: return 0.6;
        case 42: return 0.25;
        default: return 1.0;
    }
}

static int default_sched() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_SCHED");
        v = e ? atoi(e) : 1;
    }
    return v;
}

static int default_dma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CF_DMA");
        v = e ? atoi(e) : 1;
    }
    return v;
}

// dry: validate the descriptor and CHOOSE the tile exactly as a real launch would (tile_used, g_last_launch.kernel), but put nothing on
// the stream -- what cf_conv_plan and tests/test_kernel_selection_cpu.py use to pin the launcher's heuristics without a GPU
hipError_t launch_conv(const ConvParams& p_in, int batch, hipStream_t s, int tile, int* tile_used, bool dry) {
    ConvParams p = p_in;
#if defined(CF_STAMP) || defined(CF_CENSUS)
    { const char* e = getenv("CF_STAMP_BUF"); p.stamp = e ? reinterpret_cast<long long*>(strtoull(e, nullptr, 10)) : nullptr; }
#endif
    if (p.sched < 0) p.sched = default_sched();
    // ---- host-side shape checks: a bad descriptor must never reach the GPU ----
    if (p.Ktot <= 0 || (p.Ktot % KC) != 0 || p.cout <= 0 || batch <= 0 || p.Ho <= 0 || p.Wo <= 0)
        return hipErrorInvalidValue;
    if (!p.out || !p.w || !p.in[0]) return hipErrorInvalidValue;
    if (p.a_mode == A_GATHER) {
        if (p.g_cin <= 0 || p.Ktot < p.KH * p.KW * p.g_cin || p.Ktot > 256 || p.KH > 255 || p.KW > 255) return hipErrorInvalidValue;
        if (p.Hin != p.Hsrc + p.g_offy || p.Win != p.Wsrc + p.g_offx) return hipErrorInvalidValue;
    } else {
        if (p.cin_pad <= 0 || (p.cin_pad % KC) != 0 || p.Ktot != p.KH * p.KW * p.cin_pad)
            return hipErrorInvalidValue;
        int sum = 0;
        if (p.nseg < 1 || p.nseg > 3) return hipErrorInvalidValue;
        for (int i = 0; i < p.nseg; ++i) {
            if (!p.in[i] || (p.seg_c[i] % KC) != 0 || (p.seg_ld[i] % 4) != 0 || p.seg_ld[i] < p.seg_c[i])
                return hipErrorInvalidValue;
            if ((reinterpret_cast<uintptr_t>(p.in[i]) & 15) != 0) return hipErrorInvalidValue;
            sum += p.seg_c[i];
        }
        if (sum != p.cin_pad) return hipErrorInvalidValue;
        if (p.a_mode == A_UPS2X && (p.Hin != 2 * p.Hsrc || p.Win != 2 * p.Wsrc)) return hipErrorInvalidValue;
        if (p.a_mode == A_NHWC && (p.Hin != p.Hsrc || p.Win != p.Wsrc)) return hipErrorInvalidValue;
    }
    if (p.pad_mode == 1 && (p.padT >= p.Hin || p.padL >= p.Win || p.KH - 1 - p.padT >= p.Hin || p.KW - 1 - p.padL >= p.Win))
        return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(p.w) & 15) != 0) return hipErrorInvalidValue;
    if (p.w_rows < p.cout) return hipErrorInvalidValue;
    // the last input row/col a valid output touches must exist (stride/pad consistency)
    if ((p.Ho - 1) * p.stride - p.padT + 0 >= p.Hin || (p.Wo - 1) * p.stride - p.padL >= p.Win)
        return hipErrorInvalidValue;

    if (p.st_partial && (p.epi != EPI_NONE || p.addend || p.cout <= 2 || (reinterpret_cast<uintptr_t>(p.st_partial) & 15) != 0))
        return hipErrorInvalidValue;       // statistics are taken over acc + bias, MFMA tiles only
    {   // 16-byte tail accesses need every row the epilogue touches to be quad aligned
        auto al = [](const void* q, long ld, long bs) {
            return q == nullptr || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld % 4) == 0 && (bs % 4) == 0);
        };
        p.epi_vec = al(p.out, p.out_ld, p.out_bs) && al(p.out2, p.out2_ld, p.out2_bs) && al(p.aux0, p.aux0_ld, p.aux0_bs) &&
                    al(p.aux1, p.aux1_ld, p.aux1_bs) && al(p.aux2, p.aux2_ld, p.aux2_bs) &&
                    al(p.addend, p.addend_ld, p.addend_bs) && (p.split % 4) == 0 && p.out_cs == 1 && p.epi != EPI_ADD_AUX;
        // the fast tail addresses rows with 32-bit byte offsets inside one image
        long max_ld = p.out_ld;
        for (long l : {(long)p.out2_ld, (long)p.aux0_ld, (long)p.aux1_ld, (long)p.aux2_ld, (long)p.addend_ld}) max_ld = l > max_ld ? l : max_ld;
        if ((long)p.Ho * p.Wo * max_ld * 4 >= 0x7FFFFF00L) p.epi_vec = 0;
        if ((reinterpret_cast<uintptr_t>(p.bias) & 15) != 0 || (reinterpret_cast<uintptr_t>(p.lam) & 15) != 0)
            return hipErrorInvalidValue;
        if (p.epi == EPI_ADD_AUX_SHRINK && !p.lam) return hipErrorInvalidValue;
        if (p.addend && p.epi == EPI_LSTC) return hipErrorInvalidValue;      // they share a register slot
        // the fused ConvLSTM cell exists only in the fast tail (gate-interleaved rows, whole 32-column patches)
        if (p.epi == EPI_LSTM_CELL && (!p.epi_vec || p.addend || (p.cout % 32) != 0 || !p.aux0 || !p.out2)) return hipErrorInvalidValue;
    }
    const bool auto_tile = tile == 0;
    if (tile == 0 && smalln_ok(p)) tile = 7;
    if (tile == 7 || tile == 15) {
        if (!smalln_ok(p)) return hipErrorInvalidValue;
        if (tile_used) *tile_used = 7;
        if (dry) { g_last_launch.kernel = conv_tile_name(7); return hipSuccess; }
        return launch_smalln(p, batch, s, tile == 15);
    }
    // 3x3 / stride 1 layers whose caller supplied the transformed weights run as Winograd F(2x2,3x3): 1.2-1.55x the direct
    // kernel (tools/conv_bench.py TILES=0,40) once the launch has >= ~128 workgroups; below that a workgroup's chain of
    // Cin/8 chunk steps (~1.4 us each, nothing to alternate with on its CU) is longer than the direct kernel's launch
    // (1/8-resolution maps at B <= 4: measured 0.7-1.0x)
    if (tile == 0 && wino_ok(p)) {
        const long tb = p.tile_batch > 0 ? p.tile_batch : batch;
        const long wg = (long)wino_regions(p.Ho, p.Wo) * ((p.cout + 31) / 32) * tb;
        static const long wmin = getenv("CF_WINO_MIN") ? atol(getenv("CF_WINO_MIN")) : 128;
        if (wg >= wmin) tile = 40;
        // Launches of at most CF_WINO_SK2_MAX workgroups (every CU gets ONE or none) split the chunks over 2 wave groups per workgroup
        // (conv_wino_sk.hip): +0.4 % on the step in three alternating A/B pairs on one box (1564 -> 1570 frames/s; menc.conv, encoder
        // stage 3).  CF_WINO_SK=0 turns it off.
        static const int sk_on = getenv("CF_WINO_SK") ? atoi(getenv("CF_WINO_SK")) : 1;
        // measured (tools/conv_bench.py, 1/8-resolution layers at B = 8): the split pays only while the workgroups still fit ONE round of
        // their larger footprint -- 192 workgroups (menc.conv, encoder stage 3): 36.7 -> 31.5 us / 18.1 -> 16.4 us with two groups, the same with
        // four; 288 (convc2), 384 (fh.conv1), 576 (stage 2): equal or slower (a second round of 64 KB / 128 KB workgroups)
        static const long sk4_max = getenv("CF_WINO_SK4_MAX") ? atol(getenv("CF_WINO_SK4_MAX")) : 0;
        static const long sk2_max = getenv("CF_WINO_SK2_MAX") ? atol(getenv("CF_WINO_SK2_MAX")) : 256;
        if (tile == 40 && sk_on && p.prec == 0 && p.cin_pad >= 64) {
            if (wg <= sk4_max) tile = 45;
            else if (wg <= sk2_max) tile = 44;
        }
    }
    // Half-size Winograd workgroups (tile 47, conv_wino16.hip: 16 tiles x 32 channels on the 16x16x4 MFMA) where conv_wino_kernel's
    // 128-pixel workgroups are too few to spread over 256 CUs (tools/conv_bench.py, B = 8): above the single-round split-K range and
    // up to CF_WINO16_MAX workgroups (convc2 288: 45 -> 39 us, fh.conv1 384: 26 -> 22.5, encoder stage 2 576: 28.5 -> 25.6; stage 1 with
    // 1536 loses: 38 -> 40), and below conv_wino_kernel's own threshold instead of the direct kernel (convf2, 96: 21.7 -> 12.9 us)
    static const int w16_kmin = getenv("CF_WINO16_KMIN") ? atoi(getenv("CF_WINO16_KMIN")) : 96;      // fewer channels: the doubled prologues / tails outweigh the better spread (cista-idnet's 32- and 64-channel layers: -0.8 % without this)
    if (auto_tile && (tile == 40 || tile == 0) && p.cin_pad >= w16_kmin && wino16_ok(p) && wino16_max() > 0) {
        const long tb = p.tile_batch > 0 ? p.tile_batch : batch;
        const long wg = (long)wino_regions(p.Ho, p.Wo) * ((p.cout + 31) / 32) * tb;
        const long wg16 = (long)wino16_regions(p.Ho, p.Wo) * ((p.cout + 31) / 32) * tb;
        static const long sk2_max = getenv("CF_WINO_SK2_MAX") ? atol(getenv("CF_WINO_SK2_MAX")) : 256;
        // 24: the smallest launch measured (r04, B = 1 at 180x240: convf2 24 workgroups 16.7 -> 12.5 us, menc 48: 29.8 -> 23.8, convc2 72:
        // 25.7 -> 20.0; the step at B = 1 / 2: +5 % / +3.5 % together with CF_WINO1D_MIN below, tools/r4_small.sh); was 96
        static const long w16min = getenv("CF_WINO16_MIN") ? atol(getenv("CF_WINO16_MIN")) : 24;
        if (tile == 40 && wg > sk2_max && wg <= wino16_max()) tile = 47;
        else if (tile == 0 && wg16 >= w16min) tile = 47;
        // at most ONE workgroup per CU: the deep-prefetch / software-pipelined instantiation (tile 50, bit-identical), see conv_wino16.hip
        const char* ed = getenv("CF_WINO16_DEEP_MAX");      // read per launch: tests flip it
        const long deep_max = ed ? atol(ed) : 400;      // (B = 8: convf2 only; convc2 / fh.conv1 at 576 / 768 gain 2-5 % per launch, nothing on the step)
        if (tile == 47 && wg16 <= deep_max) tile = 50;
    }
    // 1x5 / 5x1 layers with transformed weights (the separable GRU): one-dimensional Winograd F(2,5), 1.67x fewer MFMAs (conv_wino1d.hip)
    if (tile == 0 && wino1d_ok(p)) {
        const long tb = p.tile_batch > 0 ? p.tile_batch : batch;
        static const long w1min = getenv("CF_WINO1D_MIN") ? atol(getenv("CF_WINO1D_MIN")) : 44;      // B = 1: 88 (z, r) / 44 (q) workgroups, 16.7 -> 14.6 us each; was 128
        if (wino1d_workgroups(p, tb) >= w1min) tile = 46;
    }
    // F(4x4,3x3) (tile 42): 1.78x fewer MFMAs again, in workgroups of 512 output pixels x 32 channels -- taken when the launch
    // still has enough of them to fill the chip (CF_WINO4_MIN workgroups; 0 = never)
    if ((tile == 0 || (auto_tile && tile == 40)) && wino4_ok(p)) {
        const long tb = p.tile_batch > 0 ? p.tile_batch : batch;
        const long wg = (long)wino4_regions(p.Ho, p.Wo) * ((p.cout + 31) / 32) * tb;
        const long w4min = g_wino4_min;             // cf_create reads CF_WINO4_MIN (0 = never: the default)
        if (w4min > 0 && wg >= w4min) tile = 42;
    }
    // conv_wino_kernel's launches as persistent workgroups (tile 48, conv_wino_p.hip: bit-identical results).  OFF by default: measured
    // (r04, one box, three alternating pairs) 1599-1601 frames/s with tile 40 against 1585-1590 with tile 48, and +-3 % per layer either way
    // (tools/r4_winop.sh): with four workgroups resident per CU the hardware dispatcher already starts a fresh workgroup's prologue
    // beside the other three's MFMAs, which is all the walk buys.  CF_WINOP=1 turns it on.
    if (auto_tile && tile == 40 && wino_p_ok(p)) {
        const char* ew = getenv("CF_WINOP");               // read per launch: the model-level bit-identity test flips it
        const int winop = ew ? atoi(ew) : 0;
        if (winop) tile = winop == 2 ? 49 : 48;
    }
    if (tile == 0) {
        // Pick the largest tile that still yields >= ~2 workgroups per CU (measured with tools/conv_bench.py on
        // MI355X): big tiles reuse operands best, but a launch with fewer than ~512 workgroups leaves CUs idle,
        // so small-M layers step down to 64x64 and then to the intra-workgroup split-K tiles (32x64, 32x32).
        const long M = (long)p.Ho * p.Wo;
        const long tb = p.tile_batch > 0 ? p.tile_batch : batch;
        auto wgs = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.cout + bn - 1) / bn) * tb; };
        const long FILL = 512;
        if (p.cout <= 32) {
            tile = (wgs(128, 32) >= FILL || !splitk_ok(p, 4)) ? 6 : 8;
        } else if (p.cout <= 64) {
            if (wgs(128, 64) >= FILL) tile = 2;
            else if (wgs(64, 64) >= FILL) tile = 4;
            else if (splitk_ok(p, 2) && wgs(32, 64) >= FILL) tile = 9;
            else tile = splitk_ok(p, 4) ? 8 : 4;
        } else {
            const bool n96 = (p.cout % 96) == 0 && (p.cout % 128) != 0;
            if (n96 && wgs(128, 96) >= FILL) tile = 3;
            else if (!n96 && wgs(128, 128) >= FILL) tile = 1;
            else if (!n96 && wgs(64, 128) >= FILL) tile = 5;
            else if (wgs(64, 64) >= FILL) tile = 4;
            else if (splitk_ok(p, 2) && wgs(32, 64) >= FILL) tile = 9;
            else tile = splitk_ok(p, 4) ? 8 : (splitk_ok(p, 2) ? 9 : 4);
        }
    }
    if (tile == 0) return hipErrorInvalidValue;
    if (auto_tile) {
        const long wgs128x128 = (((long)p.Ho * p.Wo + 127) / 128) * ((p.cout + 127) / 128) * (p.tile_batch > 0 ? p.tile_batch : batch);
        // wide stages (32 k-columns per wave between barriers) measured 3-8 % faster wherever the channel
        // segments allow them (tools/conv_bench.py, MI355X)
        if (tile == 9 && stage_ok(p, 64)) tile = 10;
        else if (tile == 4 && stage_ok(p, 32)) tile = 12;
        // plain NHWC reads in fp32 go through the LDS-DMA kernel (3-15 % faster per layer, same arithmetic order)
        if (p.a_mode == A_NHWC && p.prec == 0 && default_dma() && dma_range_ok(p)) {
            if (tile == 9 || tile == 10) tile = 20;
            else if (tile == 8 && stage_ok(p, 64)) tile = 22;
            else if (tile == 4 || tile == 12) tile = 23;
            else if (tile == 2) tile = 23;      // 64x64 edges out 128x64 since the DMA pieces hide behind the MFMAs
            // (tile 34 = 32x96 split-K covers a 96-wide cout exactly and wins in isolation, 85 vs 73 TFLOP/s on 96->96
            // @48x64, but loses in the model -- 2.19 vs 2.11 ms for the encoder phase: 3 WGs/CU by registers and the
            // statistics tail of three sub-tiles on one wave -- so the launcher does not pick it)
            if (false) {}
            else if (tile == 5) tile = 28;
            else if (tile == 3) tile = 29;
            else if (tile == 6) tile = 30;
            else if (tile == 1)
                // 128x128 once there are >= 3 full rounds of it (config 4 / 5 sizes: 125-135 TFLOP/s); with fewer
                // workgroups the smaller tiles overlap prologue / tail better (180x240 B=8: tools/tile_sweep.sh)
                tile = wgs128x128 >= 2304 ? 25 : ((p.cout >= 256 && p.cout % 128 == 0) ? 28 : 23);
        }
    }
    if (((tile >= 20 && tile <= 40) || tile == 42 || tile == 44 || tile == 45 || tile == 47 || tile == 48 || tile == 49 || tile == 50) && !(p.a_mode == A_NHWC && dma_range_ok(p))) return hipErrorInvalidValue;
    if (tile_used) *tile_used = tile;
    g_last_launch.kernel = conv_tile_name(tile);
    if (p.prec != 0 && p.prec != 1 && p.prec != 3) return hipErrorInvalidValue;
    if (dry) return hipSuccess;
    if (p.prec != 0) {          // f16 modes: pre-split weights when there are some, else split B while staging
        if (p.w16 && (p.w_bs == 0 || p.w_div > 1)) {
            p.w = static_cast<const float*>(p.w16);
            p.b_f32 = 0;
        } else {
            p.b_f32 = 1;
        }
    }
    switch (tile) {
        case 1: return launch_t<128, 128, 2, 2, 1>(p, batch, s);
        case 2: return launch_t<128, 64, 2, 2, 1>(p, batch, s);
        case 3: return launch_t<128, 96, 4, 1, 1>(p, batch, s);
        case 4: return launch_t<64, 64, 2, 2, 1>(p, batch, s);
        case 5: return launch_t<64, 128, 2, 2, 1>(p, batch, s);
        case 6: return launch_t<128, 32, 4, 1, 1>(p, batch, s);
        case 8: return splitk_ok(p, 4) ? launch_t<32, 32, 1, 1, 4>(p, batch, s) : hipErrorInvalidValue;
        case 9: return splitk_ok(p, 2) ? launch_t<32, 64, 1, 2, 2>(p, batch, s) : hipErrorInvalidValue;
        case 10: return stage_ok(p, 64) ? launch_t<32, 64, 1, 2, 2, 32>(p, batch, s) : hipErrorInvalidValue;
        case 11: return stage_ok(p, 128) ? launch_t<32, 32, 1, 1, 4, 32>(p, batch, s) : hipErrorInvalidValue;
        case 12: return stage_ok(p, 32) ? launch_t<64, 64, 2, 2, 1, 32>(p, batch, s) : hipErrorInvalidValue;
        case 13: return stage_ok(p, 32) ? launch_t<128, 128, 2, 2, 1, 32>(p, batch, s) : hipErrorInvalidValue;
        case 14: return stage_ok(p, 32) ? launch_t<128, 64, 2, 2, 1, 32>(p, batch, s) : hipErrorInvalidValue;
        case 20: return launch_dma<32, 64, 1, 2, 2, 16>(p, batch, s);
        case 21: return launch_dma<32, 64, 1, 2, 2, 32>(p, batch, s);
        case 22: return launch_dma<32, 32, 1, 1, 4, 16>(p, batch, s);
        case 23: return launch_dma<64, 64, 2, 2, 1, 16>(p, batch, s);
        case 24: return launch_dma<64, 64, 2, 2, 1, 32>(p, batch, s);
        case 25: return launch_dma<128, 128, 2, 2, 1, 16>(p, batch, s);
        case 26: return launch_dma<128, 64, 2, 2, 1, 16>(p, batch, s);
        case 27: return launch_dma<128, 64, 2, 2, 1, 32>(p, batch, s);
        case 28: return launch_dma<64, 128, 2, 2, 1, 16>(p, batch, s);
        case 29: return launch_dma<128, 96, 4, 1, 1, 16>(p, batch, s);
        case 30: return launch_dma<128, 32, 4, 1, 1, 16>(p, batch, s);
        case 31: return launch_dma<32, 64, 1, 2, 2, 16, 4>(p, batch, s);
        case 32: return launch_dma<64, 64, 2, 2, 1, 16, 4>(p, batch, s);
        case 33: return launch_dma<128, 64, 2, 2, 1, 16, 4>(p, batch, s);
        case 34: return launch_dma<32, 96, 1, 1, 4, 8>(p, batch, s);
        case 40: return launch_wino(p, batch, s);
        case 42: return launch_wino4(p, batch, s);
        case 46: return launch_wino1d(p, batch, s);
        case 47: return launch_wino16(p, batch, s);
        case 50: return launch_wino16(p, batch, s, true);
        case 48: return launch_wino_p(p, batch, s, 0);
        case 49: return launch_wino_p(p, batch, s, 1);
        case 44: return wino_ok(p) ? launch_wino_sk(p, batch, s, 2) : hipErrorInvalidValue;
        case 45: return wino_ok(p) ? launch_wino_sk(p, batch, s, 4) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin,
                                   int KH, int KW, int cin_pad, int Ktot, int row0, int gather, int c_begin,
                                   int c_count, int dst_coff, int accum, const float* bn_w, const float* bn_b,
                                   const float* bn_mean, const float* bn_var, float bn_eps, const float* bias_src,
                                   float* bias_dst, int interleave) {
    const long total = (long)Cout * c_count * KH * KW;
    // interleave = G > 1: output channel o = g*(Cout/G) + j lands in row j*G + g (the G gates of hidden channel j
    // become one quad of the conv tail -- EPI_LSTM_CELL)
    const int per = interleave > 1 ? Cout / interleave : 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) {
        const int kw = idx % KW;
        long t = idx / KW;
        const int kh = t % KH;
        t /= KH;
        const int c = t % c_count;
        const int o = (int)(t / c_count);
        float v = src[(((long)o * Cin + (c_begin + c)) * KH + kh) * KW + kw];
        if (bn_var) v *= bn_w[o] / sqrtf(bn_var[o] + bn_eps);
        const int tap = kh * KW + kw;
        const long k = gather ? ((long)tap * c_count + c) : ((long)tap * cin_pad + dst_coff + c);
        const int orow = interleave > 1 ? (o % per) * interleave + o / per : o;
        float* d = dst + (long)(row0 + orow) * Ktot + k;
        *d = accum ? (*d + v) : v;
    }
    if (idx < Cout && bias_dst) {
        const int o = (int)idx;
        float bv = bias_src ? bias_src[o] : 0.f;
        if (bn_var) bv = (bv - bn_mean[o]) * (bn_w[o] / sqrtf(bn_var[o] + bn_eps)) + bn_b[o];
        bias_dst[row0 + (interleave > 1 ? (o % per) * interleave + o / per : o)] = bv;
    }
}

hipError_t launch_pack_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int cin_pad,
                              int Ktot, int row0, int gather, int c_begin, int c_count, int dst_coff, int accum,
                              const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var,
                              float bn_eps, const float* bias_src, float* bias_dst, hipStream_t s, int interleave) {
    if (c_count <= 0) { c_begin = 0; c_count = Cin; }
    if (interleave > 1 && (Cout % interleave) != 0) return hipErrorInvalidValue;
    const long total = (long)Cout * c_count * KH * KW;
    if (total <= 0 || c_begin < 0 || c_begin + c_count > Cin || dst_coff < 0) return hipErrorInvalidValue;
    if (gather) {
        if (Ktot < KH * KW * c_count || dst_coff != 0) return hipErrorInvalidValue;
    } else {
        if (cin_pad < dst_coff + c_count || Ktot != KH * KW * cin_pad) return hipErrorInvalidValue;
    }
    const int threads = 256;
    const long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, src, dst, Cout, Cin, KH, KW,
                       cin_pad, Ktot, row0, gather, c_begin, c_count, dst_coff, accum, bn_w, bn_b, bn_mean, bn_var, bn_eps,
                       bias_src, bias_dst, interleave);
    return hipGetLastError();
}

// fp32 packed matrix [rows][Ktot] -> f16 split copy with the LDS chunk format: per 16-column chunk 16 x hi then
// 16 x lo (64 bytes, the footprint of the 16 fp32 it replaces)
__global__ void split_weight_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = fminf(fmaxf(src[i], -65504.f), 65504.f);
    const _Float16 h = (_Float16)x;
    const long chunk = i >> 4;
    const int j = (int)(i & 15);
    dst[chunk * 32 + j] = h;
    dst[chunk * 32 + 16 + j] = (_Float16)(x - (float)h);
}

hipError_t launch_split_weight_f16(const float* src, void* dst, long rows, int Ktot, hipStream_t s) {
    const long n = rows * Ktot;
    if (!src || !dst || n <= 0 || (Ktot % 16) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(split_weight_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                       static_cast<_Float16*>(dst), n);
    return hipGetLastError();
}

}  // namespace cf
