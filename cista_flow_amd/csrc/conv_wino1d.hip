// conv_wino1d.hip -- one-dimensional Winograd F(2,5) for the 1x5 / 5x1 convolutions of the separable ConvGRU (SepConvGRU of the RAFT-style
// update block: reference DCEIFlow/update.py, ERAFT/update.py), fp32, stride 1, "same" padding (round 3).  Launched through launch_conv
// (conv_igemm.hip) as tile 46.  gfx950 only.
//
//     y[2t + j'] = sum_p A^T[j'][p] * ( sum_c (G g)[p][c] * (B^T d_t)[p][c] ),   d_t = the 6 input pixels 2t-2 .. 2t+3 along the line,
// i.e. six [tiles x Cin] x [Cin x Cout] products per pair of output pixels instead of ten taps: 1.67x fewer matrix-core flops with the
// same fp32 products and accumulation (points 0, +-1, +-2, inf -- B^T is the F(4,3) matrix --, error ~1e-6 of the output scale).
// The four GRU convolutions of an update iteration are 152 of its ~300 us: 5,520-row GEMMs with K = 1280 that the
// direct kernel runs at 65-90 TFLOP/s.
//
// Workgroup = 256 threads = 32 tiles (64 output pixels: consecutive tiles of the image's lines -- rows for 1x5, columns for 5x1) x 32
// output channels, K = Cin in chunks of 16 channels:
//   raw     [6 window pixels][4 channel quads][32 tiles] x 16 bytes per chunk (12 KB), LDS-DMA'd straight from the NHWC tensor, padding
//           resolved in the per-lane source offset (out of range = zeros); ring of two; tile-fastest, so a wave's ds_read_b128 of one
//           (pixel, quad) covers 512 contiguous bytes;
//   MFMA    wave (ph, kh) owns positions 3 ph .. 3 ph + 2 for channels [8 kh, 8 kh + 8) of every chunk: 3 accumulators [32 tiles x 32
//           couts], 12 x v_mfma_f32_32x32x2_f32 per chunk; lane (tile, lh) reads five pixels x 4 channels and computes exactly its own
//           12 A operands in registers (12 VALU ops); ONE barrier per chunk (the raw hand-off);
//   U       = G g, made once at weight-pack time, [n-block][chunk][6 pos][32 n][16 k]; every lane loads its own B fragments (16 bytes)
//           one chunk ahead, position by position behind that position's MFMAs;
//   tail    every wave applies its three columns of A^T in registers (two partial 32 x 32 patches), the four partials of a patch are
//           summed in a fixed order by the common fused epilogue (patch_tail with nparts = 4 and a row -> pixel table): wave w finishes
//           16 rows of output column j' = w & 1.
#include "conv_common.h"

namespace cf {

static constexpr int W1_KC = 16;                            // channels per chunk
static constexpr int W1_RAW = 6 * 4 * 32 * 4;               // floats per raw buffer (12 KB)
static constexpr int W1_UV = 6 * 32 * W1_KC;                // floats of a chunk's U block (3072)
static constexpr int W1_P = 32 * EPI_S;                     // floats of one partial patch

__global__ __launch_bounds__(256, 4) void conv_wino1d_kernel(const ConvParams p) {
    static_assert(2 * W1_RAW <= 8 * W1_P, "the raw ring overlays the partial patches");
    __shared__ __attribute__((aligned(16))) float smem[8 * W1_P + 64];   // ONE __shared__ object (see conv_wino_kernel): 37,120 bytes
    float* const sRaw = smem;
    int* const sMtab = reinterpret_cast<int*>(smem + 8 * W1_P);
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long st_wait = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave & 1, kh = wave >> 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int dir = p.KH == 5 ? 1 : 0;                      // 0: along a row (1x5), 1: along a column (5x1)
    const int L = dir ? p.Ho : p.Wo, nlines = dir ? p.Wo : p.Ho;
    const int TL = (L + 1) >> 1;                            // tiles per line
    const int nti = nlines * TL;                            // tiles per image
    const int ngrp = (nti + 31) >> 5;
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
    const int grp = rest % ngrp;
    const int b = rest / ngrp;
    const int n0 = nblk * 32;

    // this lane's tile (the same for its DMA slots, its A operands and -- lanes 0..31 of waves 0, 1 -- its row of the pixel table)
    const int T = grp * 32 + lr;
    const int line = T / TL, tt = T - line * TL;
    const bool t_ok = T < nti;

    // ---- raw DMA: wave-instruction wave + 4 i (i = 0..2) covers slots [64 (wave + 4 i), + 64): slot -> tile = slot & 31,
    // quad = (slot >> 5) & 3 = (2 wave + lh) & 3, window pixel j = slot >> 7 = (wave >> 1) + 2 i ----
    int a_pix[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (wave >> 1) + 2 * i;
        int u = 2 * tt - 2 + j;
        bool ok = t_ok;
        if (p.pad_mode == 1) u = reflect_idx(u, L);
        else ok = ok && u >= 0 && u < L;
        a_pix[i] = ok ? (dir ? u * p.Win + line : line * p.Win + u) : -1;
    }
    const unsigned a_q = (unsigned)((2 * wave + lh) & 3) * 16u;
    const int nchunk = p.cin_pad / W1_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino + (long)wgroup(p, b) * p.wino_gs + (long)nblk * nchunk * W1_UV);

    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    // three DMA instructions per wave and chunk, unconditionally (dead slots and the chunk past the end fetch out of range = zeros):
    // with a fixed count the s_waitcnt vmcnt before a position's MFMAs waits for that position's U only
    auto issue_raw = [&](int buf_off, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
        float* rbase = sRaw + buf_off;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = (a_pix[i] < 0 || !live) ? BUF_OOB : (unsigned)a_pix[i] * ld4 + a_q;
            dma16_to_lds(rs, rbase + 64 * (wave + 4 * i) * 4, off, so);
        }
        it_cs += W1_KC;
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

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // lane (tile lr, lh) reads quad 2 kh + lh of window pixel j at float offset ((4 j + 2 kh + lh) * 32 + lr) * 4
    const int rd = ((2 * kh + lh) * 32 + lr) * 4;
    const unsigned uoff = (unsigned)((3 * ph) * 512 + lr * W1_KC + 8 * kh + 4 * lh) * 4u;       // + pj * 2 KiB: position 3 ph + pj

    f32x4 bu[3];
    auto chunk_step = [&](int k) __attribute__((always_inline)) {
        // in flight, oldest first: the three raw(k) pieces, then the three U(k) loads
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        raw_barrier();                              // raw(k) has landed for everybody; everybody has read raw(k-1)
#ifdef CF_STAMP
        st_wait += __builtin_readcyclecounter() - t0;
#endif
        const bool more = k + 1 < nchunk;
        issue_raw(((k + 1) & 1) * W1_RAW, more);
        __builtin_amdgcn_sched_barrier(0);          // raw(k+1) before U(k+1) in issue order: the vmcnt(3) above counts on it
        const float* r = sRaw + (k & 1) * W1_RAW + rd;
        f32x4 v[3];
        if (ph == 0) {                              // wave-uniform
            // rows 0..2 of B^T d:  4 d0 - 5 d2 + d4 | -4 d1 - 4 d2 + d3 + d4 | 4 d1 - 4 d2 - d3 + d4
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(r + 0 * 512), d1 = *reinterpret_cast<const f32x4*>(r + 1 * 512);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(r + 2 * 512), d3 = *reinterpret_cast<const f32x4*>(r + 3 * 512);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(r + 4 * 512);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = __builtin_fmaf(-4.f, d2[e], d4[e]), s = __builtin_fmaf(-4.f, d1[e], d3[e]);
                v[0][e] = __builtin_fmaf(4.f, d0[e], t - d2[e]);
                v[1][e] = t + s;
                v[2][e] = t - s;
            }
        } else {
            // rows 3..5:  -2 d1 - d2 + 2 d3 + d4 | 2 d1 - d2 - 2 d3 + d4 | 4 d1 - 5 d3 + d5
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(r + 1 * 512), d2 = *reinterpret_cast<const f32x4*>(r + 2 * 512);
            const f32x4 d3 = *reinterpret_cast<const f32x4*>(r + 3 * 512), d4 = *reinterpret_cast<const f32x4*>(r + 4 * 512);
            const f32x4 d5 = *reinterpret_cast<const f32x4*>(r + 5 * 512);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = d4[e] - d2[e], s = d3[e] - d1[e];
                v[0][e] = __builtin_fmaf(2.f, s, t);
                v[1][e] = __builtin_fmaf(-2.f, s, t);
                v[2][e] = __builtin_fmaf(4.f, d1[e], __builtin_fmaf(-5.f, d3[e], d5[e]));
            }
        }
        const unsigned u_next = (unsigned)(more ? k + 1 : k) * (W1_UV * 4u);      // past the end: a harmless re-load
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][s2], bu[j][s2], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bu[j] = buf_load4(u_rsrc, uoff + 2048u * j, u_next);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_raw(0, true);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) bu[j] = buf_load4(u_rsrc, uoff + 2048u * j, 0);
    __builtin_amdgcn_sched_barrier(0);
    // (requesting raw and U TWO chunks ahead -- ring of three, two sets of U registers -- measured no faster: 34.8 vs 34.1 us; handing the
    // chunks off in PAIRS -- ring of four, one barrier per two chunks, 48 KB of LDS -- measured slower: 36.0 vs 33.3 us, 1585 vs 1601 frames/s; reading and
    // transforming chunk k + 1 in the shadow of chunk k's MFMAs -- 119 VGPRs -- measured no faster either: 35.3 us.
    // r04, for the one-wave-per-SIMD launches of B = 1 / 2 (what conv_wino16_kernel<1> is for the 3x3 layers): U two chunks ahead, ring of four,
    // pipelined A side, twelve hand-placed MFMA slots, branch-free iterator -- bit-identical, 1362 -> 1196 cycles per chunk for 768 of MFMA, but
    // 14.8 -> 14.4 us per launch only (prologue +1.5 k cycles) and 33.4 -> 34.5 us at B = 8; spreading the fillers evenly over the slots made it
    // SLOWER (15.0 us): profiles/r04_small_batch.txt.  Not kept.)
#ifdef CF_STAMP
    const long long t_loop_begin = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < nchunk; ++k) chunk_step(k);
#ifdef CF_STAMP
    const long long t_loop_end = __builtin_readcyclecounter();
#endif

    // the last chunk step's dead past-the-end DMA (zeros into the ring) must have landed before anything overlays the ring: an explicit
    // wait, so that this does not rest on the compiler's LDS-DMA bookkeeping in front of the barrier
    wait_vmcnt0();
    __syncthreads();                                // every wave is done with the raw ring before the patches go on top of it
    // ---- this wave's columns of A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1]: two partial patches [tile][cout], row stride EPI_S ----
    float* const sP = smem + wave * 2 * W1_P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float y0, y1;
        if (ph == 0) {
            y0 = (acc[0][r] + acc[1][r]) + acc[2][r];
            y1 = acc[1][r] - acc[2][r];
        } else {
            y0 = acc[0][r] + acc[1][r];
            y1 = __builtin_fmaf(2.f, acc[0][r] - acc[1][r], acc[2][r]);
        }
        sP[trow * EPI_S + lr] = y0;
        sP[W1_P + trow * EPI_S + lr] = y1;
    }
    if (wave < 2 && lane < 32) {                    // row -> pixel table of output column j' = wave
        const int u = 2 * tt + wave;
        sMtab[wave * 32 + lane] = (t_ok && u < L) ? (dir ? u * p.Wo + line : line * p.Wo + u) : -1;
    }
    __syncthreads();
    // wave w finishes rows [16 (w >> 1), + 16) of patch j' = w & 1: the sum of the four waves' partials, in wave order
    const int jp = wave & 1, q0 = (wave >> 1) * 2;
    patch_tail(p, smem + jp * W1_P, b, 0, n0, lane, p.Ho * p.Wo, q0, q0 + 2, 4, 2 * W1_P, sMtab + jp * 32);
#ifdef CF_STAMP
    if (p.stamp && lane == 0) {      // [DMA wait + barrier, -, -, prologue, chunks, loop, tail, MHz] cycles of this wave
        long long* q = p.stamp + ((long)blockIdx.x * 4 + wave) * 8;
        q[0] = st_wait; q[1] = 0; q[2] = 0; q[3] = t_loop_begin - t_begin; q[4] = nchunk;
        q[5] = t_loop_end - t_loop_begin; q[6] = __builtin_readcyclecounter() - t_loop_end;
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
}

// U = G g of a packed direct matrix w [rows][5 taps][cin_pad], stored [n-block][chunk][pos 6][32 n][16 k]; rows past `rows` are zero
__global__ void wino1d_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int rows, int cin_pad, int nblk) {
    const int nchunk = cin_pad / W1_KC;
    const long total = (long)nblk * nchunk * W1_UV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kk = (int)(idx & 15);
    const int nl = (int)((idx >> 4) & 31);
    const long pb = idx >> 9;
    const int pos = (int)(pb % 6);
    const long blk = pb / 6;
    const int chunk = (int)(blk % nchunk);
    const int nb = (int)(blk / nchunk);
    const int n = nb * 32 + nl;
    const int c = chunk * W1_KC + kk;
    float val = 0.f;
    if (n < rows) {
        // rows of G: s_i [1, a_i, a_i^2, a_i^3, a_i^4] for the points 0, 1, -1, 2, -2 with s = 1/4, -1/6, -1/6, 1/24, 1/24; the point at
        // infinity takes the last tap
        const double a[5] = {0.0, 1.0, -1.0, 2.0, -2.0};
        const double sc[5] = {0.25, -1.0 / 6.0, -1.0 / 6.0, 1.0 / 24.0, 1.0 / 24.0};
        const float* wr = w + (long)n * 5 * cin_pad + c;
        double acc = 0.0;
        if (pos == 5) {
            acc = (double)wr[4L * cin_pad];
        } else {
            double pw = 1.0;
            for (int t = 0; t < 5; ++t) {
                acc += pw * (double)wr[(long)t * cin_pad];
                pw *= a[pos];
            }
            acc *= sc[pos];
        }
        val = (float)acc;
    }
    u[idx] = val;
}

hipError_t launch_wino1d_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s) {
    if (!w || !u || rows <= 0 || cin_pad <= 0 || (cin_pad % W1_KC) != 0) return hipErrorInvalidValue;
    const int nblk = (rows + 31) / 32;
    const long total = (long)nblk * (cin_pad / W1_KC) * W1_UV;
    hipLaunchKernelGGL(wino1d_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, rows, cin_pad, nblk);
    return hipGetLastError();
}
long wino1d_weight_floats(int rows, int cin_pad) { return (long)((rows + 31) / 32) * (cin_pad / W1_KC) * W1_UV; }

static long wino1d_groups(const ConvParams& p) {
    const long L = p.KH == 5 ? p.Ho : p.Wo, nlines = p.KH == 5 ? p.Wo : p.Ho;
    return (nlines * ((L + 1) / 2) + 31) / 32;
}

bool wino1d_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || p.prec != 0 || !p.w_wino || p.stride != 1 || p.st_partial) return false;
    const bool row = p.KH == 1 && p.KW == 5 && p.padT == 0 && p.padL == 2, col = p.KH == 5 && p.KW == 1 && p.padT == 2 && p.padL == 0;
    if (!row && !col) return false;
    if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin < 1 || p.Win < 1) return false;
    if (p.pad_mode == 1 && (col ? p.Hin : p.Win) < 4) return false;       // one reflection only
    if (p.w_bs != 0 && p.w_div <= 1) return false;
    if (p.cin_pad % W1_KC) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg_c[i] % W1_KC) return false;
    return dma_range_ok(p);
}

long wino1d_workgroups(const ConvParams& p, long batch) { return wino1d_groups(p) * ((p.cout + 31) / 32) * batch; }

hipError_t launch_wino1d(const ConvParams& p, int batch, hipStream_t s) {
    if (!wino1d_ok(p)) return hipErrorInvalidValue;
    const long wgs = wino1d_workgroups(p, batch);
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256;
    hipLaunchKernelGGL(conv_wino1d_kernel, dim3((unsigned)wgs), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
