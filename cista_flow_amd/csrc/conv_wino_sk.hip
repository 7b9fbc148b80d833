// conv_wino_sk.hip -- conv_wino_kernel (Winograd F(2x2,3x3), conv_igemm.hip) with the channel chunks SPLIT over SK wave groups of
// one workgroup (round 3).  Launched through launch_conv (conv_igemm.hip) as tile 44 (SK = 2) / 45 (SK = 4).  gfx950 only.
//
// STATUS: parity-green (test_conv_winograd, tiles 44 / 45; test_conv_fused_inorm_stats); SK = 2 is what launch_conv takes for Winograd
// launches of at most 256 workgroups (CF_WINO_SK2_MAX; CF_WINO_SK=0 turns it off): +0.4 % on the 180x240 B=8 step in three alternating
// A/B pairs (+0.5 % at B=4, neutral at B=1).  The hypothesis below was half right: with 190-290 workgroups the launch is not bound by one
// workgroup's serial chain but by the matrix-pipe time of the ONE workgroup a CU gets (convc2: 288 workgroups x 131 k pipe cycles on 256
// CUs = two rounds of 16 us at best, whatever happens inside the workgroup) -- the split overlaps the exposed per-chunk overhead (menc.conv
// 36.7 -> 31.5 us, encoder stage 3 18.1 -> 16.4 us in isolation) but cannot spread the work over more CUs.  What the layers above 256
// workgroups need is MORE, smaller workgroups (16 tiles or 16 output channels each) -- or a bigger batch.
//
// Why: the 3x3 layers at 1/8 resolution (convc2, the motion encoder's conv, FlowHead.conv1 -- 18 launches per step -- fusion.convo,
// encoder stage 3) and stage 2 of the encoders are launches of 190-580 workgroups on a chip with 1024 slots.  Every workgroup then has
// a CU (almost) to itself, and the launch lasts as long as ONE workgroup's serial chain of Cin/8 chunk steps: a wave alone spends
// ~2.6 k cycles per chunk for 1.0 k cycles of MFMA (the hand-off barrier, the DMA issue, the LDS reads and the transform are exposed
// when no other workgroup shares the SIMD) -- 32 chunks = 41 us for convc2 whatever the chip could do.  With SK groups of four waves the
// chain is SK times shorter: group g takes chunks g, g + SK, ...; its accumulators go through the same cross-wave exchange as before
// (X[g][i][..]) and the output transform sums the groups in a fixed order (deterministic, batch slots stay bit-identical).
// Everything else -- region geometry, raw-patch layout, U layout, epilogue -- is conv_wino_kernel's (see there).
#include "conv_common.h"

namespace cf {

template <int SK>
__global__ __launch_bounds__(256 * SK, SK == 2 ? 4 : 4) void conv_wino_sk_kernel(const ConvParams p) {
    constexpr int WG_A = 4 * 2 * 32 * 32;                       // floats of one group's exchange block
    static_assert(2 * WG_RAW <= WG_A && 4 * 32 * EPI_S + 4 * 32 <= WG_A, "raw ring, epilogue patches and tables overlay the exchange blocks");
    __shared__ __attribute__((aligned(16))) float smem[SK * WG_A];        // ONE __shared__ object: see conv_wino_kernel
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all >> 2, wave = wave_all & 3;
    const int gtid = tid & 255;                                 // thread index inside the group
    float* const sRaw = smem + grp * WG_A;                      // this group's raw ring (its exchange block later)

    const int Ho = p.Ho, Wo = p.Wo;
    const int tall = wino_tall(Ho, Wo);
    const int TWr = tall ? 4 : 8;
    const int RH = tall ? 16 : 8, RW = tall ? 8 : 16;
    const int PC = RW + 2, PCh = PC >> 1;
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

    int a_pix[2];
    unsigned a_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = gtid + 256 * j;
        const int quad = sl / WG_PLANE, cell = sl - quad * WG_PLANE;
        a_q[j] = (unsigned)(quad & 1) * 16u;
        const int py = cell / PC, pc = cell - py * PC;
        const int px = pc < PCh ? 2 * pc : 2 * (pc - PCh) + 1;
        int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        bool ok = quad < 2 && cell < WG_PIX && iy <= p.Hin && ix <= p.Win;
        if (p.pad_mode == 1) {
            iy = reflect_idx(iy, p.Hin);
            ix = reflect_idx(ix, p.Win);
        } else {
            ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        }
        a_pix[j] = ok ? iy * p.Win + ix : -1;
    }
    const int nchunk = p.cin_pad / WG_KC;
    const int nstep = (nchunk + SK - 1) / SK;                   // chunk steps of a group; group g's step s is chunk s * SK + g (past the end: zeros)
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino + (long)wgroup(p, b) * p.wino_gs + (long)nblk * nchunk * WG_UV);

    // chunk iterator over the channel segments (wave-uniform); a group starts at chunk `grp` and advances SK chunks per step
    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    auto advance = [&]() __attribute__((always_inline)) {
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
    for (int i = 0; i < grp; ++i) advance();
    auto issue_raw = [&](int buf, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
        float* rbase = sRaw + buf * WG_RAW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned off = (a_pix[j] < 0 || !live) ? BUF_OOB : (unsigned)a_pix[j] * ld4 + a_q[j];
            dma16_to_lds(rs, rbase + (256 * j + 64 * wave) * 4, off, so);
        }
#pragma unroll
        for (int i = 0; i < SK; ++i) advance();
    };

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int tty = lr / TWr, ttx = lr - tty * TWr;
    const int ra = wave == 0 ? 0 : 1, rb = wave == 0 ? 2 : wave == 3 ? 3 : 2;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const int cell0 = 2 * tty * PC + ttx;
    const int rd_a = (lh * WG_PLANE + cell0 + ra * PC) * 4, rd_b = (lh * WG_PLANE + cell0 + rb * PC) * 4;
    const unsigned uoff = (unsigned)((wave * 4) * 256 + lr * WG_KC + ((lh ^ ((lr >> 3) & 1)) << 2)) * 4u;

    f32x4 bu[4];
    auto chunk_of = [&](int s) { const int c = s * SK + grp; return c < nchunk ? c : nchunk - 1; };     // dead steps re-load a valid U (times a zero patch)
    auto step = [&](int s) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // in flight, oldest first: the two raw(s) pieces, then the four U(s) loads
        raw_barrier();
        const bool more = (s + 1) * SK + grp < nchunk;
        issue_raw((s + 1) & 1, more);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 af[4];
        {
            const float* r = sRaw + (s & 1) * WG_RAW;
            f32x4 da[4], db[4], t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = ((c >> 1) + (c & 1) * PCh) * 4;
                da[c] = *reinterpret_cast<const f32x4*>(r + rd_a + col);
                db[c] = *reinterpret_cast<const f32x4*>(r + rd_b + col);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sgn, db[c][e], da[c][e]);
            af[0] = t[0] - t[2];
            af[1] = t[1] + t[2];
            af[2] = t[2] - t[1];
            af[3] = t[1] - t[3];
        }
        const unsigned u_next = (unsigned)chunk_of(s + 1) * (WG_UV * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][s2], bu[j][s2], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, u_next);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_raw(0, grp < nchunk);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, (unsigned)chunk_of(0) * (WG_UV * 4u));
    __builtin_amdgcn_sched_barrier(0);
    for (int s = 0; s < nstep; ++s) step(s);

    wait_vmcnt0();                                  // the dead past-the-end DMA of the last chunk step has landed (explicit: ADVICE r3)
    __syncthreads();                                // every wave of every group is done with its raw ring
    // ---- output transform, j direction (registers) into this group's exchange block X[g][i = wave][bcol][tile][cout] ----
    float* X = smem + grp * WG_A;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        X[((wave * 2 + 0) * 32 + trow) * 32 + lr] = (acc[0][r] + acc[1][r]) + acc[2][r];
        X[((wave * 2 + 1) * 32 + trow) * 32 + lr] = (acc[1][r] - acc[2][r]) - acc[3][r];
    }
    __syncthreads();
    // ---- group 0 finishes: sum over the groups (fixed order), i direction across the waves, one tile row per wave ----
    float yv[16];
    if (grp == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int prow = lh * 16 + q;
            const int tl = prow >> 2, a = (prow >> 1) & 1, bb = prow & 1;
            const int t = wave * 8 + tl;
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = smem[((i * 2 + bb) * 32 + t) * 32 + lr];
#pragma unroll
                for (int g = 1; g < SK; ++g) v += smem[g * WG_A + ((i * 2 + bb) * 32 + t) * 32 + lr];
                x[i] = v;
            }
            yv[q] = a == 0 ? (x[0] + x[1]) + x[2] : (x[1] - x[2]) - x[3];
        }
    }
    __syncthreads();                                // everybody has read X: the patches go on top of block 0
    if (grp != 0) return;                           // no barrier below this line
    float* sW = smem + wave * (32 * EPI_S);
    int* mtab = reinterpret_cast<int*>(smem + 4 * 32 * EPI_S) + wave * 32;
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
}

hipError_t launch_wino_sk(const ConvParams& p, int batch, hipStream_t s, int sk) {
    const int tall = wino_tall(p.Ho, p.Wo);
    const long nreg = tall ? (long)((p.Ho + 15) / 16) * ((p.Wo + 7) / 8) : (long)((p.Ho + 7) / 8) * ((p.Wo + 15) / 16);
    const long wgs = nreg * ((p.cout + 31) / 32) * batch;
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL || (sk != 2 && sk != 4)) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256 * sk;
    if (sk == 2) hipLaunchKernelGGL(conv_wino_sk_kernel<2>, dim3((unsigned)wgs), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv_wino_sk_kernel<4>, dim3((unsigned)wgs), dim3(1024), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
