// conv_wino_p.hip -- Winograd F(2x2,3x3) with PERSISTENT workgroups that walk several regions (round 4).  Launched through
// launch_conv (conv_igemm.hip) as tile 48.  gfx950 only.
//
// Why: conv_wino_kernel (tile 40) is one workgroup per (region, n-block).  Its loop is matrix-pipe-bound, but a workgroup spends
// 7-14 k cycles before its first MFMA (kernel arguments and instruction cache cold, index arithmetic, the first raw patch and U loads
// issued into a cold memory system) and 9-12 k after its last one (exchange + fused epilogue): 40 % of a cista.P workgroup's life, 21 %
// of cista.D's, 22 % of the gate convolution's (profiles/r03_stamps.txt), and a launch is 1.4-5.6 rounds of such workgroups.
//
// This kernel is the same arithmetic (bit-identical results: same products, same accumulation order, same transforms) in a grid of at
// most 1024 workgroups (4 per CU) in which a workgroup keeps its n-block and WALKS the (image, region) items of its XCD's share:
//   * kernel arguments, index tables and instruction cache are paid once per workgroup, not once per region;
//   * the chunk loop runs on across items: the LDS-DMA of the NEXT region's first raw patch and the loads of its first U block are what
//     the last chunk step of a region issues in place of conv_wino_kernel's dead past-the-end requests, so they are in flight during the
//     region's output transform / exchange / epilogue and have landed when the next region's first chunk step begins;
//   * for that the exchange buffer no longer overlays raw buffer 0 (LDS: raw0 8 KB | raw1 8 KB = first quarter of the 32 KB exchange
//     buffer; 40,960 bytes, still four workgroups per CU), and the tail's barriers are raw s_barriers (a __syncthreads() would drain the
//     prefetch);
//   * workgroups drift apart over their items, so one workgroup's tail runs beside the other three's MFMAs instead of beside their tails.
// Work split: grid = 8 x nt x R; workgroup (xcd = id & 7, i = id >> 3) has n-block i % nt and is walker i / nt of R; XCD x owns a
// contiguous run of the NR = batch x regions items (so the n-blocks of a region and neighbouring regions share an L2), walker r takes
// items r, r + R, ... of that run.
#include "conv_common.h"

namespace cf {

static constexpr int WP_A = 4 * 2 * 32 * 32;                // floats of the exchange buffer X[4][2][32][32]
static_assert(4 * 32 * EPI_S + 4 * 32 <= WP_A, "epilogue patches + row -> pixel tables fit the dead exchange buffer");

// PIPE = 1 (tile 49): the chunk loop is SOFTWARE-PIPELINED -- the A operands of chunk k + 1 are read from LDS and transformed in the
// issue slots BETWEEN the sixteen MFMAs of chunk k (af_cur / af_nxt), the raw patch is requested two chunks ahead, and nothing but the
// hand-off barrier stands between one chunk's MFMAs and the next one's.  Why: PMC + stamps of the gate convolution (r04) put the matrix
// pipe at 82 % busy over the waves' lifetime although four waves share each SIMD -- per chunk a wave spends ~1.6 k cycles (barrier, DMA
// issue, eight LDS reads, 48 VALU) in which it cannot feed the pipe, and during a workgroup's tail only three waves are left to cover
// for one another.  The pipelined loop needs 16 more registers (af_nxt) and keeps both raw buffers live across the tail (the next item's
// chunk 1 is landing): 48 KB of LDS and <= 168 VGPRs, i.e. THREE workgroups per CU -- which is enough once a wave's off-pipe time per
// chunk is a few hundred cycles.  Same arithmetic, bit for bit.
template <int PIPE>
__global__ __launch_bounds__(256, PIPE ? 3 : 4) void conv_wino_p_kernel(const ConvParams p, const int NR, const int R) {
    constexpr int WP_X0 = PIPE ? 2 * WG_RAW : WG_RAW;       // floats: the exchange buffer begins behind raw buffer 0 (PIPE: behind both)
    constexpr int WP_SMEM = WP_X0 + WP_A;                   // 40,960 bytes (PIPE: 49,152)
    static_assert(2 * WG_RAW <= WP_SMEM, "raw buffer 1 lies inside the exchange buffer");
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long st_wait = 0, st_issue = 0, st_loop = 0, st_tail = 0, st_first = 0, st_items = 0;
#endif
    // ONE __shared__ object (see conv_wino_kernel): [raw 0: 2048 floats][X: 8192 floats, raw 1 = its first 2048]; after the exchange the
    // four epilogue patches (4 x 32 x EPI_S) and, behind them, the row -> pixel tables lie over X
    __shared__ __attribute__((aligned(16))) float smem[WP_SMEM];
    float* const sRaw = smem;
    float* const X = smem + WP_X0;
    float* const sPatch = X;
    int* const sMtab = reinterpret_cast<int*>(X + 4 * 32 * EPI_S);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Ho = p.Ho, Wo = p.Wo;
    const int tall = wino_tall(Ho, Wo);
    const int TWr = tall ? 4 : 8;                            // tile columns of a region (tile rows = 32 / TWr)
    const int RH = tall ? 16 : 8, RW = tall ? 8 : 16;        // region size in output pixels
    const int PC = RW + 2, PCh = PC >> 1;                    // patch columns (rows = RH + 2; PC * (RH + 2) = 180 either way)
    const int nrx = (Wo + RW - 1) / RW, nry = (Ho + RH - 1) / RH;
    const int nreg = nrx * nry;
    const int nt = (p.cout + 31) / 32;
    // ---- this workgroup's n-block and its walk over the XCD's items ----
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int nblk = wi % nt, walker = wi / nt;
    const int q8 = NR >> 3, r8 = NR & 7;
    const int s_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt_x = q8 + (xcd < r8 ? 1 : 0);
    if (walker >= cnt_x) return;                             // (the launcher sizes R so that this does not happen beyond the ragged XCDs)
    const int n0 = nblk * 32;
    const int nchunk = p.cin_pad / WG_KC;

    // lane index, re-derived wherever it is needed outside the chunk loop (two VALU instructions) instead of being held in a register
    // across it: this kernel sits at the 128-register limit of four waves per SIMD
    // (volatile asm: the builtin form is hoisted out of the item loop as an invariant and spilled)
    auto lane_now = [&]() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    int a_pix[2];
    unsigned a_q[2];
    // chunk iterator over the channel segments (wave-uniform)
    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    int it_b = 0;                                            // image the iterator reads
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino);

    // item `it` of this XCD's run -> image, region origin, byte offset of its first U block; re-arms the raw-patch offsets and the iterator
    auto setup = [&](int it, int& o_b, int& o_reg, int& o_oy0, int& o_ox0, unsigned& o_ubase) __attribute__((always_inline)) {
        const int id = s_x + it;
        const int b = id / nreg, reg = id - b * nreg;
        const int ry = reg / nrx, rx = reg - ry * nrx;
        const int oy0 = ry * RH, ox0 = rx * RW;
        // raw patch DMA slots: slot s = tid + 256 j -> channel quad s / 192, patch cell s % 192 (180 live; quad 2 = the dead slots 384..511);
        // cells of a patch row are stored even columns first, then odd columns.  cell / PC by multiply-shift (PC = 10 or 18, cell < 192)
        const int tid_ = wave * 64 + lane_now();    // (not an item-loop invariant for the optimiser: the decode below is redone per item)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int sl = tid_ + 256 * j;
            const int quad = sl >= 2 * WG_PLANE ? 2 : (sl >= WG_PLANE ? 1 : 0), cell = sl - quad * WG_PLANE;
            const int py = tall ? (cell * 205) >> 11 : (cell * 57) >> 10, pc = cell - py * PC;
            const int px = pc < PCh ? 2 * pc : 2 * (pc - PCh) + 1;
            a_q[j] = (unsigned)(quad & 1) * 16u;
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
        it_seg = 0;
        it_cs = 0;
        it_b = b;
        seg_base = p.in[0] + (long)b * p.seg_bs[0];
        seg_ld = p.seg_ld[0];
        seg_cn = p.seg_c[0];
        o_b = b; o_reg = reg; o_oy0 = oy0; o_ox0 = ox0;
        o_ubase = (unsigned)(((long)wgroup(p, b) * p.wino_gs + (long)nblk * nchunk * WG_UV) * 4L);
    };
    // Every wave issues exactly two DMA instructions per chunk, unconditionally (dead slots and the chunk past the very end fetch out
    // of range = zeros into unused LDS): with a fixed count the compiler's s_waitcnt vmcnt before each position's MFMAs waits for that
    // position's U registers only, not for the raw patch behind them in the queue.
    auto issue_raw = [&](int buf, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
        float* rbase = sRaw + buf * WG_RAW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // 24-bit multiply (one v_mad_u32_u24): pixels per image < 2^24 and the pixel stride < 2^24 bytes (wino_p_ok)
            const unsigned off = (a_pix[j] < 0 || !live) ? BUF_OOB : __umul24((unsigned)a_pix[j], ld4) + a_q[j];
            dma16_to_lds(rs, rbase + (256 * j + 64 * wave) * 4, off, so);
        }
        it_cs += WG_KC;
        if (it_cs >= seg_cn) {
            it_cs = 0;
            ++it_seg;
            if (it_seg < p.nseg) {
                seg_base = sel3(p.in, it_seg) + (long)it_b * (it_seg == 1 ? p.seg_bs[1] : p.seg_bs[2]);
                seg_ld = it_seg == 1 ? p.seg_ld[1] : p.seg_ld[2];
                seg_cn = it_seg == 1 ? p.seg_c[1] : p.seg_c[2];
            }
        }
    };

    f32x16 acc[4];
    // lane (lr, lh): tile lr = (ty, tx) of the region, channel quad lh; its patch rows: ra, rb of {0,2} {1,2} {1,2} {1,3} for wave 0..3
    const int lane0 = lane_now();
    const int lr = lane0 & 31, lh = lane0 >> 5;
    const int tty = tall ? lr >> 2 : lr >> 3, ttx = lr & (TWr - 1);
    const int ra = wave == 0 ? 0 : 1, rb = wave == 0 ? 2 : wave == 3 ? 3 : 2;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const int cell0 = 2 * tty * PC + ttx;                                             // cell of patch pixel (2 ty, 2 tx)
    const int rd_a = (lh * WG_PLANE + cell0 + ra * PC) * 4, rd_b = (lh * WG_PLANE + cell0 + rb * PC) * 4;   // floats; + column cell * 4
    const unsigned uoff = (unsigned)((wave * 4) * 256 + lr * WG_KC + ((lh ^ ((lr >> 3) & 1)) << 2)) * 4u;  // + j KiB: position (wave, j) of a chunk's U block

    f32x4 bu[4];
    // one chunk: hand-off of raw(k); LDS-DMA of the next raw patch; this wave's row of the transform; 16 MFMAs, position by position, each
    // position's U registers refilled for the next chunk step as soon as its four MFMAs are issued (a whole chunk of lead)
    auto chunk_step = [&](int k, bool more, unsigned u_next) __attribute__((always_inline)) {
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        // in flight, oldest first: the two raw(k) pieces, then the four U(k) loads -- raw(k) has landed once at most four are outstanding
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        raw_barrier();                              // raw(k) has landed for everybody; everybody has read raw(k-1) / finished the previous item's tail
#ifdef CF_STAMP
        const long long t1 = __builtin_readcyclecounter();
#endif
        issue_raw((k + 1) & 1, more);
        __builtin_amdgcn_sched_barrier(0);          // raw before U in issue order: the vmcnt(4) above counts on it
        f32x4 af[4];
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
#ifdef CF_STAMP
        wait_lgkm0();
        const long long t2 = __builtin_readcyclecounter();
        st_wait += t1 - t0;
        st_issue += t2 - t1;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][s2], bu[j][s2], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, u_next);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- PIPE: one chunk step = sixteen slots of {one MFMA of chunk k; a slice of the work for chunk k + 1 / k + 2}, pinned in this order ----
    f32x4 af_cur[4], af_nxt[4];
    auto transform_now = [&](int buf, f32x4 (&af)[4]) __attribute__((always_inline)) {      // prologue only: raw(0) -> operands of chunk 0
        const float* r = sRaw + buf * WG_RAW;
        f32x4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = ((c >> 1) + (c & 1) * PCh) * 4;
            const f32x4 da = *reinterpret_cast<const f32x4*>(r + rd_a + col), db = *reinterpret_cast<const f32x4*>(r + rd_b + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[c][e] = __builtin_fmaf(sgn, db[e], da[e]);
        }
        af[0] = t[0] - t[2];
        af[1] = t[1] + t[2];
        af[2] = t[2] - t[1];
        af[3] = t[1] - t[3];
    };
    // k: chunk whose MFMAs run; live: the chunk two steps ahead exists (in this item or the next); u_next: byte offset of the U block one step ahead
    auto pchunk_step = [&](int k, bool live, unsigned u_next) __attribute__((always_inline)) {
        const float* r = sRaw + ((k + 1) & 1) * WG_RAW;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned ld4 = (unsigned)seg_ld * 4u, so = (unsigned)it_cs * 4u;
        float* rbase = sRaw + (k & 1) * WG_RAW;        // raw(k + 2) goes where raw(k) was
        f32x4 da, db, t[4];
#define WP_SLOT(j, s2, ...)                                                                                   \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_cur[j][s2], bu[j][s2], acc[j], 0, 0, 0);             \
        __VA_ARGS__;                                                                                          \
        __builtin_amdgcn_sched_barrier(0)
#define WP_DMA(jj)                                                                                            \
        dma16_to_lds(rs, rbase + (256 * jj + 64 * wave) * 4,                                                  \
                     (a_pix[jj] < 0 || !live) ? BUF_OOB : __umul24((unsigned)a_pix[jj], ld4) + a_q[jj], so)
#define WP_RD(c)                                                                                              \
        da = *reinterpret_cast<const f32x4*>(r + rd_a + ((c >> 1) + (c & 1) * PCh) * 4);                     \
        db = *reinterpret_cast<const f32x4*>(r + rd_b + ((c >> 1) + (c & 1) * PCh) * 4)
        // (pure arithmetic carries no ordering: without the empty asm the compiler sinks the whole transform behind the last MFMA)
#define WP_PIN(x) asm volatile("" : "+v"(x))
#define WP_T(c)                                                                                               \
        t[c][0] = __builtin_fmaf(sgn, db[0], da[0]); t[c][1] = __builtin_fmaf(sgn, db[1], da[1]);            \
        t[c][2] = __builtin_fmaf(sgn, db[2], da[2]); t[c][3] = __builtin_fmaf(sgn, db[3], da[3]); WP_PIN(t[c])
        // The MFMAs only need registers (af_cur, bu), so position 0's four run BEFORE the hand-off barrier: 256 cycles of matrix-pipe work
        // during which the workgroup's other waves arrive (a wave that waits at the barrier while its SIMD's pipe has a free slot is the
        // bubble this removes); everything that touches the raw buffers comes behind the barrier.
        WP_SLOT(0, 0, (void)0);
        WP_SLOT(0, 1, (void)0);
        WP_SLOT(0, 2, (void)0);
        WP_SLOT(0, 3, (void)0);
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        // in flight, oldest first: the two pieces of raw(k + 1) (issued one step earlier), then the four U(k) loads
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        raw_barrier();                              // raw(k + 1) has landed for everybody; everybody has read raw(k) (one step earlier) / finished its tail
        __builtin_amdgcn_sched_barrier(0);
#ifdef CF_STAMP
        const long long t1 = __builtin_readcyclecounter();
        st_wait += t1 - t0;
#endif
        WP_SLOT(1, 0, WP_DMA(0));
        WP_SLOT(1, 1, WP_DMA(1));
        // (a slot of its own: inside one slot the compiler is free to put this load BETWEEN the two DMA pieces, and the vmcnt(4) at the
        // top of the next step -- "the two raw pieces are the oldest two of six" -- then no longer covers the second piece: a race that
        // the single-operator tests did not show and the model's determinism soak did, r04)
        bu[0] = buf_load4(u_rsrc, uoff, u_next);
        __builtin_amdgcn_sched_barrier(0);
        WP_SLOT(1, 2, WP_RD(0));
        WP_SLOT(1, 3, bu[1] = buf_load4(u_rsrc, uoff + 1024u, u_next));
        WP_SLOT(2, 0, WP_T(0); WP_RD(1));
        WP_SLOT(2, 1, WP_T(1); WP_RD(2));
        WP_SLOT(2, 2, WP_T(2); WP_RD(3));
        WP_SLOT(2, 3, WP_T(3); bu[2] = buf_load4(u_rsrc, uoff + 2048u, u_next));
        WP_SLOT(3, 0, af_nxt[0] = t[0] - t[2]; WP_PIN(af_nxt[0]));
        WP_SLOT(3, 1, af_nxt[1] = t[1] + t[2]; WP_PIN(af_nxt[1]));
        WP_SLOT(3, 2, af_nxt[2] = t[2] - t[1]; WP_PIN(af_nxt[2]));
        WP_SLOT(3, 3, af_nxt[3] = t[1] - t[3]; WP_PIN(af_nxt[3]); bu[3] = buf_load4(u_rsrc, uoff + 3072u, u_next));
#undef WP_SLOT
#undef WP_DMA
#undef WP_RD
#undef WP_T
#undef WP_PIN
#pragma unroll
        for (int j = 0; j < 4; ++j) af_cur[j] = af_nxt[j];
        // advance the raw-patch iterator (it names chunk k + 3 now)
        it_cs += WG_KC;
        if (it_cs >= seg_cn) {
            it_cs = 0;
            ++it_seg;
            if (it_seg < p.nseg) {
                seg_base = sel3(p.in, it_seg) + (long)it_b * (it_seg == 1 ? p.seg_bs[1] : p.seg_bs[2]);
                seg_ld = it_seg == 1 ? p.seg_ld[1] : p.seg_ld[2];
                seg_cn = it_seg == 1 ? p.seg_c[1] : p.seg_c[2];
            }
        }
#ifdef CF_STAMP
        st_issue += __builtin_readcyclecounter() - t1;
#endif
    };

    int item = walker;
    int c_b, c_reg, c_oy0, c_ox0;
    unsigned u_base;
    setup(item, c_b, c_reg, c_oy0, c_ox0, u_base);
    issue_raw(0, true);
    if constexpr (PIPE) issue_raw(1, true);             // (nchunk >= 2: wino_p_ok)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, u_base);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PIPE) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // raw(0) has landed (raw(1) and U(0) may still be on their way)
        raw_barrier();
        transform_now(0, af_cur);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef CF_STAMP
    st_first = __builtin_readcyclecounter() - t_begin;
#endif

    for (;;) {
#ifdef CF_STAMP
        const long long tl0 = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const int nxt = item + R;
        const bool has_next = nxt < cnt_x;
        int n_b = c_b, n_reg = c_reg, n_oy0 = c_oy0, n_ox0 = c_ox0;
        unsigned n_ubase = u_base;
        if constexpr (PIPE) {
            // steps 0 .. nchunk - 3 request raw patches of THIS item (two ahead); the last two request the next item's chunks 0 and 1, and
            // the very last one transforms the next item's chunk 0 and loads its U(0): the chunk loop runs on across the tail
            for (int k = 0; k < nchunk - 2; ++k) pchunk_step(k, true, u_base + (unsigned)(k + 1) * (WG_UV * 4u));
            if (has_next) setup(nxt, n_b, n_reg, n_oy0, n_ox0, n_ubase);
            pchunk_step(nchunk - 2, has_next, u_base + (unsigned)(nchunk - 1) * (WG_UV * 4u));
            pchunk_step(nchunk - 1, has_next, has_next ? n_ubase : u_base + (unsigned)(nchunk - 1) * (WG_UV * 4u));
        } else {
        for (int k = 0; k < nchunk - 1; ++k) chunk_step(k, true, u_base + (unsigned)(k + 1) * (WG_UV * 4u));
        // the last chunk step of the item requests the NEXT item's first raw patch and U block (none left: dead requests, as in conv_wino_kernel)
        if (has_next) setup(nxt, n_b, n_reg, n_oy0, n_ox0, n_ubase);
        chunk_step(nchunk - 1, has_next, has_next ? n_ubase : u_base + (unsigned)(nchunk - 1) * (WG_UV * 4u));
        }
#ifdef CF_STAMP
        const long long tl1 = __builtin_readcyclecounter();
        st_loop += tl1 - tl0;
#endif

        // ---- tail of the current item (c_*).  Raw barriers: the next item's raw patch is landing in raw buffer 0 meanwhile.  The lane /
        // wave indices go through an empty asm first: the tail's LDS and table addresses are item-invariant, and hoisted out of the item
        // loop they would be spilled to scratch across the MFMA loop (37 VGPR spills without this) ----
        int t_lane = lane_now(), t_wave = wave;
        asm volatile("" : "+v"(t_lane), "+s"(t_wave));
        const int t_lr = t_lane & 31, t_lh = t_lane >> 5;
        wait_lgkm0();
        raw_barrier();                              // every wave is done with raw buffer 1 before it becomes part of the exchange buffer
        // output transform, j direction (in registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * t_lh;
            X[((t_wave * 2 + 0) * 32 + trow) * 32 + t_lr] = (acc[0][r] + acc[1][r]) + acc[2][r];
            X[((t_wave * 2 + 1) * 32 + trow) * 32 + t_lr] = (acc[1][r] - acc[2][r]) - acc[3][r];
        }
        wait_lgkm0();
        raw_barrier();
        // i direction across the waves + patch of this wave's tile row: 8 tiles x (2 x 2) pixels x 32 couts
        float* sW = sPatch + t_wave * (32 * EPI_S);
        int* mtab = sMtab + t_wave * 32;
        float yv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int prow = t_lh * 16 + q;             // patch row = tl * 4 + a * 2 + bb
            const int tl = prow >> 2, a = (prow >> 1) & 1, bb = prow & 1;
            const int t = t_wave * 8 + tl;
            const float x0 = X[((0 * 2 + bb) * 32 + t) * 32 + t_lr], x1 = X[((1 * 2 + bb) * 32 + t) * 32 + t_lr];
            const float x2 = X[((2 * 2 + bb) * 32 + t) * 32 + t_lr], x3 = X[((3 * 2 + bb) * 32 + t) * 32 + t_lr];
            yv[q] = a == 0 ? (x0 + x1) + x2 : (x1 - x2) - x3;
        }
        wait_lgkm0();
        raw_barrier();                              // everybody has read X: the patches go on top of it
#pragma unroll
        for (int q = 0; q < 16; ++q) sW[(t_lh * 16 + q) * EPI_S + t_lr] = yv[q];
        if (t_lane < 32) {
            const int tl = t_lane >> 2, a = (t_lane >> 1) & 1, bb = t_lane & 1;
            const int t = t_wave * 8 + tl, ty = tall ? t >> 2 : t >> 3, tx = t & (TWr - 1);
            const int oy = c_oy0 + 2 * ty + a, ox = c_ox0 + 2 * tx + bb;
            mtab[t_lane] = (oy < Ho && ox < Wo) ? oy * Wo + ox : -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // the tail reads its parameters (output / aux tensors, bias, epilogue kind ...) from the kernarg segment HERE, through a
            // pointer the optimiser cannot see through: hoisted out of the item loop they would sit in SGPRs across the MFMA loop
            unsigned long long kv = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();      // ConvParams is the first argument
            asm volatile("" : "+s"(kv));
            KernargParams& tp = *reinterpret_cast<KernargParams*>(kv);
            patch_tail(tp, sW, c_b, 0, n0, t_lane, Ho * Wo, 0, 4, 1, 0, mtab);
            if (tp.st_partial) patch_stats(tp, sW, c_b, 0, n0, t_lane, Ho * Wo, mtab, c_reg * 4 + t_wave, nreg * 4);
        }
#ifdef CF_STAMP
        st_tail += __builtin_readcyclecounter() - tl1;
        ++st_items;
#endif
        if (!has_next) break;
        // Everything this wave has in flight (the prefetched raw patch and U block, the epilogue's stores) is waited for HERE, with the
        // builtin the compiler's waitcnt pass understands: the chunk loop is then entered with an empty scoreboard, as from the prologue,
        // and its steady-state counts (vmcnt(5) in front of a position's MFMAs) are not weakened by whatever the epilogue left pending
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched
        item = nxt;
        c_b = n_b; c_reg = n_reg; c_oy0 = n_oy0; c_ox0 = n_ox0;
        u_base = n_ubase;
    }
#ifdef CF_STAMP
    if (p.stamp && lane_now() == 0) {      // [DMA wait + barrier, items, issue + transform, begin -> first loop, chunks (all items), loops, tails, MHz]
        long long* q = p.stamp + ((long)blockIdx.x * 4 + wave) * 8;
        q[0] = st_wait; q[1] = st_items; q[2] = st_issue; q[3] = st_first; q[4] = (long long)nchunk * st_items;
        q[5] = st_loop; q[6] = st_tail;
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
}

bool wino_p_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || (p.prec != 0 && p.prec != 3) || !p.w_wino || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.padT != 1 || p.padL != 1) return false;
    if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin < 12 || p.Win < 12) return false;
    if (p.w_bs != 0 && p.w_div <= 1) return false;          // per-image matrices (correlation GEMM); weight groups are fine
    if (p.cin_pad < 2 * WG_KC || (long)p.Hin * p.Win >= (1L << 24)) return false;
    for (int i = 0; i < p.nseg; ++i)
        if ((long)p.seg_ld[i] * 4 >= (1L << 24)) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg_c[i] % WG_KC) return false;
    return dma_range_ok(p);
}

// walkers per (XCD, n-block): as many as fit 4 workgroups per CU (CF_WINOP_SLOTS workgroups in all), no more than an XCD has items
int wino_p_walkers(const ConvParams& p, long NR, int pipe) {
    const char* es = getenv("CF_WINOP_SLOTS");              // read per launch: the tests shrink it to make every walker carry several items
    const long slots = es ? atol(es) : (pipe ? 768 : 1024);     // workgroups the chip holds at once (3 / 4 per CU)
    const long nt = (p.cout + 31) / 32;
    long R = slots / (8 * nt);
    if (R < 1) R = 1;
    const long per_xcd = (NR + 7) / 8;
    if (R > per_xcd) R = per_xcd;
    return (int)R;
}

hipError_t launch_wino_p(const ConvParams& p, int batch, hipStream_t s, int pipe) {
    if (!wino_p_ok(p)) return hipErrorInvalidValue;
    const long nreg = wino_tall(p.Ho, p.Wo) ? (long)((p.Ho + 15) / 16) * ((p.Wo + 7) / 8) : (long)((p.Ho + 7) / 8) * ((p.Wo + 15) / 16);
    const long NR = nreg * batch;
    const long nt = (p.cout + 31) / 32;
    if (NR <= 0 || NR >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    // U blocks are addressed with 32-bit byte offsets from the first weight group's matrix
    const long last_group = p.w_div > 1 ? (batch - 1) / p.w_div : batch - 1;        // wgroup() of the last image
    if (p.wino_gs < 0) return hipErrorInvalidValue;
    const long ubytes = (last_group * p.wino_gs + nt * (p.cin_pad / WG_KC) * WG_UV) * 4L;
    if (ubytes >= 0x7FFFFF00L) return hipErrorInvalidValue;
    const int R = wino_p_walkers(p, NR, pipe);
    const long wgs = 8 * nt * R;
    if (wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256;
    if (pipe) hipLaunchKernelGGL(conv_wino_p_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, p, (int)NR, R);
    else hipLaunchKernelGGL(conv_wino_p_kernel<0>, dim3((unsigned)wgs), dim3(256), 0, s, p, (int)NR, R);
    return hipGetLastError();
}

}  // namespace cf
