// conv_wino16.hip -- Winograd F(2x2,3x3) in SMALL workgroups for the low-resolution layers (round 3).  Launched through launch_conv
// (conv_igemm.hip) as tile 47.  gfx950 only.
//
// Why: at 1/8 and 1/4 resolution conv_wino_kernel's workgroups (32 tiles x 32 output channels = 128 pixels) number 190-580 for 256 CUs, so a
// launch lasts as long as the matrix-pipe time of the MOST loaded CU -- convc2: 288 workgroups = two on 32 CUs, one on the rest, 32 us of
// pipe time where the chip-wide figure is 18 (DESIGN.md sections 9, 10; the GRU's F(2,5) kernel showed what small workgroups buy).  This
// kernel is conv_wino_kernel with HALF the tiles per workgroup on v_mfma_f32_16x16x4_f32 (same flop rate as 32x32x2, 4 accumulator
// registers per 16 x 16 block): twice as many workgroups of half the work.
//   region  4 x 4 tiles = 8 x 8 output pixels, patch 10 x 10 pixels; 32 output channels; K in chunks of 8 channels;
//   raw     [2 channel quads][128 cells (100 live), even patch columns first] x 16 bytes = 4 KB per chunk, LDS-DMA, ONE instruction per
//           wave and chunk, ring of two;
//   MFMA    wave w owns positions (w, 0..3): 4 positions x 2 channel halves = 8 accumulators of 4 registers; per chunk 16 MFMAs of 32
//           cycles; lane (tile l & 15, k slot l >> 4) reads 2 rows x 4 pixels x 2 k-steps as single floats (ds_read_b32, two-way bank
//           conflicts at worst) and computes its own 8 A operands;
//   U       = G g G^T in THIS kernel's order [n-block][chunk][pos][lane][half * 2 + k-step] (launch_wino16_weights): one 16-byte load per
//           lane and position, a chunk ahead;
//   tail    as conv_wino_kernel: j direction in registers, i direction across the waves through LDS, then the 64 x 32 output block as two
//           32-row patches through the common fused epilogue (wave w finishes 16 rows of patch w & 1).
//
// DEEP = 1 (tile 50, round 4): the same arithmetic for launches that put at most ONE workgroup on a CU (B = 1 / 2 at 1/8 resolution: 24-144
// workgroups).  There a wave has its SIMD to itself and nothing hides what the plain loop leaves exposed: the stamps of the plain kernel at
// B = 1 (profiles/r04_small_batch.txt) say 1094 cycles per chunk for 512 cycles of MFMAs with the raw patch long landed (50 cycles of wait) --
// the U loads issued one chunk (~1100 cycles = 470 ns at the 2.33 GHz these launches hold) ahead are an L2 round trip ahead, no more, and the 16
// ds_read_b32 + the transform sit in front of the first MFMA.  So: U TWO chunks ahead (two register sets, the chunk loop unrolled by two), the raw
// patch three chunks ahead (ring of four = the 16 KB the exchange buffer has anyway), and a software-pipelined A side -- step k reads raw(k + 1) and
// builds chunk k + 1's operands between the MFMAs of chunk k.  ~125 registers more than matter at four workgroups per CU (11.2 of DESIGN.md: the
// pipelined loop alone bought nothing at B = 8), which is why it is a separate instantiation with launch_bounds(256, 2).
// Bit-identical to DEEP = 0 (same products, same accumulation order): tests/test_ops_gpu.py.
#include "conv_common.h"

namespace cf {

static constexpr int W6_KC = 8;
static constexpr int W6_PLANE = 128;                        // 16-byte slots per channel-quad plane (100 live)
static constexpr int W6_RAW = 2 * W6_PLANE * 4;             // floats per raw buffer (4 KB)
static constexpr int W6_UV = 16 * 64 * 4;                   // floats of a chunk's U block (4096: 16 pos x 32 n x 8 k)
static constexpr int W6_XS = 20;                            // exchange row = 16 tiles of one cout, padded to 20 floats: 16-byte accesses of 16 lanes hit 64 different banks
static constexpr int W6_X = 4 * 2 * 32 * W6_XS;             // floats of the exchange buffer X[i][bcol][cout 32][tile 16 (+4)] (20 KB)

template <int DEEP>
__global__ __launch_bounds__(256, DEEP ? 2 : 4) void conv_wino16_kernel(const ConvParams p) {
    static_assert(4 * W6_RAW <= W6_X && 2 * 32 * EPI_S <= W6_X, "raw ring (two buffers, four when DEEP) and the two epilogue patches overlay the exchange buffer");
    __shared__ __attribute__((aligned(16))) float smem[W6_X + 64];       // ONE __shared__ object (see conv_wino_kernel)
    float* const sRaw = smem;
    int* const sMtab = reinterpret_cast<int*>(smem + W6_X);
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long st_wait = 0, t_loop_begin = 0, t_loop_end = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = p.Ho, Wo = p.Wo;
    const int nrx = (Wo + 7) >> 3, nry = (Ho + 7) >> 3;
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
    const int oy0 = (reg / nrx) * 8, ox0 = (reg % nrx) * 8;
    const int n0 = nblk * 32;

    // ---- raw patch DMA: slot tid -> quad tid >> 7, cell tid & 127; a patch row is stored even columns first, then odd ----
    int a_pix;
    {
        const int cell = tid & 127;
        const int py = cell / 10, pc = cell - py * 10;
        const int px = pc < 5 ? 2 * pc : 2 * (pc - 5) + 1;
        int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        bool ok = cell < 100 && iy <= p.Hin && ix <= p.Win;              // beyond the halo of the last row / column: unused
        if (p.pad_mode == 1) {
            iy = reflect_idx(iy, p.Hin);
            ix = reflect_idx(ix, p.Win);
        } else {
            ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        }
        a_pix = ok ? iy * p.Win + ix : -1;
    }
    const unsigned a_q = (unsigned)(tid >> 7) * 16u;
    const int nchunk = p.cin_pad / W6_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino16 + (long)wgroup(p, b) * p.wino16_gs + (long)nblk * nchunk * W6_UV);

    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    // one DMA instruction per wave and chunk, unconditionally (dead slots and the chunk past the end fetch out of range = zeros)
    auto issue_raw = [&](int buf, bool live) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
        const unsigned off = (a_pix < 0 || !live) ? BUF_OOB : (unsigned)a_pix * ((unsigned)seg_ld * 4u) + a_q;
        dma16_to_lds(rs, sRaw + buf * W6_RAW + 64 * wave * 4, off, (unsigned)it_cs * 4u);
        it_cs += W6_KC;
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

    f32x4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) acc[j][hh] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane (tile t16 = (ty, tx), k slot kk): patch rows ra, rb of {0,2} {1,2} {1,2} {1,3} for wave 0..3 (conv_wino_kernel's formulas)
    const int t16 = lane & 15, kk = lane >> 4;
    const int ty = t16 >> 2, tx = t16 & 3;
    const int ra = wave == 0 ? 0 : 1, rb = wave == 0 ? 2 : wave == 3 ? 3 : 2;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const int cell0 = 2 * ty * 10 + tx;                                     // cell of patch pixel (2 ty, 2 tx)
    const int rd_a = (cell0 + ra * 10) * 4 + kk, rd_b = (cell0 + rb * 10) * 4 + kk;      // floats; + quad plane s * 512, + column cell * 4
    const unsigned uoff = (unsigned)((wave * 4) * 64 + lane) * 16u;         // + j KiB: position (wave, j) of a chunk's U block

    if constexpr (DEEP) {
        // ---- U two chunks ahead, raw three chunks ahead (ring of four), A operands of chunk k + 1 built between the MFMAs of chunk k ----
        // ONE wave per SIMD issues in order: whatever is not placed BETWEEN two MFMAs runs while the matrix pipe idles (the first deep
        // version, MFMAs in blocks of four and everything else in front of them: 968 cycles per chunk for 512 of MFMA).  So a step is sixteen
        // hand-placed slots, one MFMA each, fenced by sched_barriers: slot 0 / 1 the raw DMA of chunk k + 3 (offset, then the request),
        // 2..5 the sixteen LDS reads of raw(k + 1), 6 / 7 the segment iterator, 10..13 the transform (reads long back), U(k + 2) for position j
        // behind j's last MFMA.  The segment iterator is branch-free here (selects over three precomputed descriptors): a branch would end the
        // scheduling region in the middle of the slots.
        // (pick(): by-value arguments -- `c ? x : y` of two captured variables is an LVALUE conditional, i.e. a select of two ADDRESSES inside the
        // closure object, which keeps the closure, every variable it captures and a copy of the parameter block in scratch)
        auto pick = [](bool c, auto x, auto y) __attribute__((always_inline)) { return c ? x : y; };
        const int nseg = p.nseg;
        const float* const sb1 = nseg > 1 ? p.in[1] + (long)b * p.seg_bs[1] : seg_base;
        const float* const sb2 = nseg > 2 ? p.in[2] + (long)b * p.seg_bs[2] : seg_base;
        const int ld1 = p.seg_ld[1], ld2 = p.seg_ld[2], cn1 = p.seg_c[1], cn2 = p.seg_c[2];
        auto advance = [&]() __attribute__((always_inline)) {
            it_cs += W6_KC;
            const bool adv = it_cs >= seg_cn;
            it_cs = pick(adv, 0, it_cs);
            it_seg += pick(adv, 1, 0);
            const bool upd = adv && it_seg < nseg;
            const bool one = it_seg == 1;
            seg_base = pick(upd, pick(one, sb1, sb2), seg_base);
            seg_ld = pick(upd, pick(one, ld1, ld2), seg_ld);
            seg_cn = pick(upd, pick(one, cn1, cn2), seg_cn);
        };
        auto raw_off = [&](bool live) __attribute__((always_inline)) {
            return (a_pix < 0 || !live) ? BUF_OOB : (unsigned)a_pix * ((unsigned)seg_ld * 4u) + a_q;
        };
        auto raw_dma = [&](int buf, unsigned off) __attribute__((always_inline)) {
            dma16_to_lds(make_rsrc(seg_base), sRaw + buf * W6_RAW + 64 * wave * 4, off, (unsigned)it_cs * 4u);
        };
        f32x4 bu[2][4];
        float afc[4][2];                                // A operands of the chunk whose MFMAs run in this step
        auto transform = [&](const float (&t)[2][4], float (&af)[4][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                af[0][s] = t[s][0] - t[s][2];
                af[1][s] = t[s][1] + t[s][2];
                af[2][s] = t[s][2] - t[s][1];
                af[3][s] = t[s][1] - t[s][3];
            }
        };
        auto load_u = [&](int chunk, f32x4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = buf_load4(u_rsrc, uoff + 1024u * j, (unsigned)chunk * (W6_UV * 4u));
        };
#define W6_FENCE __builtin_amdgcn_sched_barrier(0)
#define W6_PIN(x) asm volatile("" : "+v"(x))
        // MFMA m of a step: position m >> 2; within a position the two accumulators alternate (a0 b0 | a0 b2 | a1 b1 | a1 b3)
#define W6_MFMA(m)                                                                                                                          \
        acc[(m) >> 2][(m) & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(afc[(m) >> 2][((m) >> 1) & 1], u[(m) >> 2][(((m) & 1) << 1) | (((m) >> 1) & 1)], \
                                                                      acc[(m) >> 2][(m) & 1], 0, 0, 0)
        // raw values of patch column c (both rows) of k-step s: two LDS reads
#define W6_RD(s_, c_)                                                                                                   \
        {                                                                                                               \
            const int col = (((c_) >> 1) + ((c_) & 1) * 5) * 4 + (s_) * (W6_PLANE * 4);                                 \
            va[s_][c_] = r[rd_a + col];                                                                                 \
            vb[s_][c_] = r[rd_b + col];                                                                                 \
        }
        // in flight at the top of step k, oldest first: raw(k+1) | U(k) x 4 | raw(k+2) | U(k+1) x 4 (the prologue's order differs, its count
        // does not): raw(k+1) has landed once at most NINE are outstanding
        auto deep_step = [&](int k, f32x4 (&u)[4]) __attribute__((always_inline)) {
#ifdef CF_STAMP
            const long long t0 = __builtin_readcyclecounter();
#endif
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            raw_barrier();                              // raw(k+1) has landed for everybody; everybody has read raw(k) (in step k - 1)
#ifdef CF_STAMP
            st_wait += __builtin_readcyclecounter() - t0;
#endif
            const float* r = sRaw + ((k + 1) & 3) * W6_RAW;
            // (with every U load aimed at ONE hot 16 KB block the step took 900 instead of 915 cycles: the loop does not wait for U)
            const unsigned u_next = (unsigned)(k + 2 < nchunk ? k + 2 : k) * (W6_UV * 4u);      // past the end: a harmless re-load
            float va[2][4], vb[2][4], t[2][4], afn[4][2];
            W6_FENCE;
            W6_MFMA(0);  unsigned off = raw_off(k + 3 < nchunk); W6_PIN(off);                                   W6_FENCE;
            W6_MFMA(1);  raw_dma((k + 3) & 3, off);             /* over raw(k-1), read in step k - 2 */         W6_FENCE;
            W6_MFMA(2);  W6_RD(0, 0); W6_RD(0, 2);                                                              W6_FENCE;
            W6_MFMA(3);  W6_RD(0, 1); W6_RD(0, 3);                                                              W6_FENCE;
            u[0] = buf_load4(u_rsrc, uoff, u_next);             /* raw(k+3) before U(k+2) in issue order */     W6_FENCE;
            W6_MFMA(4);  W6_RD(1, 0); W6_RD(1, 2);                                                              W6_FENCE;
            W6_MFMA(5);  W6_RD(1, 1); W6_RD(1, 3);                                                              W6_FENCE;
            W6_MFMA(6);  advance();                                                                             W6_FENCE;
            W6_MFMA(7);                                                                                         W6_FENCE;
            u[1] = buf_load4(u_rsrc, uoff + 1024u, u_next);                                                     W6_FENCE;
            W6_MFMA(8);                                                                                         W6_FENCE;
            W6_MFMA(9);                                                                                         W6_FENCE;
            W6_MFMA(10);                                                                                        W6_FENCE;
#pragma unroll
            for (int c = 0; c < 4; ++c) { t[0][c] = __builtin_fmaf(sgn, vb[0][c], va[0][c]); W6_PIN(t[0][c]); }      // row `wave` of B^T d
            W6_FENCE;
            W6_MFMA(11);                                                                                        W6_FENCE;
            u[2] = buf_load4(u_rsrc, uoff + 2048u, u_next);                                                     W6_FENCE;
#pragma unroll
            for (int c = 0; c < 4; ++c) { t[1][c] = __builtin_fmaf(sgn, vb[1][c], va[1][c]); W6_PIN(t[1][c]); }
            W6_FENCE;
            W6_MFMA(12);                                                                                        W6_FENCE;
            transform(t, afn);
#pragma unroll
            for (int j = 0; j < 4; ++j) { W6_PIN(afn[j][0]); W6_PIN(afn[j][1]); }
            W6_FENCE;
            W6_MFMA(13);                                                                                        W6_FENCE;
            W6_MFMA(14);                                                                                        W6_FENCE;
            W6_MFMA(15);                                                                                        W6_FENCE;
            u[3] = buf_load4(u_rsrc, uoff + 3072u, u_next);                                                     W6_FENCE;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) afc[j][s] = afn[j][s];
        };
        {
            raw_dma(0, raw_off(true));
            advance();
            raw_dma(1, raw_off(1 < nchunk));
            advance();
            raw_dma(2, raw_off(2 < nchunk));
            advance();
        }
        __builtin_amdgcn_sched_barrier(0);
        load_u(0, bu[0]);
        load_u(1 < nchunk ? 1 : 0, bu[1]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // raw(0): everything behind it is raw(1), raw(2), 8 U loads
        raw_barrier();
        {
            const float* r = sRaw;
            float va[2][4], vb[2][4], t[2][4];
            W6_RD(0, 0); W6_RD(0, 1); W6_RD(0, 2); W6_RD(0, 3);
            W6_RD(1, 0); W6_RD(1, 1); W6_RD(1, 2); W6_RD(1, 3);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) t[s][c] = __builtin_fmaf(sgn, vb[s][c], va[s][c]);
            transform(t, afc);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef CF_STAMP
        t_loop_begin = __builtin_readcyclecounter();
#endif
        int k = 0;
        for (; k + 1 < nchunk; k += 2) {
            deep_step(k, bu[0]);
            deep_step(k + 1, bu[1]);
        }
        if (k < nchunk) deep_step(k, bu[0]);
#ifdef CF_STAMP
        t_loop_end = __builtin_readcyclecounter();
#endif
#undef W6_FENCE
#undef W6_PIN
#undef W6_MFMA
#undef W6_RD
    } else {
    f32x4 bu[4];
    auto chunk_step = [&](int k) __attribute__((always_inline)) {
        // in flight, oldest first: the raw(k) piece, then the four U(k) loads
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        raw_barrier();                              // raw(k) has landed for everybody; everybody has read raw(k-1)
#ifdef CF_STAMP
        st_wait += __builtin_readcyclecounter() - t0;
#endif
        const bool more = k + 1 < nchunk;
        issue_raw((k + 1) & 1, more);
        __builtin_amdgcn_sched_barrier(0);          // raw(k+1) before U(k+1) in issue order: the vmcnt(4) above counts on it
        const float* r = sRaw + (k & 1) * W6_RAW;
        float af[4][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = ((c >> 1) + (c & 1) * 5) * 4 + s * (W6_PLANE * 4);
                t[c] = __builtin_fmaf(sgn, r[rd_b + col], r[rd_a + col]);      // row `wave` of B^T d (row 2 negated, as its U is)
            }
            af[0][s] = t[0] - t[2];
            af[1][s] = t[1] + t[2];
            af[2][s] = t[2] - t[1];
            af[3][s] = t[1] - t[3];
        }
        const unsigned u_next = (unsigned)(more ? k + 1 : k) * (W6_UV * 4u);      // past the end: a harmless re-load
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // alternate the two accumulators of a position: back-to-back MFMAs never depend on each other
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][0], bu[j][0], acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][0], bu[j][2], acc[j][1], 0, 0, 0);
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][1], bu[j][1], acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][1], bu[j][3], acc[j][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, u_next);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_raw(0, true);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bu[j] = buf_load4(u_rsrc, uoff + 1024u * j, 0);
    __builtin_amdgcn_sched_barrier(0);
#ifdef CF_STAMP
    t_loop_begin = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < nchunk; ++k) chunk_step(k);
#ifdef CF_STAMP
    t_loop_end = __builtin_readcyclecounter();
#endif

    }

    // the last chunk step's dead past-the-end DMA (zeros into the ring) must have landed before anything overlays the ring: an explicit
    // wait, so that this does not rest on the compiler's LDS-DMA bookkeeping in front of the barrier
    wait_vmcnt0();
    __syncthreads();                                // every wave is done with the raw ring before it becomes the exchange buffer
#ifdef CF_STAMP
    const long long t_tail_a = __builtin_readcyclecounter();      // MFMA drain + the dead past-the-end loads + barrier
#endif
    // ---- output transform, j direction (registers): R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3.  C layout of the 16x16x4 MFMA: lane (t16, kk) holds
    // cout 16 hh + t16 of tiles 4 kk .. 4 kk + 3, i.e. four CONSECUTIVE tiles: X is laid out [i][bcol][cout][tile], so a lane's four values of a
    // (bcol, hh) are one 16-byte write (r04: was [tile][cout] with sixteen ds_write_b32 and thirty-two ds_read_b32 per lane; at one workgroup
    // per CU the exchange was 2.35 k of a launch's 25 k cycles) ----
    float* X = smem;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        f32x4 r0, r1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            r0[r] = (acc[0][hh][r] + acc[1][hh][r]) + acc[2][hh][r];
            r1[r] = (acc[1][hh][r] - acc[2][hh][r]) - acc[3][hh][r];
        }
        const int col = 16 * hh + t16;
        *reinterpret_cast<f32x4*>(X + ((wave * 2 + 0) * 32 + col) * W6_XS + 4 * kk) = r0;
        *reinterpret_cast<f32x4*>(X + ((wave * 2 + 1) * 32 + col) * W6_XS + 4 * kk) = r1;
    }
    __syncthreads();
#ifdef CF_STAMP_FINE
    const long long t_x1 = __builtin_readcyclecounter();
#endif
    // ---- i direction across the waves: thread (col = tid & 31, g = tid >> 5) takes output pixel (a, bb) = ((g >> 1) & 1, g & 1) of the eight
    // tiles 8 (g >> 2) + q: patch row = tile * 4 + a * 2 + bb; two 16-byte reads per i ----
    float yv[8];
    {
        const int col = tid & 31, g = tid >> 5;
        const int a = (g >> 1) & 1, bb = g & 1, th = g >> 2;
        f32x4 x[3][2];                                  // rows a, a + 1, a + 2 of the exchange: the three this pixel's row of A^T uses
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float* src = X + (((a + i) * 2 + bb) * 32 + col) * W6_XS + 8 * th;
            x[i][0] = *reinterpret_cast<const f32x4*>(src);
            x[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float u0 = x[0][q >> 2][q & 3], u1 = x[1][q >> 2][q & 3], u2 = x[2][q >> 2][q & 3];
            yv[q] = a == 0 ? (u0 + u1) + u2 : (u0 - u1) - u2;      // (x0 + x1) + x2 | (x1 - x2) - x3
        }
    }
    __syncthreads();                                // everybody has read X: the patches go on top of it
#ifdef CF_STAMP_FINE
    const long long t_x2 = __builtin_readcyclecounter();
#endif
    {
        const int col = tid & 31, g = tid >> 5;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int prow = (8 * (g >> 2) + q) * 4 + (g & 3);       // tile * 4 + a * 2 + bb
            smem[(prow >> 5) * (32 * EPI_S) + (prow & 31) * EPI_S + col] = yv[q];
        }
        if (tid < 64) {
            const int tl = tid >> 2, a = (tid >> 1) & 1, bb = tid & 1;
            const int oy = oy0 + 2 * (tl >> 2) + a, ox = ox0 + 2 * (tl & 3) + bb;
            sMtab[tid] = (oy < Ho && ox < Wo) ? oy * Wo + ox : -1;
        }
    }
    __syncthreads();
#ifdef CF_STAMP
    const long long t_tail_b = __builtin_readcyclecounter();      // exchange (three barriers)
#endif
    const int pp = wave & 1, q0 = (wave >> 1) * 2;  // wave w finishes rows [16 (w >> 1), + 16) of patch w & 1
    patch_tail(p, smem + pp * (32 * EPI_S), b, 0, n0, lane, Ho * Wo, q0, q0 + 2, 1, 0, sMtab + pp * 32);
    // (r04: the epilogue's bias quad requested in the prologue instead of here took 900 cycles off the tail and put 800 on the prologue -- the
    // first counted wait then also waits for that cold load, which is older than the first raw patch; not kept)
    if (p.st_partial && wave < 2) patch_stats(p, smem + wave * (32 * EPI_S), b, 0, n0, lane, Ho * Wo, sMtab + wave * 32, reg * 2 + wave, nreg * 2);
#ifdef CF_STAMP
    if (p.stamp && lane == 0) {      // [DMA wait + barrier, -, -, prologue, chunks, loop, tail, MHz] cycles of this wave
        long long* q = p.stamp + ((long)blockIdx.x * 4 + wave) * 8;
#ifdef CF_STAMP_FINE
        q[0] = t_x1 - t_tail_a; q[1] = t_x2 - t_x1; q[2] = t_tail_b - t_x2; q[3] = t_loop_begin - t_begin; q[4] = nchunk;      // the exchange in three parts
#else
        q[0] = st_wait; q[1] = t_tail_a - t_loop_end; q[2] = t_tail_b - t_tail_a; q[3] = t_loop_begin - t_begin; q[4] = nchunk;      // [1], [2]: the tail's first two parts
#endif
        q[5] = t_loop_end - t_loop_begin; q[6] = __builtin_readcyclecounter() - t_loop_end;
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
}

// U = G g G^T of a packed direct matrix w [rows][9 taps][cin_pad] in conv_wino16_kernel's order
// [n-block][chunk][pos = i*4+j][lane 64][e = half * 2 + k-step]: n = 32 nb + 16 half + (lane & 15), k = 4 k-step + (lane >> 4); row i = 2 negated
// (see wino_weight_kernel); rows past `rows` are zero
__global__ void wino16_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int rows, int cin_pad, int nblk) {
    const int nchunk = cin_pad / W6_KC;
    const long total = (long)nblk * nchunk * W6_UV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3);
    const int ln = (int)((idx >> 2) & 63);
    const int pos = (int)((idx >> 8) & 15);
    const long blk = idx >> 12;
    const int chunk = (int)(blk % nchunk);
    const int nb = (int)(blk / nchunk);
    const int n = nb * 32 + 16 * (e >> 1) + (ln & 15);
    const int c = chunk * W6_KC + 4 * (e & 1) + (ln >> 4);
    float val = 0.f;
    if (n < rows) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
        const int i = pos >> 2, j = pos & 3;
        double acc = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int bq = 0; bq < 3; ++bq)
                acc += (double)G[i][a] * (double)w[(long)n * 9 * cin_pad + (long)(a * 3 + bq) * cin_pad + c] * (double)G[j][bq];
        val = i == 2 ? -(float)acc : (float)acc;
    }
    u[idx] = val;
}

hipError_t launch_wino16_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s) {
    if (!w || !u || rows <= 0 || cin_pad <= 0 || (cin_pad % W6_KC) != 0) return hipErrorInvalidValue;
    const int nblk = (rows + 31) / 32;
    const long total = (long)nblk * (cin_pad / W6_KC) * W6_UV;
    hipLaunchKernelGGL(wino16_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, rows, cin_pad, nblk);
    return hipGetLastError();
}
long wino16_weight_floats(int rows, int cin_pad) { return (long)((rows + 31) / 32) * (cin_pad / W6_KC) * W6_UV; }

int wino16_regions(int Ho, int Wo) { return ((Ho + 7) / 8) * ((Wo + 7) / 8); }

bool wino16_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || (p.prec != 0 && p.prec != 3) || !p.w_wino16 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.padT != 1 || p.padL != 1) return false;
    if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin < 4 || p.Win < 4) return false;
    if (p.w_bs != 0 && p.w_div <= 1) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg_c[i] % W6_KC) return false;
    return dma_range_ok(p);
}

hipError_t launch_wino16(const ConvParams& p, int batch, hipStream_t s, bool deep) {
    if (!wino16_ok(p)) return hipErrorInvalidValue;
    const long wgs = (long)wino16_regions(p.Ho, p.Wo) * ((p.cout + 31) / 32) * batch;
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 256;
    if (deep) hipLaunchKernelGGL(conv_wino16_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wino16_kernel<0>, dim3((unsigned)wgs), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
