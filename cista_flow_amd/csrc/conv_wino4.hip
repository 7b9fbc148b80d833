// conv_wino4.hip -- Winograd F(4x4,3x3) convolution kernel (round 3) and its weight transform; launched through launch_conv
// (conv_igemm.hip) as tile 42.  gfx950 only.
//
// STATUS: parity-green (tests/test_ops_gpu.py::test_conv_winograd_f4x4) and OPT-IN (CF_WINO4_MIN=<workgroups>; explicit tile 42).
// Measured on MI355X (tools/conv_bench.py, profiles/r03_wino4_*.txt): 1.04-1.05x the F(2x2,3x3) kernel on the 192->256 gate
// convolution (180x240 B=8: 348 vs 364 us; 480x640 B=4: 1090 vs 1142 us), 0.7-0.9x on the 64- and 128-channel layers, whose 4-12
// chunk loops do not amortise its prologue + tail (13 k + 22 k cycles per workgroup) -- so the launcher does not pick it by
// default.  Per chunk a wave spends 15.6 k cycles on 48 MFMAs (3.1 k cycles of the matrix pipe) with three waves per SIMD:
// the matrix pipe is ~0.59 busy inside the loop (tools/stamp_probe.py), against ~1.0 for conv_wino_kernel's loop, i.e. the
// 1.78x fewer MFMAs are spent again on exposed LDS latency of the in-order 4-row column reads and on the barrier skew of
// twelve waves.  What the way here established: a six-wave workgroup at > 128 VGPRs gets ONE slot per CU
// (tools/census_probe.py); 16 bytes per lane from 64 different cache lines bind an LDS-DMA gather to the L1-fill path.
#include "conv_common.h"

namespace cf {

// ---------------------------------------------------------------------------------------------------------
// Winograd F(4x4, 3x3) (round 3): Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A per 6x6 input tile d -> 4x4 output tile,
// i.e. 36 [tiles x Cin] x [Cin x Cout] products per 16 outputs = 2.25 multiplies per output and channel instead of 4
// (F(2x2,3x3)) or 9 (direct): 1.78x fewer matrix-core flops than conv_wino_kernel.  fp32 throughout; the transforms'
// interpolation points 0, +-1, +-2, inf cost ~1e-5 of tensor scale per layer (measured through the whole recurrent
// network on the CPU oracle against the reference goldens BEFORE this kernel was written: worst 2.1e-5 with EVERY
// 3x3 / stride-1 layer replaced; the regression bar of the tests is 2e-4, the contract 1e-3).
//
// One six-wave group = a region of 4 x 8 tiles (16 x 32 output pixels) x 32 output channels (a workgroup holds two such groups, see
// wino4_body); K = Cin in chunks of 16 channels:
//   raw    the (18 x 34 pixel) x 16-channel input patch of a chunk, LDS-DMA'd from the NHWC tensor (reflect / zero padding
//          resolved in the per-lane source offset once per workgroup), double buffered.  FOUR consecutive lanes of a DMA
//          instruction fetch the four channel quads of one patch cell = 64 contiguous bytes: an instruction touches 16 cache
//          lines instead of 64 (the first version fetched 16 bytes per lane from 64 different lines and was bound by the
//          texture-address / L1-fill path at 0.35 of the matrix peak, with 4x more bytes moved from L2 than used).  Cell
//          (py, px) lives at 16-byte slot 4 c + (f ^ quad), c = 34 py + perm(px) (columns stored by residue class px mod 4, so
//          the tiles of a tile row read neighbouring cells), f = 2 bits of the cell's position: the 16 lanes of a ds_read_b64
//          group then hit 16 different bank granules;
//   MFMA   wave i owns positions (i, 0..5) = row i of B^T d B: 6 accumulators [32 tiles x 32 couts] = 96 registers, 48 x
//          v_mfma_f32_32x32x2_f32 per chunk in four steps of 12 (one channel pair of each lane's quad per step);
//   V      never stored.  Row i of B^T d is a 4-term combination of patch rows (coefficients and rows are wave-uniform
//          scalars: 4 d0 - 5 d2 + d4 | d4 - 4 d2 +- (d3 - 4 d1) | d4 - d2 +- 2 (d3 - d1) | 4 d1 - 5 d3 + d5): lane (tile, pair)
//          reads 4 rows x 6 pixels (ds_read_b64 straight from the raw patch), then applies the row transform along the 6
//          columns in registers.  Software pipeline: the A operands of step s + 1 are built in the shadow of the 12 MFMAs of
//          step s (column j right behind position j's MFMAs), inside 168 registers = three waves per SIMD;
//   U      = G g G^T (fp64 at weight-pack time, rounded once), [n-block][chunk][36 pos][4 steps][32 n][2][2]: the 64 lanes of
//          a load read 512 contiguous bytes; every lane loads its own B fragments one step ahead, behind that position's MFMAs;
//   tail   A^T . A is separable: along j in registers (6 -> 4), along i across the six waves through LDS (two output
//          columns j' per round); waves 0..3 each finish output row i' of every tile as four 32-row patches through the
//          common fused epilogue (patch_tail with a row -> pixel table) once the accumulators are dead.
// ---------------------------------------------------------------------------------------------------------
#ifndef W4_ABL
#define W4_ABL 0                                             // ablation builds (tools/build_variant.sh ... -DW4_ABL=n): 8 = no raw DMA
#endif
static constexpr int W4_KC = 16;                             // channels per chunk
static constexpr int W4_CELLS = 18 * 34;                     // patch cells (612)
static constexpr int W4_NDMA = (W4_CELLS * 4 + 63) / 64;     // wave-instructions of LDS-DMA per chunk (39; the last one has 48 dead slots)
static constexpr int W4_RAW = W4_NDMA * 64 * 4;              // floats per raw buffer (39,936 bytes)
static constexpr int W4_UV = 36 * 4 * 32 * 4;                // floats of a chunk's U block (18432)
static constexpr int W4_X = 6 * 2 * 32 * 32;                 // floats of the cross-wave exchange buffer of the tail (48 KB)
static constexpr int W4_TAIL = W4_X + 4 * 32 * EPI_S + 4 * 32;   // + four epilogue patches + their row -> pixel tables (floats)
static constexpr int W4_SMEM = 4 * (2 * W4_RAW > W4_TAIL ? 2 * W4_RAW : W4_TAIL);   // bytes per six-wave group: 79,872 (a workgroup = two groups = 159,744 of 163,840)

__device__ __forceinline__ void wino4_body(const ConvParams& p, float* smem) {
#ifdef CF_STAMP
    const long long t_begin = __builtin_readcyclecounter();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();
    long long st_wait = 0, st_dma = 0, st_s012 = 0;
#endif
    constexpr int TWr = 8;                                   // tile columns of a region (4 tile rows)
    constexpr int RH = 16, RW = 32;                          // region size in output pixels
    constexpr int PC = 34;                                   // patch columns (18 rows)
    constexpr int OFF1 = 9, OFF2 = 18, OFF3 = 26;            // first cell of column class px mod 4 = 1, 2, 3 within a patch row (9 + 9 + 8 + 8 columns)

    // A workgroup is TWELVE waves = two independent six-wave groups (two neighbouring regions, the same 32 output channels), each with
    // its own half of the LDS: a six-wave workgroup lands 2, 2, 1, 1 on the four SIMDs, so at three waves per SIMD a second one never
    // fits beside it (measured with tools/census_probe.py: one resident workgroup per CU); twelve waves land 3, 3, 3, 3.  The groups
    // only share the barriers.
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave12 >= 6 ? 1 : 0;
    const int wave = wave12 - 6 * grp;
    smem += grp * (W4_SMEM / 4);
    float* const sRaw = smem;
    const int Ho = p.Ho, Wo = p.Wo;
    const int nrx = (Wo + RW - 1) / RW, nry = (Ho + RH - 1) / RH;
    const int nreg = nrx * nry;
    const int npair = (nreg + 1) / 2;
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
    const int reg_raw = 2 * (rest % npair) + grp;
    const bool dead = reg_raw >= nreg;              // odd region count: the last pair's second group computes on zeros and stores nothing
    const int reg = dead ? nreg - 1 : reg_raw;
    const int b = rest / npair;
    const int oy0 = (reg / nrx) * RH, ox0 = (reg % nrx) * RW;
    const int n0 = nblk * 32;

    // ---- raw patch DMA: wave-instruction id = wave + 6 j covers slots [64 id, 64 id + 64) of a buffer; slot -> (cell, quad).
    // Per slot a 16-bit descriptor (two per register: the loop has no registers to spare): source pixel relative to the patch origin
    // after padding resolution -- row 0..63 | column 0..127 << 6 | quad << 13 | dead << 15 ----
    unsigned a_src[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int sl = 64 * (wave + 6 * j) + lane;
        const int c = sl >> 2;
        const int py = c / PC, u = c - py * PC;
        const int r = u >= OFF3 ? 3 : u >= OFF2 ? 2 : u >= OFF1 ? 1 : 0;
        const int px = 4 * (u - (r == 3 ? OFF3 : r == 2 ? OFF2 : r == 1 ? OFF1 : 0)) + r;
        const int f = ((u >> 2) & 1) | (((py >> 2) & 1) << 1);
        const int quad = (sl & 3) ^ f;
        int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        bool ok = !dead && c < W4_CELLS && iy <= p.Hin && ix <= p.Win;       // beyond the halo of the last row / column: unused
        if (p.pad_mode == 1) {
            iy = reflect_idx(iy, p.Hin);
            ix = reflect_idx(ix, p.Win);
        } else {
            ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        }
        // reflection moves a coordinate by at most 2 around the patch: (iy - oy0 + 8) in 0..63, (ix - ox0 + 8) in 0..127
        const unsigned d = ok ? (unsigned)(iy - oy0 + 8) | ((unsigned)(ix - ox0 + 8) << 6) | ((unsigned)quad << 13) : 0x8000u;
        a_src[j >> 1] |= d << (16 * (j & 1));
    }
    const int pix0 = (oy0 - 8) * p.Win + (ox0 - 8);               // pixel index of descriptor (0, 0)
    const int nchunk = p.cin_pad / W4_KC;
    const __amdgpu_buffer_rsrc_t u_rsrc = make_rsrc(p.w_wino4 + (long)wgroup(p, b) * p.wino4_gs + (long)nblk * nchunk * W4_UV);

    int it_seg = 0, it_cs = 0;
    const float* seg_base = p.in[0] + (long)b * p.seg_bs[0];
    int seg_ld = p.seg_ld[0], seg_cn = p.seg_c[0];
    // dead slots and the chunks past the end fetch out of range (= zeros into LDS): the issue pattern never changes.  The pieces of a
    // chunk are issued one at a time (issue_piece) so that the main loop can tuck them between its MFMAs.
    auto issue_piece = [&](int j, int buf, bool live) __attribute__((always_inline)) {
        // pieces 0..5 exist for every wave and are issued UNCONDITIONALLY: behind a branch the compiler's vmcnt bookkeeping assumes the
        // piece may be missing and waits for the slow gather in front of the next MFMAs; the seventh piece exists for waves 0..2 only
        static_assert(5 + 6 * 5 < W4_NDMA && 6 * 6 + 3 == W4_NDMA, "39 pieces over six waves");
        if (j < 6 || wave < 3) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(seg_base);
            unsigned w = a_src[j >> 1];
            asm volatile("" : "+v"(w));             // decode HERE, per chunk: hoisted out of the loop the seven decoded offsets would spill
            const unsigned d = (w >> (16 * (j & 1))) & 0xFFFFu;
            // 24-bit multiplies (full rate; the 32-bit forms are quarter rate and this runs in the MFMA shadow): row < 64, Win < 2^24,
            // pixel index < 2^24 (wino4_ok), pixel stride in bytes < 2^24
            const unsigned pix = (unsigned)pix0 + __umul24(d & 63u, (unsigned)p.Win) + ((d >> 6) & 127u);
            const unsigned off = ((d & 0x8000u) || !live) ? BUF_OOB : __umul24(pix, (unsigned)seg_ld * 4u) + ((d >> 13) & 3u) * 16u;
#if !(W4_ABL & 8)
            dma16_to_lds(rs, sRaw + buf * W4_RAW + 64 * (wave + 6 * j) * 4, off, (unsigned)it_cs * 4u);
#else
            if (off == 12345u) smem[0] = 1.f;
#endif
        }
    };
    auto next_chunk = [&]() __attribute__((always_inline)) {
        it_cs += W4_KC;
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
    auto issue_raw = [&](int buf, bool live) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 7; ++j) issue_piece(j, buf, live);
        next_chunk();
    };

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // lane (lr, lh): tile lr = (tty, ttx) of the region, channel pair lh of the step's quad.  Row `wave` of B^T d =
    // k0 d[ra] + k1 d[rb] + k2 d[rc] + k3 d[rd] (waves 0 and 5 have three terms: their last row is taken twice at half weight --
    // exact -- instead of a zero coefficient)
    const int lr = lane & 31, lh = lane >> 5;
    const int tty = lr / TWr, ttx = lr - tty * TWr;
    const bool edge = wave == 0 || wave == 5;
    const int ra = wave == 0 ? 0 : 1, rb = wave == 5 ? 3 : 2, rc = wave == 0 ? 4 : wave == 5 ? 5 : 3, rd = wave == 5 ? 5 : 4;
    const float k0 = edge ? 4.f : wave == 1 ? -4.f : wave == 2 ? 4.f : wave == 3 ? -2.f : 2.f;
    const float k1 = edge ? -5.f : (wave == 1 || wave == 2) ? -4.f : -1.f;
    const float k2 = edge ? .5f : wave == 1 ? 1.f : wave == 2 ? -1.f : wave == 3 ? 2.f : -2.f;
    const float k3 = edge ? .5f : 1.f;
    auto sgpr = [](float x) __attribute__((always_inline)) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
    const float k0s = sgpr(k0), k1s = sgpr(k1), k2s = sgpr(k2), k3s = sgpr(k3);      // scalar operands of the column FMAs
    // byte address of (row a, column bq, quad t) in a raw buffer: 64 c + 16 (f ^ t) + 8 lh with c = 34 (4 tty + a) + perm(bq) + ttx,
    // f = bit 2 of (perm(bq) + ttx) | ((tty + (a >= 4)) & 1) << 1.  perm(bq) mod 8 is 0, 1 or 2, so three lane bases cover the six
    // columns; rows >= 4 and the quad enter as an XOR on bits 4..5, the row / column displacement as a scalar + immediate offset
    int abase[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
        abase[m] = 64 * (34 * 4 * tty + ttx) + 16 * ((((m + ttx) >> 2) & 1) | ((tty & 1) << 1)) + 8 * lh;
    const int ro_a = 64 * 34 * ra, ro_b = 64 * 34 * rb, ro_c = 64 * 34 * rc, ro_d = 64 * 34 * rd;    // wave-uniform row displacements (bytes)
    const int xr_c = rc >= 4 ? 32 : 0;                                                               // rows 4, 5 flip the row bit (rd always does)
    // U: [pos][step][n][lh][2]: lane part n * 16 + lh * 8 bytes; the rest is wave-uniform
    const unsigned uoff = (unsigned)(lr * 16 + lh * 8);
    const unsigned ubase = (unsigned)(wave * 6) * 2048u;

    f32x2 bu[6];
    struct Col { f32x2 a, b, c, d; };
    // column bq of the patch for quad t (compile-time) out of the raw buffer at byte offset `bo` (wave-uniform)
    auto read_col = [&](int bo, int bq, int t) __attribute__((always_inline)) {
        constexpr int PERM[6] = {0, OFF1, OFF2, OFF3, 1, OFF1 + 1};
        const int m = PERM[bq] & 7;                                   // 0, 1, 2, 2, 1, 2
        int base = abase[m];
        asm volatile("" : "+v"(base));              // keep the XOR / add per read: hoisted, the 36 loop-invariant addresses would spill
        const char* r = reinterpret_cast<const char*>(sRaw);
        const int disp = bo + 64 * PERM[bq];
        Col x;
        x.a = *reinterpret_cast<const f32x2*>(r + ((base ^ (16 * t)) + (ro_a + disp)));
        x.b = *reinterpret_cast<const f32x2*>(r + ((base ^ (16 * t)) + (ro_b + disp)));
        x.c = *reinterpret_cast<const f32x2*>(r + ((base ^ ((16 * t) ^ xr_c)) + (ro_c + disp)));
        x.d = *reinterpret_cast<const f32x2*>(r + ((base ^ ((16 * t) ^ 32)) + (ro_d + disp)));
        return x;
    };
    // scalar FMAs on purpose: the coefficients are wave-uniform and ride in SGPR operands (the packed forms want them in VGPR pairs,
    // six registers this kernel does not have)
    auto col_xf = [&](const Col& x) __attribute__((always_inline)) {
        f32x2 c;
        c[0] = __builtin_fmaf(k0s, x.a[0], __builtin_fmaf(k1s, x.b[0], __builtin_fmaf(k2s, x.c[0], k3s * x.d[0])));
        c[1] = __builtin_fmaf(k0s, x.a[1], __builtin_fmaf(k1s, x.b[1], __builtin_fmaf(k2s, x.c[1], k3s * x.d[1])));
        return c;
    };
    auto row_xf = [&](const f32x2 (&c)[6], f32x2 (&v)[6]) __attribute__((always_inline)) {
        const f32x2 m4 = {-4.f, -4.f}, m5 = {-5.f, -5.f}, p4 = {4.f, 4.f}, p2 = {2.f, 2.f}, m2 = {-2.f, -2.f};
        const f32x2 t1 = __builtin_elementwise_fma(m4, c[2], c[4]), t2 = __builtin_elementwise_fma(m4, c[1], c[3]);
        const f32x2 t3 = c[4] - c[2], t4 = c[3] - c[1];
        v[0] = __builtin_elementwise_fma(p4, c[0], __builtin_elementwise_fma(m5, c[2], c[4]));
        v[5] = __builtin_elementwise_fma(p4, c[1], __builtin_elementwise_fma(m5, c[3], c[5]));
        v[1] = t1 + t2;
        v[2] = t1 - t2;
        v[3] = __builtin_elementwise_fma(p2, t4, t3);
        v[4] = __builtin_elementwise_fma(m2, t4, t3);
    };
    // one step: the 12 MFMAs on (v, bu) with the transform of the NEXT step (raw buffer at byte offset bo, quad tn) in their shadow;
    // u_off = scalar byte offset of the U block that refills bu (the next step's)
    // dma_buf >= 0: the raw pieces of the chunk after next go out between the MFMA groups as well (after the hand-off barrier every
    // wave used to issue its seven pieces back to back with the matrix pipe empty: 2.2 k cycles per chunk)
    auto step = [&](f32x2 (&v)[6], int bo, int tn, unsigned u_off, int dma_buf, bool dma_live) __attribute__((always_inline)) {
        f32x2 cn[6];
        Col xa = read_col(bo, 0, tn), xb;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            // the empty asm statements pin WHERE the packed-math results exist: without them instruction selection places the whole
            // transform behind the last MFMA (pure arithmetic carries no chain), and the wave is back to two serial phases
            asm volatile("" : "+v"(v[j]));
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][0], bu[j][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][1], bu[j][1], acc[j], 0, 0, 0);
            bu[j] = buf_load2(u_rsrc, uoff, u_off + 2048u * j);
            if (dma_buf >= 0) {
                issue_piece(j, dma_buf, dma_live);
                if (j == 5) issue_piece(6, dma_buf, dma_live);
            }
            if (j < 5) xb = read_col(bo, j + 1, tn);
            cn[j] = col_xf(xa);
            asm volatile("" : "+v"(cn[j]));
            xa = xb;
            __builtin_amdgcn_sched_barrier(0);
        }
        row_xf(cn, v);
        if (dma_buf >= 0) next_chunk();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: raw(0), raw(1) in flight; U of step 0; A operands of step 0 ----
    issue_raw(0, true);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) bu[j] = buf_load2(u_rsrc, uoff, ubase + 2048u * j);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // the raw(0) pieces are older than the six U loads
    raw_barrier();
    issue_raw(1, nchunk > 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 v[6];
    {
        f32x2 c0[6];
#pragma unroll
        for (int bq = 0; bq < 6; ++bq) c0[bq] = col_xf(read_col(0, bq, 0));
        row_xf(c0, v);
    }
#ifdef CF_STAMP
    const long long t_loop_begin = __builtin_readcyclecounter();
#endif
    for (int k = 0; k < nchunk; ++k) {
        const int bo = (k & 1) * (W4_RAW * 4), bn = ((k + 1) & 1) * (W4_RAW * 4);
        const unsigned ub = ubase + (unsigned)k * (W4_UV * 4u);
#ifdef CF_STAMP
        const long long t0 = __builtin_readcyclecounter();
#endif
        step(v, bo, 1, ub + 512u, -1, false);       // (k, 0): builds (k, 1); bu <- U(k, 1)
        step(v, bo, 2, ub + 1024u, -1, false);      // (k, 1)
        step(v, bo, 3, ub + 1536u, -1, false);      // (k, 2)
#ifdef CF_STAMP
        const long long t1 = __builtin_readcyclecounter();
#endif
        // hand-off: raw(k+1) has landed (its pieces are older than the six U loads just issued), everybody is done with raw(k)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        raw_barrier();
#ifdef CF_STAMP
        const long long t2 = __builtin_readcyclecounter();
#endif
#ifdef CF_STAMP
        const long long t3 = __builtin_readcyclecounter();
        st_s012 += t1 - t0; st_wait += t2 - t1; st_dma += t3 - t2;
#endif
        // (k, 3): builds (k+1, 0) from the next raw buffer (zeros past the end); bu <- U(k+1, 0) (past the end: a re-load)
        // raw(k+2) goes into the buffer raw(k) just left, piece by piece between this step's MFMAs
        step(v, bn, 0, k + 1 < nchunk ? ub + W4_UV * 4u : ub, k & 1, k + 2 < nchunk);
    }

#ifdef CF_STAMP
    const long long t_loop_end = __builtin_readcyclecounter();
    if (p.stamp && lane == 0) {      // [vmcnt wait + barrier, DMA issue, steps 0..2, prologue, chunks, loop, -, MHz] cycles of this wave
        long long* q = p.stamp + ((long)blockIdx.x * 12 + wave12) * 8;
        q[0] = st_wait; q[1] = st_dma; q[2] = st_s012; q[3] = t_loop_begin - t_begin; q[4] = nchunk;
        q[5] = t_loop_end - t_loop_begin;
        const long long dr = (long long)__builtin_amdgcn_s_memrealtime() - r_begin;
        q[7] = dr > 0 ? ((__builtin_readcyclecounter() - t_begin) * 100) / dr : 0;
    }
#endif
    // the last chunk step's dead past-the-end DMA (zeros into the ring) must have landed before anything overlays the ring: an explicit
    // wait, so that this does not rest on the compiler's LDS-DMA bookkeeping in front of the barrier
    wait_vmcnt0();
    __syncthreads();                                // every wave is done with the raw ring before it becomes the exchange buffer
    // ---- output transform.  Along j in registers, IN PLACE (acc[j'] <- sum_j A^T[j'][j] M[i][j], j' = 0..3); along i across the
    // waves through LDS, two output columns j' per round: X[i = wave][2][tile 32][cout 32] = 48 KB.  Waves 0..3 (= output row i')
    // collect their 4 x 16 outputs per lane in registers; only after the second round -- when the accumulators are dead -- do the
    // register-hungry fused epilogues run, one 32-tile patch per (i', j') ----
    float* const X = smem;
    float* const sW = smem + W4_X + (wave & 3) * (32 * EPI_S);
    int* const mtab = reinterpret_cast<int*>(smem + W4_X + 4 * 32 * EPI_S) + (wave & 3) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        acc[0][r] = (m0 + s12) + s34;
        acc[1][r] = __builtin_fmaf(2.f, d34, d12);
        acc[2][r] = __builtin_fmaf(4.f, s34, s12);
        acc[3][r] = __builtin_fmaf(8.f, d34, d12) + m5;
    }
    // A^T row i' = wave for the second stage (waves 0..3): (1 1 1 1 1 0) (0 1 -1 2 -2 0) (0 1 1 4 4 0) (0 1 -1 8 -8 1)
    const float e0 = wave == 0 ? 1.f : 0.f, e2 = (wave & 1) ? -1.f : 1.f;
    const float e3 = wave == 0 ? 1.f : wave == 1 ? 2.f : wave == 2 ? 4.f : 8.f, e4 = (wave & 1) ? -e3 : e3, e5 = wave == 3 ? 1.f : 0.f;
    float o[4][16];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        if (rnd) __syncthreads();                   // everybody has read round 0
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                X[((wave * 2 + jj) * 32 + trow) * 32 + lr] = acc[2 * rnd + jj][r];
            }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int t = lh * 16 + q;
                    const float x0 = X[((0 * 2 + jj) * 32 + t) * 32 + lr], x1 = X[((1 * 2 + jj) * 32 + t) * 32 + lr];
                    const float x2 = X[((2 * 2 + jj) * 32 + t) * 32 + lr], x3 = X[((3 * 2 + jj) * 32 + t) * 32 + lr];
                    const float x4 = X[((4 * 2 + jj) * 32 + t) * 32 + lr], x5 = X[((5 * 2 + jj) * 32 + t) * 32 + lr];
                    float y = e5 * x5;
                    y = __builtin_fmaf(e4, x4, y);
                    y = __builtin_fmaf(e3, x3, y);
                    y = __builtin_fmaf(e2, x2, y);
                    y = y + x1;
                    y = __builtin_fmaf(e0, x0, y);
                    o[2 * rnd + jj][q] = y;
                }
        }
    }
#ifdef CF_STAMP
    if (p.stamp && lane == 0) p.stamp[((long)blockIdx.x * 12 + wave12) * 8 + 6] = __builtin_readcyclecounter() - t_loop_end;   // exchange part of the tail
#endif
    if (wave >= 4) return;                          // no barrier below this line
#pragma unroll 1
    for (int jp = 0; jp < 4; ++jp) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float y = jp == 0 ? o[0][q] : jp == 1 ? o[1][q] : jp == 2 ? o[2][q] : o[3][q];
            sW[(lh * 16 + q) * EPI_S + lr] = y;
        }
        if (lane < 32) {
            const int ty = lane / TWr, tx = lane - ty * TWr;
            const int oy = oy0 + 4 * ty + wave, ox = ox0 + 4 * tx + jp;
            mtab[lane] = (!dead && oy < Ho && ox < Wo) ? oy * Wo + ox : -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        patch_tail(p, sW, b, 0, n0, lane, Ho * Wo, 0, 4, 1, 0, mtab);
        if (p.st_partial && !dead) patch_stats(p, sW, b, 0, n0, lane, Ho * Wo, mtab, reg * 16 + wave * 4 + jp, nreg * 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();            // the patch and its table are rewritten by the next column
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

__global__ __launch_bounds__(768, 3) void conv_wino4_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * W4_SMEM / 4]; // ONE __shared__ object (see conv_wino_kernel): 159,744 of the CU's 163,840 bytes
#ifdef CF_CENSUS
    long long c0;
    census_begin(p, c0);
#endif
    wino4_body(p, smem);
#ifdef CF_CENSUS
    census_end(p, c0);
#endif
}

// U = G g G^T for F(4x4,3x3) of a packed direct matrix w [rows][tap][cin_pad], stored
// [n-block][chunk of 16][pos = i*6+j][step t][32 n][lh][2] (channel = 16 chunk + 4 t + 2 lh + e: what lane (n, lh) of the owning wave
// feeds to the two MFMAs of step t); computed in fp64 and rounded once; rows past `rows` are zero
__global__ void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int rows, int cin_pad, int nblk) {
    const int nchunk = cin_pad / W4_KC;
    const long total = (long)nblk * nchunk * W4_UV;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 1), lh = (int)((idx >> 1) & 1);
    const int nl = (int)((idx >> 2) & 31);
    const int t = (int)((idx >> 7) & 3);
    const long rest = idx >> 9;
    const int pos = (int)(rest % 36);
    const long blk = rest / 36;
    const int chunk = (int)(blk % nchunk);
    const int nb = (int)(blk / nchunk);
    const int n = nb * 32 + nl;
    const int c = chunk * W4_KC + 4 * t + 2 * lh + e;
    float val = 0.f;
    if (n < rows) {
        const double G[6][3] = {{1.0 / 4, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
        const int i = pos / 6, j = pos - i * 6;
        double acc = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int bq = 0; bq < 3; ++bq)
                acc += G[i][a] * (double)w[(long)n * 9 * cin_pad + (long)(a * 3 + bq) * cin_pad + c] * G[j][bq];
        val = (float)acc;
    }
    u[idx] = val;
}

hipError_t launch_wino4_weights(const float* w, float* u, int rows, int cin_pad, hipStream_t s) {
    if (!w || !u || rows <= 0 || cin_pad <= 0 || (cin_pad % W4_KC) != 0) return hipErrorInvalidValue;
    const int nblk = (rows + 31) / 32;
    const long total = (long)nblk * (cin_pad / W4_KC) * W4_UV;
    hipLaunchKernelGGL(wino4_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, rows, cin_pad, nblk);
    return hipGetLastError();
}
long wino4_weight_floats(int rows, int cin_pad) { return (long)((rows + 31) / 32) * (cin_pad / W4_KC) * W4_UV; }
int wino4_regions(int Ho, int Wo) { return ((Ho + 15) / 16) * ((Wo + 31) / 32); }

// F(4x4,3x3): same layer class as wino_ok, with its own transformed weights
bool wino4_ok(const ConvParams& p) {
    if (p.a_mode != A_NHWC || p.prec != 0 || !p.w_wino4 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.padT != 1 || p.padL != 1) return false;
    if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin < 12 || p.Win < 12) return false;
    if ((long)p.Hin * p.Win >= (1L << 24)) return false;           // 24-bit pixel arithmetic in the raw-patch DMA
    if (p.w_bs != 0 && p.w_div <= 1) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg_c[i] % W4_KC || p.seg_ld[i] * 4L >= (1L << 24)) return false;
    return dma_range_ok(p);
}

hipError_t launch_wino4(const ConvParams& p, int batch, hipStream_t s) {
    if (!wino4_ok(p)) return hipErrorInvalidValue;
    const long wgs = (long)((wino4_regions(p.Ho, p.Wo) + 1) / 2) * ((p.cout + 31) / 32) * batch;       // two regions per workgroup
    if (wgs <= 0 || wgs >= 0x7FFFFFFFL) return hipErrorInvalidValue;
    g_last_launch.threads = wgs * 768;
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)wgs), dim3(768), 0, s, p);
    return hipGetLastError();
}

}  // namespace cf
