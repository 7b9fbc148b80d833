// conv_common.h -- device helpers shared by the convolution kernels (conv_igemm.hip, conv_wino4.hip): vector types, raw buffer
// loads / stores, the fused epilogues (epilogue4, patch_tail, patch_stats) and the LDS-DMA primitives.  Everything here is
// __device__ __forceinline__ or static: each translation unit gets its own copy.
#pragma once
#include "cf_kernels.h"

#include <cstdlib>

namespace cf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

static constexpr int KC = 16;      // K chunk (floats)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// nn.Conv2d(padding_mode='reflect'): mirror without repeating the edge (pad < size).
__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return i;
}

// f16 split of an fp32 value: hi = f16(x) (saturated to the f16 range), lo = f16(x - hi).  hi + lo carries 22
// mantissa bits; products of f16 values are exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16.
__device__ __forceinline__ void split_f16(const f32x4& v, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = fminf(fmaxf(v[e], -65504.f), 65504.f);
        const _Float16 h = (_Float16)x;
        hi[e] = h;
        lo[e] = (_Float16)(x - (float)h);
    }
}

// Global reads go through raw buffer loads: a wave-uniform descriptor (SGPRs) + a 32-bit per-lane byte offset
// instead of 64-bit flat addresses (far fewer VALU ops per load), and the hardware range check returns 0 for
// BUF_OOB offsets -- which is how zero padding and out-of-tile rows are produced, with no select afterwards.
static constexpr unsigned BUF_RECORDS = 0x7FFFFF00u;
static constexpr unsigned BUF_OOB = 0x7FFFFF80u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, BUF_RECORDS, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));   // u32x2
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

__device__ __forceinline__ const float* sel3(const float* const (&a)[3], int s) {
    return s == 0 ? a[0] : (s == 1 ? a[1] : a[2]);
}


// The fused-epilogue helpers below are templates over the parameter block's type CP: ConvParams itself (the by-value kernel argument,
// whose fields the compiler loads once, wherever it likes) or `const __attribute__((address_space(4))) ConvParams` read through a
// laundered kernarg-segment pointer (KernargParams below): conv_wino_p_kernel runs its tail INSIDE a loop, and with the by-value
// argument every field the tail touches would be hoisted out of that loop and held in SGPRs across the MFMA loop (183 SGPR spills).
typedef const __attribute__((address_space(4))) ConvParams KernargParams;

// weight / bias group of image b (ConvParams::w_div)
template <class CP>
__device__ __forceinline__ int wgroup(const CP& p, int b) { return p.w_div > 1 ? b / p.w_div : b; }

static constexpr int EPI_S = 36;   // per-wave epilogue patch row stride (floats)

// Element-wise tail of one conv output quad: pixel m, couts n..n+3 (n % 4 == 0).
template <class CP>
__device__ __forceinline__ void epilogue4(const CP& p, int b, int m, int n, f32x4 acc) {
    const int nv = (p.cout - n) < 4 ? (p.cout - n) : 4;      // valid couts in this quad
    const bool full = nv == 4;
    f32x4 v = acc;
    if (p.bias) {
        const float* bias = p.bias + (long)wgroup(p, b) * p.bias_gs;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (e < nv) ? bias[n + e] : 0.f;
    }
    if (p.addend) {   // iteration-invariant part of a linear layer, precomputed once per frame
        const long aoff = (long)b * p.addend_bs + (long)m * p.addend_ld + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (e < nv) ? p.addend[aoff + e] : 0.f;
    }
    const long ooff = (long)b * p.out_bs + (long)m * p.out_ld + (long)n * p.out_cs;
    const bool ovec = full && p.out_cs == 1 && ((ooff & 3) == 0);
    f32x4 o = v;
    bool to_out = true;   // false: result goes to out2 (split epilogues)
    switch (p.epi) {
        case EPI_NONE: break;
        case EPI_RELU:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.f);
            break;
        case EPI_SIGMOID:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = sigmoidf_(v[e]);
            break;
        case EPI_TANH:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = tanhf(v[e]);
            break;
        case EPI_SUB_FROM_AUX: {
            const long off = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = p.aux0[off + (e < nv ? e : 0)] - v[e];
        } break;
        case EPI_ADD_AUX_SHRINK: {
            const long off = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ee = e < nv ? e : 0;
                const float x = v[e] + p.aux0[off + ee];
                const float l = p.lam[n + ee];
                o[e] = fmaxf(x - l, 0.f) - fmaxf(-x - l, 0.f);
            }
        } break;
        case EPI_RELU_ADD_AUX: {
            const long off = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.f) + p.aux0[off + (e < nv ? e : 0)];
        } break;
        case EPI_RELU_ADD_AUX_RELU: {
            const long off = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(p.aux0[off + (e < nv ? e : 0)] + fmaxf(v[e], 0.f), 0.f);
        } break;
        case EPI_LSTC: {
            // aux0 = sigmoid gates [px][2*split] (i | f), aux1 = z0, aux2 = c_prev; out = z, out2 = c
            const long o0 = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
            const long o1 = (long)b * p.aux1_bs + (long)m * p.aux1_ld + n;
            const long o2 = (long)b * p.aux2_bs + (long)m * p.aux2_ld + n;
            const long oc = (long)b * p.out2_bs + (long)m * p.out2_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e < nv) {
                    const float ig = p.aux0[o0 + e];
                    const float fg = p.aux0[o0 + p.split + e];
                    const float z0 = p.aux1[o1 + e];
                    const float cp = p.aux2[o2 + e];
                    const float og = sigmoidf_(v[e]);
                    const float c = fg * cp + ig * z0;
                    o[e] = og * tanhf(c);
                    p.out2[oc + e] = c;
                }
            }
        } break;
        case EPI_GRU_ZR: {
            // split % 4 == 0, so a quad never straddles the z | r boundary
            if (n < p.split) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = sigmoidf_(v[e]);
            } else {
                to_out = false;
                const int nn = n - p.split;
                const long oh = (long)b * p.aux0_bs + (long)m * p.aux0_ld + nn;
                const long o2 = (long)b * p.out2_bs + (long)m * p.out2_ld + nn;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nv) p.out2[o2 + e] = sigmoidf_(v[e]) * p.aux0[oh + e];
            }
        } break;
        case EPI_GRU_Q: {
            const long oz = (long)b * p.aux0_bs + (long)m * p.aux0_ld + n;
            const long oh = (long)b * p.aux1_bs + (long)m * p.aux1_ld + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ee = e < nv ? e : 0;
                const float z = p.aux0[oz + ee];
                const float h = p.aux1[oh + ee];
                o[e] = (1.f - z) * h + z * tanhf(v[e]);
            }
        } break;
        case EPI_TANH_RELU_SPLIT: {
            if (n < p.split) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = tanhf(v[e]);
            } else {
                to_out = false;
                const long o2 = (long)b * p.out2_bs + (long)m * p.out2_ld + (n - p.split);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nv) p.out2[o2 + e] = fmaxf(v[e], 0.f);
            }
        } break;
        case EPI_SCALE:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[e] * p.scale;
            break;
        case EPI_BIAS_SCALE:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] * p.scale;
            break;
        case EPI_LSTM_ACT:
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (n < p.split) ? sigmoidf_(v[e]) : tanhf(v[e]);
            break;
        case EPI_ADD_AUX: {
            const long off = (long)b * p.aux0_bs + (long)m * p.aux0_ld + (long)n * p.aux0_cs;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] + p.aux0[off + (long)(e < nv ? e : 0) * p.aux0_cs];
        } break;
        default: break;
    }
    if (to_out) {
        if (ovec) {
            *reinterpret_cast<f32x4*>(p.out + ooff) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nv) p.out[ooff + (long)e * p.out_cs] = o[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast tail of a 32x32 patch whose 32 columns are all real couts and whose aux / out rows are 16-byte aligned
// (p.epi_vec, checked on the host).  The wave's sub-tile sits in its LDS patch (row = pixel, stride EPI_S); a
// lane owns channel quad (lane & 7) of pixel rows (lane >> 3) + 8 it.  All global accesses are raw buffer
// dwordx4 operations on a wave-uniform descriptor (tensor base of image b) with a 32-bit per-lane offset, so rows
// past M are masked by the out-of-range offset (loads return 0, stores are dropped) instead of by branches, and
// the loads of EPI_BATCH quads are in flight together -- the earlier tail (one dependent scalar load -> store
// chain per quad) cost ~26k cycles per 128x64 tile.
// ---------------------------------------------------------------------------------------------------------
#ifndef EPI_BATCH
#define EPI_BATCH 1
#endif
#ifndef WPE2
#define WPE2 4
#endif
struct EpiAux {
    f32x4 a0, a1, a2, a3;     // aux0 | aux1 | aux2 or addend | second half of aux0 (LSTC forget gate)
};

// CF_STORE_AUX: cache policy of the fused epilogue's output stores (gfx950 buffer instructions: 1 = sc0, 2 = nt, 16 = sc1 = write-through).
#ifndef CF_STORE_AUX
#define CF_STORE_AUX 0
#endif
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, CF_STORE_AUX);
}

// CLS = which aux slots the epilogue kind can touch (so that only those are allocated): 0 none, 1 aux0,
// 2 aux0 + aux1 + addend (GRU, and any kind with an addend), 3 all four (LSTC).  EB = quads of a lane whose loads are
// in flight together: the tail is a chain of dependent global round trips (~0.8 us each under load), so EB = 4
// leaves one exposed latency per sub-tile instead of four.
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, CF_STORE_AUX);
}

// it0 / it1: which of the lane's four quads (pixel rows (lane >> 3) + 8 it) to finish; nparts / pstride: the patch is the
// sum of nparts partial patches pstride floats apart (split-K tiles: every wave finishes its share of the rows).
// mtab (nullable, LDS): row r of the patch is output pixel mtab[r] (< 0: no pixel) instead of mrow0 + r -- tiles whose rows
// are not consecutive pixels (the Winograd kernel's 2x2 output blocks)
template <int EB, int CLS, class CP>
__device__ __forceinline__ void patch_tail_fast(const CP& p, const float* sW, int b, int mrow0, int nbase, int lane,
                                                int M, int it0, int it1, int nparts, int pstride, const int* mtab = nullptr) {
    const int n = nbase + (lane & 7) * 4;
    const int mb = mrow0 + (lane >> 3);
    const int epi = p.epi;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, lam4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = buf_load4(make_rsrc(p.bias + (long)wgroup(p, b) * p.bias_gs), 4u * (unsigned)n, 0);
    if (epi == EPI_ADD_AUX_SHRINK) lam4 = buf_load4(make_rsrc(p.lam), 4u * (unsigned)n, 0);
    const bool has_add = p.addend != nullptr;
    // descriptors of absent tensors are built from null + offset and never used (a pointer select here makes
    // hipcc copy the whole by-value ConvParams to scratch)
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.out + (long)b * p.out_bs);
    const __amdgpu_buffer_rsrc_t r_out2 = make_rsrc(p.out2 + (long)b * p.out2_bs);
    const __amdgpu_buffer_rsrc_t r_a0 = make_rsrc(p.aux0 + (long)b * p.aux0_bs);
    const __amdgpu_buffer_rsrc_t r_a1 = make_rsrc(p.aux1 + (long)b * p.aux1_bs);
    const __amdgpu_buffer_rsrc_t r_a2 = make_rsrc(has_add ? p.addend + (long)b * p.addend_bs : p.aux2 + (long)b * p.aux2_bs);
    const unsigned ld_a2 = 4u * (unsigned)(has_add ? p.addend_ld : p.aux2_ld);
    const unsigned n4 = 4u * (unsigned)n;
    const bool zr_hi = epi == EPI_GRU_ZR && n >= p.split;          // r half of the z|r conv (per quad)
    const bool sp_hi = epi == EPI_TANH_RELU_SPLIT && n >= p.split;
    const unsigned n4_hi = 4u * (unsigned)(n - p.split);
#pragma unroll 1
    for (int h = it0; h < it1; h += EB) {
        EpiAux x[EB];
#pragma unroll
        for (int it = 0; it < EB; ++it) {
            const int m = mtab ? mtab[((lane >> 3) + (h + it) * 8) & 31] : mb + (h + it) * 8;
            const bool ok = (mtab ? m >= 0 : m < M) && h + it < it1;
            const unsigned um = (unsigned)m;
            const unsigned o0 = ok ? um * (4u * (unsigned)p.aux0_ld) + n4 : BUF_OOB;
            if constexpr (CLS == 1) {
                if (epi == EPI_LSTM_CELL)      // one hidden channel per quad: previous cell state is a scalar
                    x[it].a0[0] = buf_load1(r_a0, ok ? um * (4u * (unsigned)p.aux0_ld) + (unsigned)n : BUF_OOB, 0);
                else
                    x[it].a0 = buf_load4(r_a0, o0, 0);
            } else if constexpr (CLS == 2) {
                if (has_add) x[it].a2 = buf_load4(r_a2, ok ? um * ld_a2 + n4 : BUF_OOB, 0);
                if (epi == EPI_GRU_ZR) {
                    x[it].a0 = buf_load4(r_a0, (ok && zr_hi) ? um * (4u * (unsigned)p.aux0_ld) + n4_hi : BUF_OOB, 0);
                } else if (epi == EPI_GRU_Q) {
                    x[it].a0 = buf_load4(r_a0, o0, 0);
                    x[it].a1 = buf_load4(r_a1, ok ? um * (4u * (unsigned)p.aux1_ld) + n4 : BUF_OOB, 0);
                } else if (epi == EPI_SUB_FROM_AUX || epi == EPI_ADD_AUX_SHRINK || epi == EPI_RELU_ADD_AUX ||
                           epi == EPI_RELU_ADD_AUX_RELU) {
                    x[it].a0 = buf_load4(r_a0, o0, 0);
                }
            } else if constexpr (CLS == 3) {
                x[it].a0 = buf_load4(r_a0, o0, 0);
                x[it].a3 = buf_load4(r_a0, ok ? um * (4u * (unsigned)p.aux0_ld) + n4 + 4u * (unsigned)p.split : BUF_OOB, 0);
                x[it].a1 = buf_load4(r_a1, ok ? um * (4u * (unsigned)p.aux1_ld) + n4 : BUF_OOB, 0);
                x[it].a2 = buf_load4(r_a2, ok ? um * ld_a2 + n4 : BUF_OOB, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < EB; ++it) {
            const int m = mtab ? mtab[((lane >> 3) + (h + it) * 8) & 31] : mb + (h + it) * 8;
            const bool ok = (mtab ? m >= 0 : m < M) && h + it < it1;
            const unsigned um = (unsigned)m;
            const int srow = ((lane >> 3) + ((h + it) & 3) * 8) * EPI_S + (lane & 7) * 4;
            f32x4 acc = *reinterpret_cast<const f32x4*>(sW + srow);
            for (int k = 1; k < nparts; ++k) {       // same order as the register reduction it replaces
                const f32x4 t = *reinterpret_cast<const f32x4*>(sW + k * pstride + srow);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += t[e];
            }
            const EpiAux& a = x[it];
            f32x4 v, o;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[e] + bias4[e];
            if constexpr (CLS == 2) {
                if (has_add) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a.a2[e];
                }
            }
            o = v;
            bool to_out = true;
            switch (epi) {
                case EPI_RELU:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.f);
                    break;
                case EPI_SIGMOID:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = sigmoidf_(v[e]);
                    break;
                case EPI_TANH:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = tanhf(v[e]);
                    break;
                case EPI_SUB_FROM_AUX:
                    if constexpr (CLS == 1 || CLS == 2) {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = a.a0[e] - v[e];
                        break;
                    }
                    break;
                case EPI_ADD_AUX_SHRINK:
                    if constexpr (CLS == 1 || CLS == 2) {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = v[e] + a.a0[e];
                            o[e] = fmaxf(t - lam4[e], 0.f) - fmaxf(-t - lam4[e], 0.f);
                        }
                        break;
                    }
                    break;
                case EPI_RELU_ADD_AUX:
                    if constexpr (CLS == 1 || CLS == 2) {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(v[e], 0.f) + a.a0[e];
                        break;
                    }
                    break;
                case EPI_RELU_ADD_AUX_RELU:
                    if constexpr (CLS == 1 || CLS == 2) {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(a.a0[e] + fmaxf(v[e], 0.f), 0.f);
                        break;
                    }
                    break;
                case EPI_LSTC: {
                    if constexpr (CLS == 3) {
                        // aux0 = sigmoid gates [px][2*split] (i | f), aux1 = z0, aux2 = c_prev; out = z, out2 = c
                        f32x4 c;
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            c[e] = a.a3[e] * a.a2[e] + a.a0[e] * a.a1[e];
                            o[e] = sigmoidf_(v[e]) * tanhf(c[e]);
                        }
                        buf_store4(r_out2, ok ? um * (4u * (unsigned)p.out2_ld) + n4 : BUF_OOB, c);
                    } break;
                    }
                    break;
                case EPI_GRU_ZR:
                    if constexpr (CLS == 2) {
                        if (zr_hi) {
                            to_out = false;
                            f32x4 r;
    #pragma unroll
                            for (int e = 0; e < 4; ++e) r[e] = sigmoidf_(v[e]) * a.a0[e];
                            buf_store4(r_out2, ok ? um * (4u * (unsigned)p.out2_ld) + n4_hi : BUF_OOB, r);
                        } else {
    #pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = sigmoidf_(v[e]);
                        }
                        break;
                    }
                    break;
                case EPI_GRU_Q:
                    if constexpr (CLS == 2) {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (1.f - a.a0[e]) * a.a1[e] + a.a0[e] * tanhf(v[e]);
                        break;
                    }
                    break;
                case EPI_TANH_RELU_SPLIT:
                    if (sp_hi) {
                        to_out = false;
                        f32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], 0.f);
                        buf_store4(r_out2, ok ? um * (4u * (unsigned)p.out2_ld) + n4_hi : BUF_OOB, r);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = tanhf(v[e]);
                    }
                    break;
                case EPI_SCALE:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[e] * p.scale;
                    break;
                case EPI_BIAS_SCALE:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[e] * p.scale;
                    break;
                case EPI_LSTM_ACT:
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (n < p.split) ? sigmoidf_(v[e]) : tanhf(v[e]);
                    break;
                case EPI_LSTM_CELL:
                    if constexpr (CLS == 1) {
                        // quad = in | remember | out | cell pre-activations of hidden channel n/4 (base_layers.py:117-132)
                        to_out = false;
                        const float c = sigmoidf_(v[1]) * a.a0[0] + sigmoidf_(v[0]) * tanhf(v[3]);
                        buf_store1(r_out2, ok ? um * (4u * (unsigned)p.out2_ld) + (unsigned)n : BUF_OOB, c);
                        buf_store1(r_out, ok ? um * (4u * (unsigned)p.out_ld) + (unsigned)n : BUF_OOB, sigmoidf_(v[2]) * tanhf(c));
                    }
                    break;
                default: break;
            }
            buf_store4(r_out, (ok && to_out) ? um * (4u * (unsigned)p.out_ld) + n4 : BUF_OOB, o);
        }
    }
}

// generic tail: any alignment, partial quads, strided outputs (out_cs != 1), EPI_ADD_AUX
template <class CP>
__device__ __forceinline__ void patch_tail(const CP& p, const float* sW, int b, int mrow0, int nbase, int lane, int M,
                                           int it0 = 0, int it1 = 4, int nparts = 1, int pstride = 0, const int* mtab = nullptr) {
    if (p.epi_vec && nbase + 32 <= p.cout) {      // wave-uniform
        const int epi = p.epi;
        const bool one_aux = epi == EPI_SUB_FROM_AUX || epi == EPI_ADD_AUX_SHRINK || epi == EPI_RELU_ADD_AUX ||
                             epi == EPI_RELU_ADD_AUX_RELU || epi == EPI_LSTM_CELL;
        if (epi == EPI_LSTC) patch_tail_fast<1, 3>(p, sW, b, mrow0, nbase, lane, M, it0, it1, nparts, pstride, mtab);
        else if (p.addend || epi == EPI_GRU_ZR || epi == EPI_GRU_Q) patch_tail_fast<2, 2>(p, sW, b, mrow0, nbase, lane, M, it0, it1, nparts, pstride, mtab);
        else if (one_aux) patch_tail_fast<4, 1>(p, sW, b, mrow0, nbase, lane, M, it0, it1, nparts, pstride, mtab);
        else patch_tail_fast<4, 0>(p, sW, b, mrow0, nbase, lane, M, it0, it1, nparts, pstride, mtab);
        return;
    }
    const int nb = nbase + (lane & 7) * 4;
    const int mb = mrow0 + (lane >> 3);
#pragma unroll 1
    for (int it = it0; it < it1; ++it) {
        const int m = mtab ? mtab[((lane >> 3) + it * 8) & 31] : mb + it * 8;
        const int srow = ((lane >> 3) + it * 8) * EPI_S + (lane & 7) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(sW + srow);
        for (int k = 1; k < nparts; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(sW + k * pstride + srow);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += t[e];
        }
        if ((mtab ? m >= 0 : m < M) && nb < p.cout) epilogue4(p, b, m, nb, v);
    }
}

// WK > 1: intra-workgroup split-K.  The 4 waves are arranged WAVES_M x WAVES_N x WK; one stage holds
// 16*WK k-columns and wave (.., wk) consumes columns [16*wk, 16*wk+16).  The partial accumulators are summed
// through LDS before the epilogue.  This keeps 4 waves busy on 32x32 / 32x64 output tiles, which is what the
// 1/8-resolution layers (M = 768 pixels per image) need to fill 256 CUs.
// PREC: 0 = v_mfma_f32_32x32x2_f32 on fp32 operands (exact fp32 products);
//       3 = "f16x3": operands split hi+lo into f16 while staging, 3 x v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo +
//           lo*hi, fp32 accumulate) -- fp32-grade products at 16/3 of the fp32-MFMA rate;
//       1 = plain f16 operands, one MFMA (the reduced-precision mode BASELINE configs[4] names).
// In the f16 modes an LDS row holds, per 16-column chunk, [16 x hi | 16 x lo] (64 bytes): the same bytes as
// 16 fp32, so tile geometry, strides and the pre-split weight copies (same format in HBM) are shared.
// KCW: k-columns one wave consumes per stage (16 or 32).  32 doubles the MFMA work between two barriers, which
// is what the small tiles need (their fixed per-stage cost -- iterator, waits, barrier -- rivals 8 MFMAs).
// Fused InstanceNorm statistics: the wave's 32-pixel x 32-channel patch (raw accumulators, row stride EPI_S)
// is summed down its columns in fp64 -- lane = channel, the two half-waves take 16 rows each -- over
// v = acc + bias exactly as stored, and written to st_partial[b][patch][cout][2] (every element once).
// mtab (nullable): row validity comes from the row -> pixel table and the patch is number `patch_id` of `npatch_` per image
// (Winograd tiles); otherwise rows mrow0.. of the image and patch number mrow0 / 32 of ceil(M / 32)
template <class CP>
__device__ __forceinline__ void patch_stats(const CP& p, const float* sW, int b, int mrow0, int nbase, int lane,
                                            int M, const int* mtab = nullptr, int patch_id = 0, int npatch_ = 0) {
    const int c = lane & 31, half = lane >> 5;
    const int n = nbase + c;
    const int nvalid = mtab ? 32 : M - mrow0;          // rows of this patch inside the image (<= 0: patch is all padding)
    const float bv = (p.bias && n < p.cout) ? p.bias[(long)wgroup(p, b) * p.bias_gs + n] : 0.f;
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = half * 16 + r;
        if (mtab ? mtab[row] >= 0 : row < nvalid) {
            const double d = (double)(sW[row * EPI_S + c] + bv);
            s += d;
            ss += d * d;
        }
    }
    s += __shfl_xor(s, 32);
    ss += __shfl_xor(ss, 32);
    if (half == 0 && nvalid > 0 && n < p.cout) {
        const long npatch = mtab ? npatch_ : (M + 31) >> 5;
        const long pid = mtab ? patch_id : (mrow0 >> 5);
        double2 o;
        o.x = s;
        o.y = ss;
        *reinterpret_cast<double2*>(p.st_partial + (((long)b * npatch + pid) * p.cout + n) * 2) = o;
    }
}


// helpers of the LDS-DMA kernel (kept out of the __global__ body: the host pass has no such builtins)
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rs, float* lds_dst, unsigned voff, unsigned soff) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void wait_vmcnt_le(int n) {   // n is wave-uniform, <= 12
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // 0, and anything larger (safe)
    }
}
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// LDS-DMA kernels (conv_dma_kernel, conv_wino*_kernel) address a source image with 32-bit byte offsets through a buffer
// resource of BUF_RECORDS bytes; beyond that the hardware range check would hand back zeros instead of failing
static inline bool dma_range_ok(const ConvParams& p) {
    for (int i = 0; i < p.nseg; ++i)
        if ((long)p.Hin * p.Win * p.seg_ld[i] * 4L >= 0x7FFFFF00L) return false;
    return true;
}


// ---- Winograd F(2x2,3x3) geometry shared by conv_wino_kernel (conv_igemm.hip) and its split-K form (conv_wino_sk.hip) ----
// tiles per region: 4 x 8 (8 x 16 output pixels, "wide") or 8 x 4 (16 x 8, "tall"), whichever wastes fewer pixels on the
// image's ragged edge (90 x 120: 96 x 128 = +13.8 % wide, 96 x 120 = +6.7 % tall); the patch is 10 x 18 or 18 x 10 pixels
static constexpr int WG_PIX = 180;
__host__ __device__ inline int wino_tall(int Ho, int Wo) {
    const long wide = (long)((Ho + 7) / 8 * 8) * ((Wo + 15) / 16 * 16), tall = (long)((Ho + 15) / 16 * 16) * ((Wo + 7) / 8 * 8);
    return tall < wide ? 1 : 0;
}
static constexpr int WG_KC = 8;                             // channels per chunk
static constexpr int WG_PLANE = 192;                        // cells per channel-quad plane of a raw buffer (180 live)
static constexpr int WG_RAW = 512 * 4;                      // floats per raw buffer: two planes of 16-byte slots, padded to 2 x 256 slots
static constexpr int WG_UV = 16 * 32 * WG_KC;               // floats of a chunk's U block (4096)


#ifdef CF_CENSUS
// residency census (tools/census_probe.py): wave 0 of every workgroup records where and when it ran --
// [HW_ID, XCC_ID, start, end] (100 MHz real-time counter) -- so that the number of workgroups a CU really holds at once can be counted
__device__ __forceinline__ void census_begin(const ConvParams& p, long long& t0) { t0 = (long long)__builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void census_end(const ConvParams& p, long long t0) {
    if (p.stamp && threadIdx.x == 0) {
        long long* q = p.stamp + (long)blockIdx.x * 4;
        q[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        q[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
        q[2] = t0;
        q[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}
#endif

}  // namespace cf
