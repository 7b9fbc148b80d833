// mfma_filler_probe.hip -- do VALU / LDS fillers hide under fp32 MFMAs on gfx950?  One wave per SIMD (every CU), a chain of
// v_mfma_f32_32x32x2_f32 (64 cycles each) or v_mfma_f32_16x16x4_f32 (32) with FILL independent v_fma_f32 between two MFMAs; and the same
// fillers issued by a SECOND wave on the SIMD (two workgroups per CU: one runs MFMAs only, the other VALU only).  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND, int FILL>
__global__ __launch_bounds__(256) void fill_loop(const float* __restrict__ src, float* __restrict__ dst, long long* __restrict__ stamps, int iters) {
    const int tid = threadIdx.x;
    float a = src[tid], b = src[256 + tid];
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = src[tid + i];
    float sink = 0.f;
    long long c0, c1;
    if (KIND == 0) {
        f16v acc = {};
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < FILL; ++i) { f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f); asm volatile("" : "+v"(f[i])); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        c1 = __builtin_readcyclecounter();
        for (int i = 0; i < 16; ++i) sink += acc[i];
    } else {
        f4v acc = {};
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < FILL; ++i) { f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f); asm volatile("" : "+v"(f[i])); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        c1 = __builtin_readcyclecounter();
        for (int i = 0; i < 4; ++i) sink += acc[i];
    }
    for (int i = 0; i < 12; ++i) sink += f[i];
    dst[(size_t)blockIdx.x * 256 + tid] = sink;
    if (tid == 0) stamps[blockIdx.x] = c1 - c0;
}

// one workgroup of EIGHT waves per CU: waves 0..3 (one per SIMD) run an MFMA chain, waves 4..7 (their SIMD partners) run what `partner` says:
// 0 = nothing (exit), 1 = an MFMA chain too, 2 = a VALU chain (12 independent v_fma_f32 per slot), 3 = an LDS-read chain (ds_read_b128)
__global__ __launch_bounds__(512) void duo_loop(const float* __restrict__ src, float* __restrict__ dst, long long* __restrict__ stamps, int iters, int partner) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = src[i & 1023];
    __syncthreads();
    const int my = wave < 4 ? 1 : partner;
    float a = src[tid & 255], b = src[256 + (tid & 255)];
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = src[(tid & 255) + i];
    float sink = 0.f;
    f16v acc = {};
    const long long c0 = __builtin_readcyclecounter();
    if (my == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (my == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) { f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f); asm volatile("" : "+v"(f[i])); }
    } else if (my == 3) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* lp = reinterpret_cast<const f4*>(lds) + (tid & 63);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                f4 v0 = lp[0], v1 = lp[64], v2 = lp[128], v3 = lp[192];
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                f[0] += v0[0] + v1[1] + v2[2] + v3[3];
            }
    }
    const long long c1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) sink += acc[i];
    for (int i = 0; i < 12; ++i) sink += f[i];
    dst[(size_t)blockIdx.x * 512 + tid] = sink;
    if ((tid & 63) == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int KIND, int FILL>
void run(const float* src, float* dst, long long* stamps, int ncu) {
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((fill_loop<KIND, FILL>), dim3(ncu), dim3(256), 0, 0, src, dst, stamps, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> st(ncu);
    CK(hipMemcpy(st.data(), stamps, ncu * 8, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    const int nom = KIND == 0 ? 64 : 32;
    printf("%-8s + %2d v_fma_f32 per MFMA (one wave per SIMD): %6.1f cycles per MFMA (MFMA alone %d, fillers alone %d)\n", KIND == 0 ? "32x32x2" : "16x16x4", FILL,
           (double)st[ncu / 2] / (iters * 8.0), nom, 4 * FILL);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<float> h(1024, 1.25f);
    float *src, *dst; long long* stamps; int* role;
    CK(hipMalloc(&src, 4096)); CK(hipMalloc(&dst, (size_t)ncu * 2 * 1024)); CK(hipMalloc(&stamps, ncu * 8 * 8)); (void)role;
    CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
    run<0, 0>(src, dst, stamps, ncu); run<0, 4>(src, dst, stamps, ncu); run<0, 8>(src, dst, stamps, ncu); run<0, 12>(src, dst, stamps, ncu);
    run<1, 0>(src, dst, stamps, ncu); run<1, 2>(src, dst, stamps, ncu); run<1, 4>(src, dst, stamps, ncu); run<1, 6>(src, dst, stamps, ncu);
    const int iters = 1000;
    const char* what[4] = {"nothing", "an MFMA chain", "a VALU chain (12 v_fma_f32 per slot)", "an LDS chain (4 ds_read_b128 per slot)"};
    for (int partner = 0; partner < 4; ++partner) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(duo_loop, dim3(ncu), dim3(512), 0, 0, src, dst, stamps, iters, partner);
        CK(hipDeviceSynchronize());
        std::vector<long long> st(ncu * 8);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<long long> m, v;
        for (int i = 0; i < ncu * 8; ++i) ((i & 7) < 4 ? m : v).push_back(st[i]);
        std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
        printf("MFMA wave (32x32x2 chain) with a SIMD partner running %-40s: %6.1f cycles per MFMA; the partner: %6.1f cycles per slot\n", what[partner],
               (double)m[m.size() / 2] / (iters * 8.0), (double)v[v.size() / 2] / (iters * 8.0));
    }
    return 0;
}
