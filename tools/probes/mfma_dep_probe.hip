// mfma_dep_probe.hip -- issue rate of fp32 MFMAs as a function of how many INDEPENDENT accumulators rotate (one wave per SIMD, every CU):
// NACC = 1 is a chain on one accumulator, NACC = 2 the distance conv_wino16_kernel had (round 4).  Prints cycles per MFMA (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void dep_loop(const float* __restrict__ src, float* __restrict__ dst, long long* __restrict__ stamps, int iters) {
    const int tid = threadIdx.x;
    float a = src[tid], b = src[256 + tid];
    float sink = 0.f;
    long long c0, c1;
    if (KIND == 0) {
        f16v acc[NACC] = {};
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        c1 = __builtin_readcyclecounter();
        for (int q = 0; q < NACC; ++q) for (int i = 0; i < 16; ++i) sink += acc[q][i];
    } else {
        f4v acc[NACC] = {};
        c0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
        c1 = __builtin_readcyclecounter();
        for (int q = 0; q < NACC; ++q) for (int i = 0; i < 4; ++i) sink += acc[q][i];
    }
    dst[(size_t)blockIdx.x * 256 + tid] = sink;
    if (tid == 0) stamps[blockIdx.x] = c1 - c0;
}

template <int KIND, int NACC>
void run(const float* src, float* dst, long long* stamps, int ncu) {
    const int iters = 2000;
    hipLaunchKernelGGL((dep_loop<KIND, NACC>), dim3(ncu), dim3(256), 0, 0, src, dst, stamps, iters);
    hipLaunchKernelGGL((dep_loop<KIND, NACC>), dim3(ncu), dim3(256), 0, 0, src, dst, stamps, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> st(ncu);
    CK(hipMemcpy(st.data(), stamps, ncu * 8, hipMemcpyDeviceToHost));
    std::sort(st.begin(), st.end());
    printf("%-8s accumulators in rotation %d: %6.1f cycles per MFMA (nominal %d)\n", KIND == 0 ? "32x32x2" : "16x16x4", NACC,
           (double)st[ncu / 2] / (iters * 16.0), KIND == 0 ? 64 : 32);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<float> h(512, 1.25f);
    float *src, *dst; long long* stamps;
    CK(hipMalloc(&src, 2048)); CK(hipMalloc(&dst, (size_t)ncu * 1024)); CK(hipMalloc(&stamps, ncu * 8));
    CK(hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice));
    run<1, 1>(src, dst, stamps, ncu); run<1, 2>(src, dst, stamps, ncu); run<1, 4>(src, dst, stamps, ncu); run<1, 8>(src, dst, stamps, ncu);
    run<0, 1>(src, dst, stamps, ncu); run<0, 2>(src, dst, stamps, ncu); run<0, 4>(src, dst, stamps, ncu);
    return 0;
}
