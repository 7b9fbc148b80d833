// mfma_power_probe.hip -- what does the fp32 matrix pipe of an MI355X sustain on REAL operand data?  (round 4)
//
// Bare loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, operands in registers, every CU busy, 1 / 2 / 4 waves per SIMD, on
// all-zero and on random-normal operands, in launches of ~1 ms repeated for ~0.3 s so that the power management has settled.  Prints
// TFLOP/s (HIP events), the shader clock over the launch (s_memtime cycles / s_memrealtime at 100 MHz, wave 0 of every workgroup,
// median) and the matrix-pipe duty (MFMA cycles issued / wave cycles).  Build: hipcc -O3 --offload-arch=gfx950 (tools/probes/build.sh).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NSET operand sets rotate through the loop, so consecutive MFMAs see different A / B bits (as in a real kernel); 4 accumulators.
template <int KIND, int NSET>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ src, float* __restrict__ dst, long long* __restrict__ stamps,
                                                 int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    float a[NSET], b[NSET];
    for (int i = 0; i < NSET; ++i) {
        a[i] = src[(i * 2 + 0) * 256 + tid];
        b[i] = src[(i * 2 + 1) * 256 + tid];
    }
    const long long c0 = __builtin_readcyclecounter();
    const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
    float sink = 0.f;
    if (KIND == 0) {
        f16v acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + q) % NSET], b[s], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 16; ++i) sink += acc[q][i];
    } else {
        f4v acc[8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s + q) % NSET], b[s], acc[q], 0, 0, 0);
            }
        }
        for (int q = 0; q < 8; ++q)
            for (int i = 0; i < 4; ++i) sink += acc[q][i];
    }
    const long long c1 = __builtin_readcyclecounter();
    const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
    dst[(size_t)blockIdx.x * 256 + tid] = sink;
    if (tid == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    (void)lane;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("# %s, %d CUs; launches of ~1 ms x %d; flops/MFMA: 32x32x2 = 4096 (64 cyc), 16x16x4 = 2048 (32 cyc); spec 157.3 TFLOP/s at 2.4 GHz\n",
           prop.name, ncu, reps);
    constexpr int NSET = 8;
    std::vector<float> h(NSET * 2 * 256);
    float *src, *dst;
    long long* stamps;
    const int maxwg = ncu * 4;
    CK(hipMalloc(&src, h.size() * 4));
    CK(hipMalloc(&dst, (size_t)maxwg * 256 * 4));
    CK(hipMalloc(&stamps, (size_t)maxwg * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    printf("%-9s %-7s %5s %10s %10s %9s %9s\n", "mfma", "data", "w/SIMD", "us/launch", "TFLOP/s", "clk GHz", "duty");
    for (int kind = 0; kind < 2; ++kind)
        for (int data = 0; data < 3; ++data)
            for (int wps = 1; wps <= 4; wps *= 2) {
                // data 0: zeros; 1: random normal; 2: random normal, small magnitudes like Winograd-transformed weights (x 2^-6)
                for (auto& v : h) v = data == 0 ? 0.f : nd(rng) * (data == 2 ? 0.015625f : 1.f);
                CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
                const int nwg = ncu * wps;                                   // 4 waves per workgroup = one per SIMD
                const int per_it = NSET * (kind == 0 ? 4 : 8);
                const int cyc_per_it = per_it * (kind == 0 ? 64 : 32);
                const int iters = (int)(2.0e6 / wps / cyc_per_it);           // ~1 ms at 2 GHz
                auto go = [&]() {
                    if (kind == 0) hipLaunchKernelGGL((mfma_loop<0, NSET>), dim3(nwg), dim3(256), 0, 0, src, dst, stamps, iters);
                    else hipLaunchKernelGGL((mfma_loop<1, NSET>), dim3(nwg), dim3(256), 0, 0, src, dst, stamps, iters);
                };
                for (int i = 0; i < reps / 2; ++i) go();                     // settle the clocks
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) go();
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<long long> st(2 * nwg);
                CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
                std::vector<double> clk(nwg), duty(nwg);
                for (int i = 0; i < nwg; ++i) {
                    clk[i] = (double)st[2 * i] / ((double)st[2 * i + 1] * 10.0);   // cycles per ns = GHz
                    duty[i] = (double)iters * cyc_per_it * wps / (double)st[2 * i];
                }
                std::sort(clk.begin(), clk.end());
                std::sort(duty.begin(), duty.end());
                const double flops = (double)nwg * 4 * iters * per_it * (kind == 0 ? 4096.0 : 2048.0);
                const double us = ms * 1e3 / reps;
                printf("%-9s %-7s %5d %10.1f %10.1f %9.3f %9.3f\n", kind == 0 ? "32x32x2" : "16x16x4",
                       data == 0 ? "zero" : data == 1 ? "normal" : "small", wps, us, flops / (us * 1e-6) * 1e-12, clk[nwg / 2], duty[nwg / 2]);
                fflush(stdout);
            }
    return 0;
}
