// mfma_shape_probe.hip -- does the fp32 MFMA shape change the clock the chip holds?  (tuning probe, GPU box only)
//   hipcc -O3 --offload-arch=gfx950 -o tools/probe/mfma_shape_probe.bin tools/probe/mfma_shape_probe.hip && tools/probe/mfma_shape_probe.bin
// Each wave re-reads its A / B fragments from LDS (ds_read_b128, random data) and accumulates a 32x32 tile per 16-deep
// k chunk, either as 8 x v_mfma_f32_32x32x2_f32 or as 16 x v_mfma_f32_16x16x4_f32 (same flops, same LDS bytes).
// Prints TFLOP/s and the shader clock (s_memtime / s_memrealtime) for 1, 2 and 4 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int RS = 68;     // LDS row stride in floats: 16-byte quads of consecutive rows fall in different bank groups
template <int SHAPE, int TM, int LD>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ out, long long* clk, int iters, unsigned tmask) {
    __shared__ __attribute__((aligned(16))) float lds[64 * RS];      // 17 KB of operands
    for (int i = threadIdx.x; i < 64 * RS; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 0xFFFFF];
    __syncthreads();
    const f32x4* tbl = reinterpret_cast<const f32x4*>(src);         // 4 MiB table: L2-resident global traffic (LD loads / lane / iteration)
    unsigned gidx = (blockIdx.x * 256 + threadIdx.x) & tmask;
    f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[TM][TM];
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TM; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int lr = lane & 31, lh = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < LD; ++l) { const f32x4 g = tbl[gidx]; gidx = (gidx + 262144 + 64) & tmask; gacc += g; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 a[TM], b[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&lds[(((wave * 7 + it + i * 3) & 31) + lr) % 64 * RS + (ks * 2 + lh) * 4 + ((it & 3) * 16)]);
#pragma unroll
                for (int j = 0; j < TM; ++j) b[j] = *reinterpret_cast<const f32x4*>(&lds[(((wave * 5 + it + j * 5 + 11) & 31) + lr) % 64 * RS + (ks * 2 + lh) * 4 + ((it & 3) * 16)]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TM; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    } else {
        f32x4 acc[2 * TM][2 * TM];
        for (int i = 0; i < 2 * TM; ++i) for (int j = 0; j < 2 * TM; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int r16 = lane & 15, kq = lane >> 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < LD; ++l) { const f32x4 g = tbl[gidx]; gidx = (gidx + 262144 + 64) & tmask; gacc += g; }
            f32x4 a[2 * TM], b[2 * TM];
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&lds[(((wave * 7 + it + i * 3) & 31) + r16 + 16 * (i & 1)) % 64 * RS + kq * 4 + ((it & 3) * 16)]);
#pragma unroll
            for (int j = 0; j < 2 * TM; ++j) b[j] = *reinterpret_cast<const f32x4*>(&lds[(((wave * 5 + it + j * 5 + 11) & 31) + r16 + 16 * (j & 1)) % 64 * RS + kq * 4 + ((it & 3) * 16)]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2 * TM; ++i) for (int j = 0; j < 2 * TM; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sum + gacc[0] + gacc[1] + gacc[2] + gacc[3];
    if (threadIdx.x == 0) clk[blockIdx.x] = r1 > r0 ? (t1 - t0) * 100 / (r1 - r0) : 0;
}

template <int SHAPE, int TM, int LD = 0>
static void run(const char* name, int wgs_per_cu, const float* src, float* out, long long* clk, int iters = 20000, int reps = 6, unsigned tmask = 0x3FFFF) {
    const int grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < (reps > 100 ? 2000 : 3); ++w) hipLaunchKernelGGL((probe<SHAPE, TM, LD>), dim3(grid), dim3(256), 0, 0, src, out, clk, iters, tmask);   // warm: let the clock settle
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<SHAPE, TM, LD>), dim3(grid), dim3(256), 0, 0, src, out, clk, iters, tmask);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), clk, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0; for (auto v : c) m += (double)v; m /= grid;
    const double flops = (double)reps * grid * 4 /*waves*/ * iters * (double)(TM * TM) * 32.0 * 32.0 * 16.0 * 2.0;
    const double gbytes = (double)reps * grid * 256.0 * iters * LD * 16.0;
    printf("%-24s table %4u MiB TM %d LD %d iters %6d  %d WG/CU: %9.3f ms/launch  %7.2f TFLOP/s  L2->reg %6.2f TB/s  shader clock %5.0f MHz\n", name, (tmask + 1) / 65536, TM, LD, iters,
           wgs_per_cu, ms / reps, flops / (ms * 1e-3) / 1e12, gbytes / (ms * 1e-3) / 1e12, m);
    fflush(stdout);
}

int main() {
    float *src, *out; long long* clk;
    hipMalloc(&src, (size_t)(1 << 27) * sizeof(float));      // 512 MiB: the first 4 MiB double as the L2-resident table
    hipMemset(src, 0, (size_t)(1 << 27) * sizeof(float)); hipMalloc(&out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&clk, 256 * 8 * sizeof(long long));
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int wg : {1, 4}) {
        run<32, 1>("v_mfma_f32_32x32x2_f32", wg, src, out, clk);
        run<16, 1>("v_mfma_f32_16x16x4_f32", wg, src, out, clk);
    }
    run<32, 2>("v_mfma_f32_32x32x2_f32", 2, src, out, clk);
    run<16, 2>("v_mfma_f32_16x16x4_f32", 2, src, out, clk);
    // short kernels back to back (~60 us each, like the conv launches of a step): does burstiness change the clock?
    run<32, 1>("32x32x2 short kernels", 4, src, out, clk, 60, 2000);
    run<16, 1>("16x16x4 short kernels", 4, src, out, clk, 60, 2000);
    // + global (L2-resident) traffic next to the MFMAs: 1 / 2 / 4 16-byte loads per lane per 16-deep k chunk
    run<32, 1, 1>("32x32x2 + L2 traffic", 4, src, out, clk);
    run<32, 1, 2>("32x32x2 + L2 traffic", 4, src, out, clk);
    run<32, 1, 4>("32x32x2 + L2 traffic", 4, src, out, clk);
    run<16, 1, 2>("16x16x4 + L2 traffic", 4, src, out, clk);
    run<32, 2, 4>("32x32x2 + L2 traffic", 2, src, out, clk);
    run<32, 1, 2>("32x32x2 + L2, short", 4, src, out, clk, 60, 2000);
    // the same loads out of a 128 MiB (Infinity Cache) and a 512 MiB (HBM) table
    run<32, 1, 1>("32x32x2 + MALL traffic", 4, src, out, clk, 20000, 6, 0x7FFFFF);
    run<32, 1, 2>("32x32x2 + MALL traffic", 4, src, out, clk, 20000, 6, 0x7FFFFF);
    run<32, 1, 1>("32x32x2 + HBM traffic", 4, src, out, clk, 20000, 6, 0x1FFFFFF);
    run<32, 1, 2>("32x32x2 + HBM traffic", 4, src, out, clk, 20000, 6, 0x1FFFFFF);
    return 0;
}
