// probe: buffer_load ... lds (LDS-DMA) semantics on gfx950: lane placement and out-of-range behaviour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* out, int nvalid) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = -7.f;           // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7FFFFF00, 0x00020000);
    // lane i fetches quad (63 - i) of the source, or an out-of-range offset
    unsigned voff = lane < nvalid ? (unsigned)(63 - lane) * 16u : 0x7FFFFF80u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 16), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, 48);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    printf("before patch: %g %g\n", r[0], r[15]);
    for (int l : {0, 1, 2, 47, 48, 63}) printf("lane %d -> lds[%d..]: %g %g %g %g\n", l, 16 + 4 * l, r[16 + 4 * l], r[17 + 4 * l], r[18 + 4 * l], r[19 + 4 * l]);
    printf("after patch: %g\n", r[16 + 256]);
    return 0;
}
