// launch_cost_probe.hip -- host cost of the HIP calls a step is made of (round 4, B = 1 regime): asynchronous launches of an empty kernel
// with a ConvParams-sized (640-byte) argument on one stream, the same alternating over four streams, and an event record + cross-stream wait.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Big { int v[160]; };
__global__ void empty_kernel(const Big b, int* out) { if (b.v[0] == 12345) out[0] = b.v[1]; }
__global__ void spin_kernel(long long cycles, int* out) {      // holds the queue while the host enqueues behind it (bounded: exits after `cycles`)
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (cycles == 12345) out[0] = 1;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* out; CK(hipMalloc(&out, 64));
    hipStream_t st[4]; hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    Big b = {};
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        // (the queue holds a few thousand packets: N launches of an empty kernel never block on the GPU for long)
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], b, out);
        double t1 = now();
        CK(hipDeviceSynchronize());
        double t1b = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[i & 3], b, out);
        double t2 = now();
        CK(hipDeviceSynchronize());
        double t3 = now();
        for (int i = 0; i < N; ++i) { CK(hipEventRecord(ev[0], st[0])); CK(hipStreamWaitEvent(st[1], ev[0], 0)); }
        double t4 = now();
        CK(hipDeviceSynchronize());
        // a dependent chain across two streams: kernel on s0 -> event -> kernel on s1 -> event -> ...
        double t5 = now();
        for (int i = 0; i < N / 2; ++i) {
            hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], b, out);
            CK(hipEventRecord(ev[0], st[0])); CK(hipStreamWaitEvent(st[1], ev[0], 0));
            hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[1], b, out);
            CK(hipEventRecord(ev[1], st[1])); CK(hipStreamWaitEvent(st[0], ev[1], 0));
        }
        double t6 = now();
        CK(hipDeviceSynchronize());
        double t7 = now();
        // the same chain on ONE stream
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], b, out);
        CK(hipDeviceSynchronize());
        double t8 = now();
        // GPU-side cost of a dependent kernel: N empty kernels queued BEHIND a 10 ms spin kernel, so the host is out of the picture
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], 20000000LL, out);
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], b, out);
        CK(hipEventRecord(e1, st[0]));
        CK(hipDeviceSynchronize());
        float ms_q = 0; CK(hipEventElapsedTime(&ms_q, e0, e1));
        if (rep == 1) {
            printf("GPU side, kernels queued behind a spin kernel: %.2f us per dependent empty kernel on one stream\n", ms_q * 1e3 / N);
            printf("launch, one stream (640-byte argument):   %.2f us per call on the host; drained %.2f us per kernel end to end\n", (t1 - t0) / N, (t1b - t0) / N);
            printf("launch, four streams round-robin:         %.2f us per call on the host; drained %.2f us per kernel end to end\n", (t2 - t1b) / N, (t3 - t1b) / N);
            printf("event record + cross-stream wait:         %.2f us per pair on the host\n", (t4 - t3) / N);
            printf("dependent chain ping-pong over 2 streams: %.2f us per kernel on the host, %.2f us per kernel end to end\n", (t6 - t5) / N, (t7 - t5) / N);
            printf("dependent chain on one stream:            %.2f us per kernel end to end\n", (t8 - t7) / N);
        }
    }
    return 0;
}
