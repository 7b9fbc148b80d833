// hip_time_shim.c -- LD_PRELOAD shim: host time spent inside the HIP calls a step is made of (round 4, B = 1 regime).
//   gcc -O2 -shared -fPIC -o tools/probes/hip_time_shim.so tools/probes/hip_time_shim.c -ldl
//   LD_PRELOAD=$PWD/tools/probes/hip_time_shim.so python bench.py --batch 1 ...      (prints totals at exit)
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
typedef int (*launch_t)(const void*, unsigned long long, unsigned, unsigned long long, unsigned, void**, size_t, void*);
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static double t_launch, t_rec, t_wait, t_other; static long n_launch, n_rec, n_wait, n_other;
static void report(void) {
    fprintf(stderr, "[hip_time_shim] hipLaunchKernel %ld calls %.1f ms (%.2f us each) | hipEventRecord %ld calls %.1f ms (%.2f us) | hipStreamWaitEvent %ld calls %.1f ms (%.2f us) | hipMemsetAsync/MemcpyAsync %ld calls %.1f ms\n",
            n_launch, t_launch / 1e3, n_launch ? t_launch / n_launch : 0.0, n_rec, t_rec / 1e3, n_rec ? t_rec / n_rec : 0.0, n_wait, t_wait / 1e3, n_wait ? t_wait / n_wait : 0.0, n_other, t_other / 1e3);
}
void hip_shim_reset(void) { t_launch = t_rec = t_wait = t_other = 0; n_launch = n_rec = n_wait = n_other = 0; }
void hip_shim_report(void) { report(); }
static void* sym(const char* name) {
    static int reg = 0;
    (void)reg;
    void* p = dlsym(RTLD_NEXT, name);
    if (!p) {       // the runtime came in through a dlopen with local scope (python -> torch): ask its handle
        static void* hip = 0;
        const char* names[] = {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", 0};
        for (int i = 0; !hip && names[i]; ++i) hip = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
        for (int i = 0; !hip && names[i]; ++i) hip = dlopen(names[i], RTLD_NOW);
        if (hip) p = dlsym(hip, name);
    }
    if (!p) { fprintf(stderr, "[hip_time_shim] %s not found\n", name); abort(); }
    return p;
}
struct dim3_ { unsigned x, y, z; };
int hipLaunchKernel(const void* f, struct dim3_ g, struct dim3_ b, void** args, size_t shm, void* st) {
    static int (*real)(const void*, struct dim3_, struct dim3_, void**, size_t, void*);
    if (!real) real = sym("hipLaunchKernel");
    const double t0 = now(); const int rc = real(f, g, b, args, shm, st); t_launch += now() - t0; ++n_launch; return rc;
}
int hipEventRecord(void* ev, void* st) {
    static int (*real)(void*, void*);
    if (!real) real = sym("hipEventRecord");
    const double t0 = now(); const int rc = real(ev, st); t_rec += now() - t0; ++n_rec; return rc;
}
int hipStreamWaitEvent(void* st, void* ev, unsigned flags) {
    static int (*real)(void*, void*, unsigned);
    if (!real) real = sym("hipStreamWaitEvent");
    const double t0 = now(); const int rc = real(st, ev, flags); t_wait += now() - t0; ++n_wait; return rc;
}
int hipMemsetAsync(void* p, int v, size_t n, void* st) {
    static int (*real)(void*, int, size_t, void*);
    if (!real) real = sym("hipMemsetAsync");
    const double t0 = now(); const int rc = real(p, v, n, st); t_other += now() - t0; ++n_other; return rc;
}
