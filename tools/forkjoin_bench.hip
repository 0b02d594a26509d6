// forkjoin_bench.hip -- what does a fork / join over two HIP streams cost on this box?  (decides whether independent
// stages of one key switch could be overlapped on internal streams)
// A: 4 kernels on one stream.  B: fork (event on main, two side streams wait), 2 kernels on each side stream, join.
// Each kernel occupies half of the CUs for ~10 us, so B's ideal is A / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long cycles, unsigned *sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned v = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < cycles) v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeefu) *sink = v;
}
int main() {
    unsigned *sink; CK(hipMalloc(&sink, 4));
    hipStream_t m, s1, s2; CK(hipStreamCreate(&m)); CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, e2, t0, t1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const unsigned long long cyc = 1000;   // 100 MHz counter: 10 us
    const int reps = 200;
    auto K = [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(256), 0, s, cyc, sink); };
    for (int w = 0; w < 2; w++) {
        CK(hipEventRecord(t0, m));
        for (int r = 0; r < reps; r++) { K(m); K(m); K(m); K(m); }
        CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1));
        float a; CK(hipEventElapsedTime(&a, t0, t1));
        CK(hipEventRecord(t0, m));
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, m));
            CK(hipStreamWaitEvent(s1, e0, 0)); CK(hipStreamWaitEvent(s2, e0, 0));
            K(s1); K(s1); K(s2); K(s2);
            CK(hipEventRecord(e1, s1)); CK(hipEventRecord(e2, s2));
            CK(hipStreamWaitEvent(m, e1, 0)); CK(hipStreamWaitEvent(m, e2, 0));
        }
        CK(hipEventRecord(t1, m)); CK(hipEventSynchronize(t1));
        float b; CK(hipEventElapsedTime(&b, t0, t1));
        if (w) printf("4 kernels on one stream: %.1f us per group; fork + 2x2 kernels on two streams + join: %.1f us per group (ideal %.1f)\n",
                      a * 1000 / reps, b * 1000 / reps, a * 1000 / reps / 2);
    }
    return 0;
}
