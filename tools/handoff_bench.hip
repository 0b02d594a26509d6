// handoff_bench.hip -- memory-side potential of a ONE-launch 2^16 NTT whose intermediate stays in the XCD's L2.
// Butterflies removed (the patterns of tools/microbench.hip MB_PATTERN): a workgroup reads + rewrites its strided
// tile (pass 1), the 16 workgroups of a limb (all placed on one XCD: block b -> XCD b % 8) meet at a counter, then
// each reads + rewrites one contiguous 4096-coefficient chunk of the same limb (pass 2).  Hand-off: plain stores
// (the line stays in the XCD's L2), s_waitcnt vmcnt(0), barrier, one relaxed agent atomic per workgroup; the
// consumer polls with one lane and reads the chunk with 16-byte loads that bypass the CU's L1 (`nt` or `sc1`,
// MI355X_MICROARCH.md price list).  Compared against the same two patterns as two launches.
// Build: hipcc --offload-arch=gfx950 -O3 -o handoff_bench tools/handoff_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint64_t u64;
typedef unsigned long long u64x2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

struct Args {
    u64 *buf;
    unsigned *done;      // [limbs] arrival counters, monotonic over launches
    unsigned *mismatch;  // workgroups whose XCC_ID differs from b % 8
    int limbs;
    unsigned target;
};

// flags kept in the XCD's own L2: arrival = non-returning atomic without scope bits (executes in this XCD's L2),
// poll = returning atomic OR 0 (never served by the CU's L1).  Valid only because every participant is on one XCD.
__device__ __forceinline__ void l2_arrive(unsigned *p) {
    asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(1u) : "memory");
}
__device__ __forceinline__ unsigned l2_poll(unsigned *p) {
    unsigned r;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(0u) : "memory");
    return r;
}

__device__ __forceinline__ void pass1(u64 *limb, unsigned tile, unsigned t) {
    const unsigned c = tile * 16 + (t & 15), r0 = t >> 4;
    u64 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = limb[(r0 + 32 * j) * 256 + c];
#pragma unroll
    for (int j = 0; j < 8; j++) limb[(r0 + 32 * j) * 256 + c] = v[j] + 1;
}

template <int LOADMODE>  // 0 plain, 1 nt, 2 sc1
__device__ __forceinline__ void pass2(u64 *limb, unsigned tile, unsigned t) {
    u64x2v *p = reinterpret_cast<u64x2v *>(limb + tile * 4096) + t;
    u64x2v w[4];
    if (LOADMODE == 2) {
        const u64x2v *p1 = p + 512, *p2 = p + 1024, *p3 = p + 1536;
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(p), "v"(p1), "v"(p2), "v"(p3) : "memory");
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) w[j] = LOADMODE == 1 ? __builtin_nontemporal_load(p + 512 * j) : p[512 * j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { w[j].x += 1; w[j].y += 1; p[512 * j] = w[j]; }
}

template <int LOADMODE, bool WAIT, bool FLAGL2 = false>
__global__ __launch_bounds__(512) void fused_kernel(const Args a) {
    const unsigned b = blockIdx.x, xcd = b & 7, within = b >> 3, tile = within & 15, slot = within >> 4;
    const unsigned limb_i = slot * 8 + xcd;
    if ((int)limb_i >= a.limbs) return;
    u64 *limb = a.buf + (size_t)limb_i * 65536;
    const unsigned t = threadIdx.x;
    if (t == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        if ((id & 7) != xcd) atomicAdd(a.mismatch, 1u);
    }
    pass1(limb, tile, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (FLAGL2) l2_arrive(&a.done[limb_i]);
        else __hip_atomic_fetch_add(&a.done[limb_i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (WAIT) {
            unsigned spins = 0;   // bounded: a broken assumption must not hang the box
            while ((FLAGL2 ? l2_poll(&a.done[limb_i]) : __hip_atomic_load(&a.done[limb_i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < a.target) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 2000000u) { atomicAdd(a.mismatch + 1, 1u); break; }
            }
        }
    }
    if (WAIT) __syncthreads();
    pass2<LOADMODE>(limb, tile, t);
}

static int check(const u64 *d, size_t n, u64 add, const char *what);
// LAG form: workgroup (slot s, tile t) runs pass 1 on the limb of slot s, then pass 2 on the limb of slot s - 1, whose 16
// pass-1 workgroups were dispatched 16 positions earlier on this XCD and have normally finished by then: the flag is
// requested before the workgroup's own pass 1 and checked after it, so nobody spins in the common case.
// OCC_LDS: dynamic LDS per workgroup, to cap the residency at what the real kernel gets (3 workgroups per CU).
template <int LOADMODE, bool FLAGL2 = false, int LAG = 1>
__global__ __launch_bounds__(512) void lag_kernel(const Args a) {
    extern __shared__ unsigned char smem[];
    const unsigned b = blockIdx.x, xcd = b & 7, within = b >> 3, tile = within & 15, slot = within >> 4;
    const unsigned limb1 = slot * 8 + xcd;                 // pass-1 work
    const bool has1 = (int)limb1 < a.limbs, has2 = slot >= LAG && (int)(limb1 - 8 * LAG) < a.limbs;
    const unsigned limb2 = has2 ? limb1 - 8 * LAG : 0;     // pass-2 work: the limb of this XCD LAG slots back
    const unsigned t = threadIdx.x;
    unsigned early = 0;
    if (has2 && t == 0) early = FLAGL2 ? l2_poll(&a.done[limb2]) : __hip_atomic_load(&a.done[limb2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (has1) {
        pass1(a.buf + (size_t)limb1 * 65536, tile, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            if (FLAGL2) l2_arrive(&a.done[limb1]);
            else __hip_atomic_fetch_add(&a.done[limb1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!has2) return;
    if (t == 0 && early < a.target) {
        unsigned spins = 0;
        while ((FLAGL2 ? l2_poll(&a.done[limb2]) : __hip_atomic_load(&a.done[limb2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < a.target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 2000000u) { atomicAdd(a.mismatch + 1, 1u); break; }
        }
        atomicAdd(a.mismatch + 2, 1u);   // workgroups that had to spin at all
    }
    __syncthreads();
    if (smem[0] == 77 && t == 99999) a.mismatch[3] = 1;   // keep the dynamic LDS allocation alive
    pass2<LOADMODE>(a.buf + (size_t)limb2 * 65536, tile, t);
}

template <int LOADMODE, bool FLAGL2 = false, int LAG = 1>
static int run_lag(u64 *buf, unsigned *done, unsigned *mm, int limbs, unsigned &epoch, const char *name, size_t lds) {
    const size_t n = (size_t)limbs * 65536;
    std::vector<u64> init(n);
    for (size_t i = 0; i < n; i++) init[i] = i;
    CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((limbs + 7) / 8 + LAG) * 16 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    {
        epoch += 16;
        Args a{buf, done, mm, limbs, epoch};
        hipLaunchKernelGGL((lag_kernel<LOADMODE, FLAGL2, LAG>), dim3(blocks), dim3(512), lds, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
    }
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) {
        epoch += 16;
        Args a{buf, done, mm, limbs, epoch};
        hipLaunchKernelGGL((lag_kernel<LOADMODE, FLAGL2, LAG>), dim3(blocks), dim3(512), lds, 0, a);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h_mm[4] = {0, 0, 0, 0}; CK(hipMemcpy(h_mm, mm, 16, hipMemcpyDeviceToHost));
    printf("  lag   %-28s LDS %3zu KiB limbs %4d: %7.2f us/launch  (XCC mismatches %u, spin timeouts %u, workgroups that spun %u)\n", name,
           lds >> 10, limbs, ms * 1000 / reps, h_mm[0], h_mm[1], h_mm[2]);
    return check(buf, n, 2 * reps, name);
}

__global__ __launch_bounds__(512) void p1_kernel(u64 *buf) { pass1(buf + (size_t)blockIdx.y * 65536, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(512) void p2_kernel(u64 *buf) { pass2<0>(buf + (size_t)blockIdx.y * 65536, blockIdx.x, threadIdx.x); }

static int check(const u64 *d, size_t n, u64 add, const char *what) {
    std::vector<u64> h(n);
    if (hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += h[i] != i + add;
    printf("    check %-28s: %zu of %zu words wrong\n", what, bad, n);
    return bad != 0;
}

template <int LOADMODE, bool WAIT, bool FLAGL2 = false>
static int run_fused(u64 *buf, unsigned *done, unsigned *mm, int limbs, unsigned &epoch, const char *name, bool verify) {
    const size_t n = (size_t)limbs * 65536;
    std::vector<u64> init(n);
    for (size_t i = 0; i < n; i++) init[i] = i;
    CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((limbs + 7) / 8) * 16 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    {   // warm launch (code object load), not timed, not counted in the check
        epoch += 16;
        Args a{buf, done, mm, limbs, epoch};
        hipLaunchKernelGGL((fused_kernel<LOADMODE, WAIT, FLAGL2>), dim3(blocks), dim3(512), 0, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
    }
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) {
        epoch += 16;
        Args a{buf, done, mm, limbs, epoch};
        hipLaunchKernelGGL((fused_kernel<LOADMODE, WAIT, FLAGL2>), dim3(blocks), dim3(512), 0, 0, a);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h_mm[2] = {0, 0}; CK(hipMemcpy(h_mm, mm, 8, hipMemcpyDeviceToHost));
    printf("  fused %-34s limbs %4d: %7.2f us/launch  (XCC_ID mismatches so far: %u, spin timeouts: %u)\n", name, limbs, ms * 1000 / reps, h_mm[0], h_mm[1]);
    if (verify) return check(buf, n, 2 * reps, name);
    return 0;
}

int main() {
    const int max_limbs = 720;
    u64 *buf; CK(hipMalloc(&buf, (size_t)max_limbs * 65536 * 8));
    unsigned *done, *mm; CK(hipMalloc(&done, max_limbs * 4)); CK(hipMalloc(&mm, 16));
    CK(hipMemset(done, 0, max_limbs * 4)); CK(hipMemset(mm, 0, 16));
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned epoch = 0;
    for (int limbs : {48, 184, 720}) {
        const size_t n = (size_t)limbs * 65536;
        CK(hipMemset(done, 0, max_limbs * 4));
        epoch = 0;
        {   // two launches
            std::vector<u64> init(n);
            for (size_t i = 0; i < n; i++) init[i] = i;
            CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 20;
            hipLaunchKernelGGL(p1_kernel, dim3(16, limbs), dim3(512), 0, 0, buf);
            hipLaunchKernelGGL(p2_kernel, dim3(16, limbs), dim3(512), 0, 0, buf);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(buf, init.data(), n * 8, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                hipLaunchKernelGGL(p1_kernel, dim3(16, limbs), dim3(512), 0, 0, buf);
                hipLaunchKernelGGL(p2_kernel, dim3(16, limbs), dim3(512), 0, 0, buf);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("two launches                               limbs %4d: %7.2f us/pair   (%.2f TB/s over 4 x 8 B per coefficient)\n", limbs,
                   ms * 1000 / reps, 4.0 * n * 8 * reps / ms / 1e9);
            check(buf, n, 2 * reps, "two launches");
        }
        run_fused<2, true>(buf, done, mm, limbs, epoch, "sc1 loads, wait", true);
        run_fused<1, false>(buf, done, mm, limbs, epoch, "nt loads, no wait (wrong)", false);
        run_lag<2>(buf, done, mm, limbs, epoch, "sc1 loads, agent flags", 0);
        run_fused<2, true, true>(buf, done, mm, limbs, epoch, "sc1 loads, wait, L2 flags", true);
        run_lag<2, true, 1>(buf, done, mm, limbs, epoch, "sc1, L2 flags, lag 1", 0);
        run_lag<2, true, 1>(buf, done, mm, limbs, epoch, "sc1, L2 flags, lag 1", 48 << 10);
        run_lag<2, true, 2>(buf, done, mm, limbs, epoch, "sc1, L2 flags, lag 2", 48 << 10);
        run_lag<2, true, 4>(buf, done, mm, limbs, epoch, "sc1, L2 flags, lag 4", 48 << 10);
        run_lag<2, false, 4>(buf, done, mm, limbs, epoch, "sc1, agent flags, lag 4", 48 << 10);
    }
    return 0;
}
