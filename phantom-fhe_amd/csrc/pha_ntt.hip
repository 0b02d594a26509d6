// pha_ntt.hip -- NTT kernels and launchers (gfx950).
//
// One templated kernel per (pass configuration, direction, epilogue); the thread program is
// pha_ntt_core.h.  The 17 launchers of include/ntt.cuh:157-226 that the hot path uses collapse to
// two drivers (forward / inverse) parameterised by a limb selector (start, special-prime remap,
// excluded range: src/ntt/fntt_2d.cu:434-437, src/ntt/ntt_modup.cu:422) and an epilogue
// (canonicalise / fused mod-down / scale), instead of 25 hand-copied kernels.
#include "../../include/phantom_amd.h"
#include "pha_internal.h"
#include "pha_experiments.h"
#include "pha_ntt_core.h"

#include <atomic>
#include <type_traits>
#include <cstdlib>

namespace pha {

// Kernel selection.  The product library holds ONE plan per degree and launch size (constants below, chosen from the r01 / r02
// sweeps); every other geometry that was built and measured (16 coefficients per thread, 512-thread contiguous passes, the
// one-workgroup N = 2^14 plan, both passes in one launch with the L2 hand-off, ...) lives behind -DPHA_EXPERIMENTS in the test-only
// library libphantom_amd_exp.so (csrc/pha_experiments.h: pha_set_tuning), because the extra instantiations alone cost ~30 us per
// key switch when they sat in the product code object.
// Variant bits (experiments build: pha_set_tuning key 0): bit 0 = 8 coefficients/thread (else 16), bit 3 = integer butterflies for
// every prime (FP64 path off), bit 4 = on-the-fly twiddles in the contiguous pass (implies bit 0), bit 5 = bit 4 automatically for
// launches of >= 1024 tiles, bit 6 = one-wavefront workgroups in the contiguous pass (NttPlan variants 3 / 4), bit 7 = N = 4096
// through the two-pass plans too, bit 8 = the one-workgroup plans of N = 8192 / 16384 for every launch size, bit 9 / 10 = always /
// never both passes in one launch (L2 hand-off), bit 11 = polynomial-fastest, XCD-grouped block order in the contiguous pass of
// batched launches, bit 12 (r04) = N = 2^16 as 64 x 1024 (NttPlan<16, 10>: strided tiles of 64 rows x 64 columns, i.e. 512-byte runs and
// ONE exchange; 1024-point rows, two wavefronts each), bit 13 = the same with one wavefront x 16 coefficients per row (plan 12),
// bit 14 = 2^16 as 128 x 512 (plan 8).  r04 (profiles/r04_experiments.md): 720 limbs 333 -> 317 us, one 45-limb polynomial 28.4 -> 26.5 us.
#ifndef PHA_X_HOIST_C
#define PHA_X_HOIST_C 0    // r04 experiment: twiddle request schedule of the contiguous pass (0 per round, 1 all up front, 2 one round ahead)
#endif
#ifndef PHA_X_HOIST_S
#define PHA_X_HOIST_S 0
#endif
#ifndef PHA_X_VARIANT
#define PHA_X_VARIANT (1 | 32 | 64 | 2048 | 4096)
#endif
constexpr int kDefaultVariant = PHA_X_VARIANT;   // 8 coefficients per thread, one-wavefront contiguous pass, on-the-fly twiddles for >= 1024 tiles of fewer than 8 polynomials, polynomial-fastest block order in large batched contiguous passes, 2^16 = 64 x 1024   // 8 coefficients per thread, one-wavefront contiguous pass, on-the-fly twiddles for >= 1024 tiles of fewer than 8 polynomials, polynomial-fastest block order in large batched contiguous passes
#if defined(PHA_EXPERIMENTS)
std::atomic<int> g_ntt_variant{kDefaultVariant};
static inline int ntt_variant() { return g_ntt_variant.load(std::memory_order_relaxed); }
std::atomic<int> g_whole14_min{1 << 30};     // key 2: limb-polynomials per launch from which N = 2^14 takes the one-workgroup plan (r02: never faster)
std::atomic<int> g_fused_split{0};           // key 5: one pass per workgroup in the one-launch transform (r02: slower still)
std::atomic<int> g_fused_lag{2};             // key 3: lag (in units per XCD) between the two passes of the one-launch transform
std::atomic<int> g_fused_min_tiles{1 << 30}; // key 4: tiles per launch from which the two passes share one launch (r02: 7 % slower at every size)
extern std::atomic<int> g_bconv_split;       // pha_rns.hip (key 1)
#else
static constexpr int ntt_variant() { return kDefaultVariant; }
#endif
#if defined(PHA_EXP_STAMPS)
__device__ unsigned long long g_stamps[8];
__device__ unsigned long long g_wg_times[2048];
#endif

struct NttKArgs {
    const u64 *in;
    u64 *out;
    u64 *mid;                // buffer between the two passes (pass 1: in -> mid, pass 2: mid -> out)
    const u64x2 *tw;         // table base [prime][n] (forward or inverse)
    const DModulus *mod;     // [prime]
    const u64x2 *ninv;       // [prime]
    const u64x2 *w1ninv;     // [prime]
    const u64 *twf;          // FP64 path: table base [prime][n] of doubles W (forward or inverse)
    const u64x2 *ninvf, *w1ninvf;
    const FpInfo *fpinfo;    // [prime]; null = FP64 path off
    const u64 *scale;        // [limb] or null
    const u64 *scale_shoup;  // [limb] or null
    const u64 *aux;          // fuse_moddown: cx base
    const u64 *aux2;         // EPI_FWD_KSRESCALE: ct base; scale2 / scale2_shoup [limb]: PInv
    const u64 *scale2, *scale2_shoup;
    size_t aux2_stride;
    LimbSel sel;
    uint32_t log_n;
    uint32_t t1, t2;         // N = t1 * t2
    uint32_t active;         // processed limbs = sel.count minus the excluded range (pipelined kernel)
    uint32_t batch;          // polynomials per launch (blockIdx.z)
    size_t poly_stride, out_stride, aux_stride;
    size_t in_stride;        // elements between the polynomials of `in` (first pass; the second pass reads mid at poly_stride)
    bool first_pass_only;    // forward: the caller runs its own (fused) second pass
    bool second_pass_only;   // inverse: the contiguous pass was folded into the producer of `mid`
    bool first_pass_done;    // forward: the strided pass was folded into the producer of `mid` (modup_conv_s1_kernel)
    uint32_t excl_step, excl_limit, excl_mod;
    const u64 *pro_src;      // rescale prologue: every limb of polynomial z reads pro_src + z * pro_stride instead
    size_t pro_stride;
    // batched launches, polynomial-fastest order: a 1-D grid in which the `batch` polynomials of one (tile, limb) run back to
    // back on ONE XCD (block b -> XCD b % 8), so that the twiddle rows they share are fetched into that L2 once
    uint32_t zfast_tiles;    // 0 = plain 3-D grid (tile, limb, polynomial); else tiles per limb of the 1-D form
    uint32_t zfast_run;      // tiles of one polynomial that run back to back before the next polynomial's (a multiple of 8)
    const u64 *h_primes;     // HOST copy of the context's primes (launchers only: which limbs run on the FP64 back end)
};

// Per-tile arguments of limb `twr` (absolute limb index in the buffer), tile `tile`.
template <bool FWD, int EPI, bool FOLD>
__device__ __forceinline__ void tile_args(const NttKArgs &k, uint32_t twr, uint32_t tile, PassArgs &a) {
    const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
    const size_t n = (size_t)1 << k.log_n;
    a.in = k.in + (size_t)twr * n;
    a.out = k.out + (size_t)twr * n;
    a.tw = k.tw + (size_t)prime * n;
    a.twd = nullptr;
    a.q = k.mod[prime].value;
    a.tile = tile;
    a.rho0 = k.t1;
    a.stride = k.t2;
    if (!FWD && FOLD) {
        a.ninv = k.ninv[prime];
        a.w1ninv = k.w1ninv[prime];
    }
    a.pro_reduce = false;
    a.pro_ratio1 = 0;
    a.fp = false;
    if (k.fpinfo) {  // primes below 2^50 take the FP64 butterflies (uniform per workgroup)
        const FpInfo fi = k.fpinfo[prime];
        if (fi.ok) {
            a.fp = true;
            a.fpm = FpMod{fi.q, fi.qinv, (fi.ok & 2) != 0, (fi.ok & 4) != 0};
            a.twd = k.twf + (size_t)prime * n;
            if (!FWD && FOLD) {
                a.ninv = k.ninvf[prime];
                a.w1ninv = k.w1ninvf[prime];
            }
        }
    }
    if (EPI == EPI_INV_SCALE || EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE) {
        a.scale.x = k.scale[twr];
        a.scale.y = k.scale_shoup[twr];
    }
    a.aux = (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE) ? k.aux + (size_t)twr * n : nullptr;
    if (EPI == EPI_FWD_KSRESCALE) {
        a.scale2.x = k.scale2[twr];
        a.scale2.y = k.scale2_shoup[twr];
        a.aux2 = k.aux2 + (size_t)twr * n;
    }
}

#if defined(PHA_PASS_OCC)
#define PHA_PASS_ATTR __attribute__((amdgpu_waves_per_eu(PHA_PASS_OCC, PHA_PASS_OCC)))
#else
#define PHA_PASS_ATTR
#endif
#if defined(PHA_EXP_STAMPS)   // timing experiment: cycle stamps of workgroup (0,0), wave 0
#define PHA_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PHA_STAMP(i) do { } while (0)
#endif

// polynomial z skips its own digit: [excl_start + z*step, min(that + len, limit))  (ntt_modup.cu:422)
__device__ __forceinline__ bool limb_excluded(const NttKArgs &k, uint32_t twr, uint32_t z) {
    const uint32_t zd = k.excl_mod ? z % k.excl_mod : z;
    const uint32_t es = k.sel.excl_start + zd * k.excl_step;
    uint32_t ee = es + (k.sel.excl_end - k.sel.excl_start);
    ee = ee < k.excl_limit ? ee : k.excl_limit;
    return twr >= es && twr < ee;
}

// Arguments of tile `tile` of limb `twr` of polynomial `z`, with the batch offsets and the rescale prologue.
template <class C, bool FWD, int EPI, bool FOLD>
__device__ __forceinline__ void full_tile_args(const NttKArgs &k, uint32_t twr, uint32_t z, uint32_t tile, PassArgs &a) {
    tile_args<FWD, EPI, FOLD>(k, twr, tile, a);
    if (k.batch > 1) {  // same limbs of several polynomials in one launch
        a.in += (size_t)z * k.in_stride;
        a.out += (size_t)z * k.out_stride;
        if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE) a.aux += (size_t)z * k.aux_stride;
        if (EPI == EPI_FWD_KSRESCALE) a.aux2 += (size_t)z * k.aux2_stride;
    }
    if (FWD && (C::STRIDED || C::WHOLE) && k.pro_src) {  // rescale prologue: transform (the last limb of polynomial z) mod this prime
        const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
        a.in = k.pro_src + (size_t)z * k.pro_stride;
        a.pro_reduce = true;
        a.pro_ratio1 = k.mod[prime].ratio1;
    }
}

// One pass over one tile.  The whole pass is emitted twice, once with a.fp known true and once known false (a.fp is
// uniform per workgroup): each copy is then scheduled and register-allocated like a kernel that has only that butterfly
// back end (sweep at 2^16, 60 / 240 / 1020 limbs: 36.3 / 118 / 505 us with one shared body, 32.8 / 107 / 458 us specialised).
// COH: the pass reads what other workgroups of this launch wrote (one-launch transform).
// ONLY: 0 = both back ends (a.fp decides), 2 = the caller knows the limb runs on the integer back end (one body: fewer registers).
template <class C, bool FWD, int EPI, bool FOLD, int HOIST, bool COH, int ONLY = 0>
__device__ __forceinline__ void exec_pass(const PassArgs &a, u64 *lds, int tid) {
    u64 reg[C::EPT];
    u64x2 twreg[C::TW_TOTAL];
    using Prog = PassProgram<C, FWD, EPI, FOLD, HOIST, COH>;
    auto pass = [&](const PassArgs &pa) __attribute__((always_inline)) {
        Prog::load_twiddles(pa, tid, twreg);
        Prog::template run<0>(pa, lds, tid, reg, twreg);
        PHA_STAMP(1);
        tile_sync<C>();
        PHA_STAMP(2);
        Prog::template run<1>(pa, lds, tid, reg, twreg);
        PHA_STAMP(3);
        if constexpr (Prog::NSEG >= 3) {
            tile_sync<C>();
            PHA_STAMP(4);
            Prog::template run<2>(pa, lds, tid, reg, twreg);
            PHA_STAMP(5);
        }
        if constexpr (Prog::NSEG == 4) {
            tile_sync<C>();
            Prog::template run<3>(pa, lds, tid, reg, twreg);
        }
    };
    if (ONLY != 2 && a.fp) {
        PassArgs b = a;
        b.fp = true;
        pass(b);
    } else {
        PassArgs b = a;
        b.fp = false;
        pass(b);
    }
}

// ---- r04: the contiguous pass of BATCHED forward launches with the twiddles resident in registers ------------------------------
// A tile of the contiguous pass (whole rows of limb j) needs the same twiddles for every polynomial of the batch.  ntt_pass_kernel
// fetches them once per (tile, polynomial) -- 26 loads per thread with their address arithmetic, from the L2 when the
// polynomial-fastest block order has kept the rows there.  Here one workgroup owns tile t of limb j for `zper` polynomials in a
// row: on the FP64 back end it requests ALL rounds' twiddles once (8 bytes per entry: two registers each) and then walks its
// polynomials -- 26 loads, their address arithmetic and their waits leave the loop, which is what an issue-bound pass is short of
// (720 limbs at N = 2^16: 300 -> 274 us per step with 4 polynomials per workgroup).  Integer-back-end limbs (60-bit primes: 16-byte twiddle pairs would need 100+ registers) take the plain pass, one
// (tile, polynomial) per workgroup, at the head of the same grid.
// Work map of one launch: a 1-D grid, the integer-back-end limbs FIRST (their tiles are the longest: one polynomial per workgroup, so that
// they run side by side from the start), then the FP64 limbs with zper polynomials per workgroup.
struct ZloopMap {
    uint32_t n_int, n_fp;      // limbs of each kind in the selection
    uint32_t zper;             // polynomials per FP64 workgroup
    uint32_t tiles;            // tiles per limb
    uint32_t int_blocks;       // n_int * tiles * batch
    uint8_t limb[128];         // selection-relative limb indices: the n_int integer limbs, then the n_fp FP64 limbs
};
template <class C, bool FWD, int EPI, bool FOLD>
__global__ __launch_bounds__(C::THREADS) void ntt_zloop_kernel(const NttKArgs k, const ZloopMap m) {
    static_assert(!C::WHOLE, "the batched form exists for the passes of the two-pass plans");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    uint32_t b = blockIdx.x;
    PassArgs a;
    if (b < m.int_blocks) {   // (uniform) integer back end: the plain pass on one (tile, limb, polynomial)
        const uint32_t tile = b % m.tiles, rest = b / m.tiles, z = rest % k.batch, twr = k.sel.start + m.limb[rest / k.batch];
        if (limb_excluded(k, twr, z)) return;    // (uniform) the mod-up's rule: digit z does not transform its own limbs
        full_tile_args<C, FWD, EPI, FOLD>(k, twr, z, tile, a);
        exec_pass<C, FWD, EPI, FOLD, 0, false, 2>(a, lds, tid);   // (integer body only: the map put no FP64 limb here)
        return;
    }
    b -= m.int_blocks;
    const uint32_t tile = b % m.tiles, rest = b / m.tiles, zgroups = (k.batch + m.zper - 1) / m.zper;
    const uint32_t twr = k.sel.start + m.limb[m.n_int + rest / zgroups];
    const uint32_t z0 = (rest % zgroups) * m.zper, z1 = (z0 + m.zper < k.batch) ? z0 + m.zper : k.batch;
    full_tile_args<C, FWD, EPI, FOLD>(k, twr, z0, tile, a);
    using Prog = PassProgram<C, FWD, EPI, FOLD, 1, false>;   // HOIST 1: every round's twiddles before the loop
    a.fp = true;
    u64x2 twreg[C::TW_TOTAL];
    Prog::load_twiddles(a, tid, twreg);
    // Polynomial z + 1's coefficients are requested before polynomial z is transformed (a second register set: 126 registers, still
    // four wavefronts per SIMD): with the twiddles out of the loop those 8 loads are the only ones inside it, so no wait for anything
    // else drains them early.  Forward step +-0, inverse step 285 -> 272 us (its contiguous pass is the transform's first).
    u64 regA[C::EPT], regB[C::EPT];
    auto args_of = [&](uint32_t z) __attribute__((always_inline)) {
        PassArgs b = a;
        const size_t dz = (size_t)(z - z0);
        b.in += dz * k.in_stride;
        b.out += dz * k.out_stride;
        if (EPI == EPI_FWD_MODDOWN || EPI == EPI_FWD_MODDOWN_ADD || EPI == EPI_FWD_KSRESCALE) b.aux += dz * k.aux_stride;
        if (EPI == EPI_FWD_KSRESCALE) b.aux2 += dz * k.aux2_stride;
        return b;
    };
    auto transform = [&](const PassArgs &b, u64 *reg) __attribute__((always_inline)) {
        Prog::template run_prefetched<0>(b, lds, tid, reg, twreg);
        tile_sync<C>();
        Prog::template run_prefetched<1>(b, lds, tid, reg, twreg);
        if constexpr (Prog::NSEG >= 3) {
            tile_sync<C>();
            Prog::template run_prefetched<2>(b, lds, tid, reg, twreg);
        }
        if constexpr (Prog::NSEG == 4) {
            tile_sync<C>();
            Prog::template run_prefetched<3>(b, lds, tid, reg, twreg);
        }
        tile_sync<C>();   // the next polynomial's first round writes the same LDS words
    };
    auto next = [&](uint32_t z) __attribute__((always_inline)) {   // (uniform) first polynomial >= z that transforms this limb (mod-up: a digit skips its own)
        while (z < z1 && limb_excluded(k, twr, z)) z++;
        return z;
    };
    uint32_t z = next(z0);
    if (z >= z1) return;
    Prog::prefetch(args_of(z), tid, regA);
    __builtin_amdgcn_s_setprio(0);
    for (;;) {
        uint32_t zn = next(z + 1);
        if (zn < z1) Prog::prefetch(args_of(zn), tid, regB);
        transform(args_of(z), regA);
        if (zn >= z1) break;
        z = zn;
        zn = next(z + 1);
        if (zn < z1) Prog::prefetch(args_of(zn), tid, regA);
        transform(args_of(z), regB);
        if (zn >= z1) break;
        z = zn;
    }
}

// r04 experiment hook: minimum wavefronts per SIMD asked of the compiler for the strided / contiguous pass (register budget 512 / n)
#if defined(PHA_X_OCC_S) || defined(PHA_X_OCC_C)
#ifndef PHA_X_OCC_S
#define PHA_X_OCC_S 2
#endif
#ifndef PHA_X_OCC_C
#define PHA_X_OCC_C 1
#endif
template <class C> constexpr int x_occ() { return C::WHOLE ? 1 : C::STRIDED ? (C::THREADS >= 512 ? PHA_X_OCC_S : 1) : (C::THREADS == 64 ? PHA_X_OCC_C : 1); }
#define PHA_PASS_BOUNDS __launch_bounds__(C::THREADS, x_occ<C>())
#else
#define PHA_PASS_BOUNDS __launch_bounds__(C::THREADS)
#endif
template <class C, bool FWD, int EPI, bool FOLD, int HOIST>
__global__ PHA_PASS_BOUNDS PHA_PASS_ATTR void ntt_pass_kernel(const NttKArgs k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);

    __builtin_amdgcn_s_setprio(3);   // until the first round's global loads are issued (PassProgram::run)
    uint32_t tile = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (k.zfast_tiles) {
        // block b = ((chunk * batch) + z) * run + i: the `run` adjacent (tile, limb) groups of a chunk for polynomial 0, then the same
        // groups for polynomial 1, ...; run is a multiple of 8, so group i of every polynomial lands on XCD i % 8
        const uint32_t b = blockIdx.x, per_chunk = k.zfast_run * k.batch, chunk = b / per_chunk, rem = b - chunk * per_chunk;
        z = rem / k.zfast_run;
        const uint32_t group = chunk * k.zfast_run + (rem - z * k.zfast_run);
        if (group >= k.zfast_tiles * k.sel.count) return;
        tile = group % k.zfast_tiles;
        y = group / k.zfast_tiles;
    }
    const uint32_t twr = k.sel.start + y;  // limb in the buffer (uniform)
    if (limb_excluded(k, twr, z)) return;
    PassArgs a;
    full_tile_args<C, FWD, EPI, FOLD>(k, twr, z, tile, a);
#if defined(PHA_EXP_STAMPS)
    const unsigned long long wg_t0 = wall_clock64();
#endif
    PHA_STAMP(0);
    exec_pass<C, FWD, EPI, FOLD, HOIST, false>(a, lds, threadIdx.x);
#if defined(PHA_EXP_STAMPS)
    __builtin_amdgcn_s_waitcnt(0);
    PHA_STAMP(6);
    if (threadIdx.x == 0) {  // wall-clock (100 MHz) start/end of every workgroup of the last launch
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
        if (id < 1024) { g_wg_times[2 * id] = wg_t0; g_wg_times[2 * id + 1] = wall_clock64(); }
    }
#endif
}

#if defined(PHA_EXPERIMENTS)
// ---- both passes in ONE launch, the intermediate handed over through the XCD's own L2 -------------------------------
// (r02; profiles/HISTORY.md has the memory-side experiment behind it: the two access patterns at 720 limbs take
// 286 us as two launches and 211 us in this form, because the intermediate never crosses the fabric a second time.)
// Placement: the workgroups of a 1-D grid are dealt to the 8 XCDs round-robin, so all workgroups with the same
// b % 8 (a "class") share one XCD -- XCD (b + r) % 8 with r = 0 for plain launches and some other constant under
// hipGraph replay.  Only the sharing matters here and it is checked in every launch: the first workgroup of a class
// records its XCC_ID, every other one compares and traps on a difference (the context also runs a census launch and
// keeps the two-launch form if the rule does not hold on the device).  A "unit" is one limb of one polynomial;
// unit u belongs to class u % 8, whose workgroups visit its units in order: workgroup (slot s, tile t) of a class first
// runs the transform's first pass on tile t of the unit of slot s, then the second pass on tile t of the unit of slot
// s - lag, whose tiles were all started lag * tiles_per_unit workgroups earlier in this class and have normally been
// written by then.  Hand-off protocol (every participant of a unit is on one XCD, whose L2 is the only cache level they
// share):
//   producer: plain stores (written through, the line stays in this XCD's L2), s_waitcnt vmcnt(0), workgroup barrier,
//             one non-returning atomic add on the unit's counter (executes in this L2);
//   consumer: one lane polls the counter with a returning atomic OR 0 (never answered by the CU's L1) until all tiles
//             have arrived, workgroup barrier, then reads the intermediate with agent-scope loads (`sc1`: miss the L1,
//             answered by the L2) -- PassProgram's COH flag.
// The counter array cleans itself: every consumer counts itself in right after its poll has succeeded, and the one that
// finds all the others counted puts the unit's two words back to zero (nobody polls them any more); that workgroup also
// counts the finished unit at agent scope, and the one that finishes the launch's last unit clears the 8 class words
// (every workgroup that has work looked at its class word before its own unit could finish).  So a launch never
// allocates or clears anything from the host, which keeps it capturable.
// Dispatch order is ascending block index, so a waiting workgroup only ever waits for workgroups that are already
// resident or finished (the same assumption every decoupled look-back scan makes); the poll is bounded and traps.
#ifndef PHA_FUSED_MIN_WAVES
#define PHA_FUSED_MIN_WAVES 6   // three 512-thread workgroups per CU
#endif
#if defined(PHA_FUSED_DEBUG)
__device__ uint32_t g_fused_dbg[8];
#endif
struct FusedArgs {
    uint32_t *cls;         // [8] XCC_ID + 1 of each workgroup class (0 = not recorded yet); [8] = units finished
    uint32_t *flags;       // [units][2]: tiles of the unit's first pass that have been written, consumers that have seen that
    uint32_t units, slots, tpl, lag, count;   // count = limbs per polynomial (unit = z * count + y)
    uint32_t active_units; // units that are not excluded
    uint32_t split;        // 1: a workgroup runs ONE pass (even positions of a class: first pass, odd: second pass of the lagged unit)
};
__device__ __forceinline__ void l2_arrive(uint32_t *p) {
    asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(1u) : "memory");
}
__device__ __forceinline__ uint32_t l2_fetch_or(uint32_t *p, uint32_t v) {
    uint32_t r;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t l2_fetch_add(uint32_t *p, uint32_t v) {
    uint32_t r;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ void l2_store(uint32_t *p, uint32_t v) {
    asm volatile("global_atomic_swap %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// run configuration C on 4096-coefficient tile `tile` with the 512 threads of the workgroup: a 512-thread configuration
// directly, a one-wavefront configuration (512-coefficient tiles) as eight independent wavefronts
template <class C, bool FWD, int EPI, bool FOLD, bool COH>
__device__ __forceinline__ void fused_run_tile(const NttKArgs &k, uint32_t twr, uint32_t z, uint32_t tile, u64 *lds) {
    static_assert(C::THREADS == 512 || C::THREADS == 64, "the one-launch transform runs 512-thread workgroups");
    PassArgs a;
    if constexpr (C::THREADS == 512) {
        full_tile_args<C, FWD, EPI, FOLD>(k, twr, z, tile, a);
        exec_pass<C, FWD, EPI, FOLD, 0, COH>(a, lds, threadIdx.x);
    } else {
        const uint32_t wave = threadIdx.x >> 6;
        full_tile_args<C, FWD, EPI, FOLD>(k, twr, z, tile * 8 + wave, a);
        exec_pass<C, FWD, EPI, FOLD, 0, COH>(a, lds + (size_t)wave * C::LDS_WORDS, threadIdx.x & 63);
    }
}

template <class PS, class PC, bool FWD, int EPI, bool FOLD>   // PS: strided pass (512 threads), PC: contiguous pass
__global__ __launch_bounds__(512, PHA_FUSED_MIN_WAVES) void ntt_fused_kernel(const NttKArgs kA, const NttKArgs kB, const FusedArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);
    const uint32_t b = blockIdx.x, xcd = b & 7u, within = b >> 3;
    // split form: the positions of a class alternate between first-pass and second-pass workgroups, so that a workgroup
    // lives for one pass only and the second pass's wavefronts are not tied to a first pass's barrier schedule
    const uint32_t role = f.split ? (within & 1u) : 2u, idx = f.split ? (within >> 1) : within;
    const uint32_t tile = idx % f.tpl, slot = idx / f.tpl;
    const uint32_t uA = slot * 8 + xcd;
    const bool has_a = role != 1u && slot < f.slots && uA < f.units && !limb_excluded(kA, kA.sel.start + uA % f.count, uA / f.count);
    const uint32_t uB = (slot - f.lag) * 8 + xcd;   // (wraps when slot < lag: has_b is false then)
    const bool has_b = role != 0u && slot >= f.lag && uB < f.units && !limb_excluded(kB, kB.sel.start + uB % f.count, uB / f.count);
    if (!has_a && !has_b) return;   // (and never touches the class words: see the clean-up rule above)
    uint32_t cls_seen = 0, cls_mine = 0;
    if (threadIdx.x == 0) {   // class check, part 1 (the answer is looked at after the first pass)
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        cls_mine = (id & 7u) + 1;
        uint32_t expected = 0;
        __hip_atomic_compare_exchange_strong(f.cls + xcd, &expected, cls_mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cls_seen = expected;
    }
    if (has_a) {
        const uint32_t z = uA / f.count, twr = kA.sel.start + uA % f.count;
        if (FWD) fused_run_tile<PS, true, EPI_NONE, false, false>(kA, twr, z, tile, lds);
        else fused_run_tile<PC, false, EPI_NONE, false, false>(kA, twr, z, tile, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's stores have reached the L2
        __syncthreads();
        if (threadIdx.x == 0) l2_arrive(f.flags + 2 * (size_t)uA);
    }
    if (threadIdx.x == 0 && cls_seen != 0 && cls_seen != cls_mine) {   // class check, part 2
#if defined(PHA_FUSED_DEBUG)
        if (atomicAdd(&g_fused_dbg[0], 1u) == 0) { g_fused_dbg[1] = b; g_fused_dbg[2] = cls_mine; g_fused_dbg[3] = cls_seen; }
#else
        __builtin_trap();   // two workgroups of one class on different XCDs: the hand-off below would not be coherent
#endif
    }
    if (!has_b) return;
    const uint32_t z = uB / f.count, twr = kB.sel.start + uB % f.count;
    uint32_t *flag = f.flags + 2 * (size_t)uB;
    // wait until every tile of the unit's first pass has been written.  One lane polls for the workgroup -- or, when the
    // second pass is the barrier-free contiguous one, one lane per wavefront, so that no wavefront waits for another's poll
    constexpr bool kWaveConsumer = FWD && PC::THREADS == 64;
    const bool poller = kWaveConsumer && f.split ? (threadIdx.x & 63) == 0 : threadIdx.x == 0;
    if (poller) {
        uint32_t spins = 0;
        while (l2_fetch_or(flag, 0u) < f.tpl) {
            __builtin_amdgcn_s_sleep(8);
#if defined(PHA_FUSED_DEBUG)
            if (++spins > (1u << 16)) { atomicAdd(&g_fused_dbg[4], 1u); break; }
#else
            if (++spins > (1u << 24)) __builtin_trap();   // never hang the device on a broken assumption
#endif
        }
    }
    if (!(kWaveConsumer && f.split)) __syncthreads();   // also: pass A no longer uses the LDS
    if (FWD) fused_run_tile<PC, true, EPI, false, true>(kB, twr, z, tile, lds);
    else fused_run_tile<PS, false, EPI, FOLD, true>(kB, twr, z, tile, lds);
    // every wavefront of this workgroup has seen the unit complete: count the workgroup in; the last one of the unit puts
    // the unit's words back to zero (nobody polls them any more)
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == f.tpl - 1) {
        l2_store(flag, 0u);
        l2_store(flag + 1, 0u);
        if (__hip_atomic_fetch_add(f.cls + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f.active_units - 1) {
            for (int i = 0; i < 9; i++) __hip_atomic_store(f.cls + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#endif  // PHA_EXPERIMENTS

template <class C, bool FWD, int EPI, bool FOLD>
static void launch_pass(const NttKArgs &k, hipStream_t s) {
    const size_t n = (size_t)1 << k.log_n;
    const size_t lds_bytes = (size_t)C::LDS_WORDS * sizeof(u64);
    const unsigned tiles_per_limb = (unsigned)(n >> C::LOGTILE);
    // (a persistent software-pipelined form was measured and dropped: DESIGN.md section 7)
    dim3 grid(tiles_per_limb, k.sel.count, k.batch);
    NttKArgs kk = k;
    kk.zfast_tiles = 0;   // (k.zfast_tiles is only the caller's request flag)
    if (!C::STRIDED && !C::WHOLE && k.zfast_tiles) {
        kk.zfast_tiles = tiles_per_limb;
        kk.zfast_run = 8;
#if defined(PHA_X_KNOBS)   // r04 experiment: longer runs of adjacent tiles per polynomial (env, read once; a multiple of 8)
        static const unsigned run_x = std::getenv("PHA_X_ZRUN") ? (unsigned)std::atol(std::getenv("PHA_X_ZRUN")) : 8u;
        kk.zfast_run = run_x;
#endif
        const unsigned groups = tiles_per_limb * k.sel.count;
        grid = dim3(((groups + kk.zfast_run - 1) / kk.zfast_run) * kk.zfast_run * k.batch, 1, 1);
    }
    // (requesting all rounds' twiddles up front, HOIST 1, was measured again in r02 for the small launches of mod-down and
    //  rescale: no gain at any size)
    if (lds_bytes > 64 * 1024) {   // beyond the default dynamic LDS limit: raise it once per kernel and device
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        PHA_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            PHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<C, FWD, EPI, FOLD, (C::WHOLE ? 0 : C::STRIDED ? PHA_X_HOIST_S : PHA_X_HOIST_C)>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    // (r03: a two-tiles-per-wavefront, software-pipelined form of the one-wavefront contiguous pass for small launches was
    //  measured and dropped -- half as many wavefronts with twice the work each lose more latency hiding than the overlap of
    //  one tile's stores with the next tile's butterflies gains: 45 limbs 17.3 -> 19.2 us, 32 limbs 15.0 -> 16.5 us)
    constexpr int HX = C::WHOLE ? 0 : C::STRIDED ? PHA_X_HOIST_S : PHA_X_HOIST_C;
#if defined(PHA_X_KNOBS)   // r04 experiment: cap the resident workgroups of a pass by padding its LDS request (bytes; env, read once)
    static const size_t pad_s = std::getenv("PHA_X_LDS_S") ? (size_t)std::atol(std::getenv("PHA_X_LDS_S")) : 0;
    static const size_t pad_c = std::getenv("PHA_X_LDS_C") ? (size_t)std::atol(std::getenv("PHA_X_LDS_C")) : 0;
    const size_t lds_x = lds_bytes + (C::STRIDED ? pad_s : pad_c);
    hipLaunchKernelGGL((ntt_pass_kernel<C, FWD, EPI, FOLD, HX>), grid, dim3(C::THREADS), lds_x, s, kk);
#else
    hipLaunchKernelGGL((ntt_pass_kernel<C, FWD, EPI, FOLD, HX>), grid, dim3(C::THREADS), lds_bytes, s, kk);
#endif
    check_launch();
}

#if defined(PHA_EXPERIMENTS)
template <class PS, class PC, bool FWD, int EPI, bool FOLD>
static bool launch_fused(Context &c, const NttKArgs &kA_in, const NttKArgs &kB_in, hipStream_t s) {
    static_assert(PS::THREADS == 512 && PS::LOGTILE == 12, "strided pass: one 4096-coefficient tile per 512-thread workgroup");
    NttKArgs kA = kA_in, kB = kB_in;
    const size_t n = (size_t)1 << kA.log_n;
    kA.zfast_tiles = kB.zfast_tiles = 0;   // (the request flag of the two-launch form means nothing here)
    FusedArgs f{};
    f.count = kA.sel.count;
    f.units = kA.sel.count * kA.batch;
    f.slots = (f.units + 7) / 8;
    f.tpl = (uint32_t)(n >> 12);
    f.lag = (uint32_t)g_fused_lag.load(std::memory_order_relaxed);
    f.cls = c.ntt_flags(s, f.units);
    if (!f.cls) return false;   // no counter arena for this stream: the caller takes the two launches
    f.flags = f.cls + 16;
    f.active_units = 0;
    for (uint32_t z = 0; z < kA.batch; z++) {   // the same rule as limb_excluded()
        const uint32_t zd = kA.excl_mod ? z % kA.excl_mod : z;
        const uint32_t es = kA.sel.excl_start + zd * kA.excl_step;
        uint32_t ee = es + (kA.sel.excl_end - kA.sel.excl_start);
        ee = ee < kA.excl_limit ? ee : kA.excl_limit;
        const uint32_t lo = es > kA.sel.start ? es : kA.sel.start;
        const uint32_t hi = ee < kA.sel.start + kA.sel.count ? ee : kA.sel.start + kA.sel.count;
        f.active_units += kA.sel.count - (hi > lo ? hi - lo : 0);
    }
    if (f.active_units == 0) return true;
    const size_t lds_a = (size_t)PS::LDS_WORDS * sizeof(u64);
    const size_t lds_b = (size_t)PC::LDS_WORDS * sizeof(u64) * (PC::THREADS == 64 ? 8 : 1);
    const size_t lds_bytes = lds_a > lds_b ? lds_a : lds_b;
    f.split = g_fused_split.load(std::memory_order_relaxed) ? 1u : 0u;
    const unsigned blocks = (f.slots + f.lag) * f.tpl * 8 * (f.split ? 2u : 1u);
    hipLaunchKernelGGL((ntt_fused_kernel<PS, PC, FWD, EPI, FOLD>), dim3(blocks), dim3(512), lds_bytes, s, kA, kB, f);
    check_launch();
    return true;
}

#endif  // PHA_EXPERIMENTS

// N = 4096 / 8192 as ONE pass (the transform fits a tile): T1 = 1, T2 = N
template <class W>
static void forward_whole(NttKArgs k, int epi, hipStream_t s) {
    k.t1 = 1;
    k.t2 = W::T;
    k.mid = k.out;
    if (epi == EPI_FWD_MODDOWN) launch_pass<W, true, EPI_FWD_MODDOWN, false>(k, s);
    else if (epi == EPI_FWD_MODDOWN_ADD) launch_pass<W, true, EPI_FWD_MODDOWN_ADD, false>(k, s);
    else if (epi == EPI_FWD_KSRESCALE) launch_pass<W, true, EPI_FWD_KSRESCALE, false>(k, s);
    else launch_pass<W, true, EPI_FWD_CANON, false>(k, s);
}
template <class W>
static void inverse_whole(NttKArgs k, int epi, hipStream_t s) {
    k.t1 = 1;
    k.t2 = W::T;
    k.mid = k.out;
    if (epi == EPI_INV_SCALE) launch_pass<W, false, EPI_INV_SCALE, true>(k, s);
    else launch_pass<W, false, EPI_INV_CANON, true>(k, s);
}

// the batched, twiddle-resident passes (ntt_zloop_kernel): plain launches of >= 8 polynomials that fill the device
// several times over (the headline step, the batched key switch's 2 B-polynomial transforms); zper polynomials per workgroup, chosen so
// that the launch still holds >= 3 generations of wavefronts
#ifndef PHA_ZLOOP_MIN_BATCH
#define PHA_ZLOOP_MIN_BATCH 8
#endif
template <class C, bool FWD, int EPI, bool FOLD>
static bool launch_zloop(const NttKArgs &k, hipStream_t s) {
    if (k.batch < (uint32_t)PHA_ZLOOP_MIN_BATCH || k.pro_src) return false;
    const size_t n = (size_t)1 << k.log_n;
    const unsigned tiles_per_limb = (unsigned)(n >> C::LOGTILE);
    const size_t waves_per_poly = (size_t)tiles_per_limb * k.sel.count * (C::THREADS / 64);
    // resident wavefront slots at this kernel's register budget (4 per SIMD) x 3 generations
    const size_t want = (size_t)256 * 4 * 4 * 3;
    // polynomials per workgroup: the gain saturates at 4-8 (r04 sweep: 4 / 8 / 16 per workgroup -> 274 / 275 / 290 us per 720-limb step),
    // so at most 8, split evenly over the workgroups of a tile, fewer while the launch would not fill the device three times
    uint32_t zper = (k.batch + ((k.batch + 7) / 8) - 1) / ((k.batch + 7) / 8);
    while (zper > 4 && waves_per_poly * ((k.batch + zper - 1) / zper) < want) zper = (zper + 1) / 2;
    if (waves_per_poly * ((k.batch + zper - 1) / zper) < want / 3) return false;   // too small a launch: the plain pass
#if defined(PHA_X_KNOBS)
    static const unsigned zper_x = std::getenv("PHA_X_ZPER") ? (unsigned)std::atol(std::getenv("PHA_X_ZPER")) : 0u;
    if (zper_x) zper = zper_x;
#endif
    // which limbs of the selection run on the FP64 back end (the others: runs of consecutive limbs through the plain kernel)
    auto is_fp = [&](uint32_t y) {
        const uint32_t twr = k.sel.start + y, prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
        return k.fpinfo != nullptr && (k.h_primes[prime] >> 50) == 0;
    };
    if (k.sel.count > 128) return false;
    ZloopMap m{};
    m.zper = zper;
    m.tiles = tiles_per_limb;
    for (uint32_t y = 0; y < k.sel.count; y++)
        if (!is_fp(y)) m.limb[m.n_int++] = (uint8_t)y;
    for (uint32_t y = 0; y < k.sel.count; y++)
        if (is_fp(y)) m.limb[m.n_int + m.n_fp++] = (uint8_t)y;
    if (m.n_fp * 2 < k.sel.count) return false;   // mostly wide primes: the plain pass for everything
    m.int_blocks = m.n_int * tiles_per_limb * k.batch;
    NttKArgs kk = k;
    kk.zfast_tiles = 0;
    const unsigned blocks = m.int_blocks + m.n_fp * tiles_per_limb * ((k.batch + zper - 1) / zper);
    const size_t lds_bytes = (size_t)C::LDS_WORDS * sizeof(u64);
    hipLaunchKernelGGL((ntt_zloop_kernel<C, FWD, EPI, FOLD>), dim3(blocks), dim3(C::THREADS), lds_bytes, s, kk, m);
    check_launch();
    return true;
}

#ifndef PHA_IP_PLAN
#define PHA_IP_PLAN 3    // NttPlan variant whose contiguous pass carries the inner product (3: 8 coefficients per thread; 5: 4)
#endif
template <int LOGN, int VARIANT>
static void forward_impl(NttKArgs k, int epi, hipStream_t s, Context *fused = nullptr) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    // the only producer that folds the strided pass away (modup_conv_s1_kernel) writes the 64 x 1024 split of NttPlan<16, 10>: a
    // plan choice that disagrees with it would transform garbage silently (the inverse's second_pass_only has the same guard)
    if (k.first_pass_done && !(LOGN == 16 && VARIANT == 10 && P1::LOGT == 6))
        throw std::logic_error("ntt_forward: first_pass_done needs the 64 x 1024 plan the fused conversion wrote");
    u64 *const final_out = k.out;
    const size_t final_stride = k.out_stride;
    k.out = k.mid;
    k.out_stride = k.poly_stride;
    const NttKArgs k1 = k;
    k.in = k.mid;
    k.in_stride = k.poly_stride;
    k.out = final_out;
    k.out_stride = final_stride;
    k.pro_src = nullptr;   // the rescale prologue belongs to the first pass
#if defined(PHA_EXPERIMENTS)
    if constexpr (P1::THREADS == 512 && P1::LOGTILE == 12 && P2::THREADS == 64) {
        if (fused && epi != EPI_FWD_KSRESCALE && !k.first_pass_done) {   // both passes in one launch
            const bool done = epi == EPI_FWD_MODDOWN ? launch_fused<P1, P2, true, EPI_FWD_MODDOWN, false>(*fused, k1, k, s)
                              : epi == EPI_FWD_MODDOWN_ADD ? launch_fused<P1, P2, true, EPI_FWD_MODDOWN_ADD, false>(*fused, k1, k, s)
                                                           : launch_fused<P1, P2, true, EPI_FWD_CANON, false>(*fused, k1, k, s);
            if (done) return;
        }
    }
#else
    (void)fused;
#endif
    // (the strided pass in the batched form measured SLOWER -- 720 limbs 282 -> 302 us: its twiddles are few and shared by a tile's columns,
    //  and a 512-thread workgroup that walks several polynomials keeps its barrier schedule for all of them)
    if (!k.first_pass_done) launch_pass<P1, true, EPI_NONE, false>(k1, s);
    if (k.first_pass_only) return;
    // pass 1 ran in -> mid with the input stride; pass 2 reads mid and writes out with the output stride
#if !defined(PHA_NO_ZLOOP)
    if constexpr (VARIANT == 10 || VARIANT == 3 || VARIANT == 4) {   // the product's plans: batched launches take the twiddle-resident contiguous pass
        using Z2 = typename NttPlan<LOGN, VARIANT == 4 ? 3 : VARIANT>::P2;   // (with the twiddles out of the loop there is nothing to form on the fly)
        const bool done = epi == EPI_FWD_MODDOWN ? launch_zloop<Z2, true, EPI_FWD_MODDOWN, false>(k, s)
                          : epi == EPI_FWD_MODDOWN_ADD ? launch_zloop<Z2, true, EPI_FWD_MODDOWN_ADD, false>(k, s)
                          : epi == EPI_FWD_KSRESCALE ? launch_zloop<Z2, true, EPI_FWD_KSRESCALE, false>(k, s)
                                                     : launch_zloop<Z2, true, EPI_FWD_CANON, false>(k, s);
        if (done) return;
    }
#endif
    if (epi == EPI_FWD_MODDOWN) launch_pass<P2, true, EPI_FWD_MODDOWN, false>(k, s);
    else if (epi == EPI_FWD_MODDOWN_ADD) launch_pass<P2, true, EPI_FWD_MODDOWN_ADD, false>(k, s);
    else if (epi == EPI_FWD_KSRESCALE) launch_pass<P2, true, EPI_FWD_KSRESCALE, false>(k, s);
    else launch_pass<P2, true, EPI_FWD_CANON, false>(k, s);
}

template <int LOGN, int VARIANT>
static void inverse_impl(NttKArgs k, int epi, hipStream_t s, Context *fused = nullptr) {
    using P1 = typename NttPlan<LOGN, VARIANT>::P1;
    using P2 = typename NttPlan<LOGN, VARIANT>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    u64 *const final_out = k.out;
    const size_t final_stride = k.out_stride;
    k.out = k.mid;
    k.out_stride = k.poly_stride;
    const NttKArgs k1 = k;
    k.in = k.mid;
    k.in_stride = k.poly_stride;
    k.out = final_out;
    k.out_stride = final_stride;
#if defined(PHA_EXPERIMENTS)
    if constexpr (P1::THREADS == 512 && P1::LOGTILE == 12 && P2::THREADS == 64) {
        if (fused) {
            const bool done = epi == EPI_INV_SCALE ? launch_fused<P1, P2, false, EPI_INV_SCALE, true>(*fused, k1, k, s)
                                                   : launch_fused<P1, P2, false, EPI_INV_CANON, true>(*fused, k1, k, s);
            if (done) return;
        }
    }
#else
    (void)fused;
#endif
    if (k.second_pass_only) {   // mid already holds the contiguous pass's output (T1 x T2 of this plan: choose_plan keeps the split of plan 3)
        // the producer (modup_ip_kernel) ran the contiguous pass of the fused mod-up's plan: this strided pass must complete THAT split
        using IpPlan = NttPlan<LOGN, (LOGN >= 14 && LOGN <= 16) ? PHA_IP_PLAN : 3>;
        if (P1::LOGT + IpPlan::P2::LOGT != LOGN)
            throw std::logic_error("second_pass_only: the chosen plan's strided pass does not complete the fused mod-up's contiguous pass");
        if (epi == EPI_INV_SCALE) launch_pass<P1, false, EPI_INV_SCALE, true>(k, s);
        else launch_pass<P1, false, EPI_INV_CANON, true>(k, s);
        return;
    }
#if !defined(PHA_NO_ZLOOP)
    if constexpr (VARIANT == 10 || VARIANT == 3 || VARIANT == 4) {   // batched launches: the contiguous pass with the twiddles resident (ntt_zloop_kernel)
        using Z2 = typename NttPlan<LOGN, VARIANT == 4 ? 3 : VARIANT>::P2;
        if (!launch_zloop<Z2, false, EPI_NONE, false>(k1, s)) launch_pass<P2, false, EPI_NONE, false>(k1, s);
        if (epi == EPI_INV_SCALE) launch_pass<P1, false, EPI_INV_SCALE, true>(k, s);
        else launch_pass<P1, false, EPI_INV_CANON, true>(k, s);
        return;
    }
#endif
    launch_pass<P2, false, EPI_NONE, false>(k1, s);
    if (epi == EPI_INV_SCALE) launch_pass<P1, false, EPI_INV_SCALE, true>(k, s);
    else launch_pass<P1, false, EPI_INV_CANON, true>(k, s);
}

static NttKArgs make_args(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, const NttExtra &x,
                          bool fwd) {
    NttKArgs k{};
    k.in = in;
    k.mid = mid;
    k.out = out;
    k.tw = fwd ? c.d_tw.p : c.d_itw.p;
    k.mod = c.d_mod.p;
    k.ninv = c.d_ninv.p;
    k.w1ninv = c.d_w1ninv.p;
    const bool use_fp = !(ntt_variant() & 8);
    k.twf = fwd ? c.d_twf.p : c.d_itwf.p;
    k.ninvf = c.d_ninvf.p;
    k.w1ninvf = c.d_w1ninvf.p;
    k.fpinfo = use_fp ? c.d_fpinfo.p : nullptr;
    k.scale = x.scale;
    k.scale_shoup = x.scale_shoup;
    k.aux = x.aux;
    k.aux2 = x.aux2;
    k.scale2 = x.scale2;
    k.scale2_shoup = x.scale2_shoup;
    k.aux2_stride = x.aux2_stride ? x.aux2_stride : x.poly_stride;
    k.sel = sel;
    k.log_n = c.log_n;
    k.h_primes = c.primes.data();
    k.batch = x.batch ? x.batch : 1;
    k.poly_stride = x.poly_stride;
    k.in_stride = x.in_stride ? x.in_stride : x.poly_stride;
    k.first_pass_only = fwd && x.first_pass_only;
    k.second_pass_only = !fwd && x.second_pass_only;
    k.first_pass_done = fwd && x.first_pass_done;
    k.out_stride = x.out_stride ? x.out_stride : x.poly_stride;
    k.aux_stride = x.aux_stride ? x.aux_stride : x.poly_stride;
    k.excl_step = x.excl_step;
    k.excl_limit = x.excl_limit;
    k.excl_mod = x.excl_mod;
    k.pro_src = fwd ? x.pro_src : nullptr;
    k.pro_stride = x.pro_stride;
    uint32_t excl = 0;
    if (sel.excl_end > sel.excl_start) {
        const uint32_t lo = sel.excl_start > sel.start ? sel.excl_start : sel.start;
        const uint32_t hi = sel.excl_end < sel.start + sel.count ? sel.excl_end : sel.start + sel.count;
        excl = hi > lo ? hi - lo : 0;
        if (excl && (lo != sel.excl_start || hi != sel.excl_end)) excl = 0xffffffffu;  // partial overlap: not pipelined
    }
    k.active = excl == 0xffffffffu ? 0 : sel.count - excl;
    return k;
}

static void check_sel(Context &c, const LimbSel &sel) {
    if (sel.count == 0) return;
    const uint32_t last = sel.start + sel.count - 1;
    const uint32_t prime_last = last >= sel.remap_from ? last + sel.remap_add : last;
    if (prime_last >= c.rows) throw std::invalid_argument("modulus index out of range of the NTT tables");
}


// ---- the key inner product as the epilogue of the mod-up's contiguous pass (r03) ------------------------------------------------
// keyswitch_inplace runs, per digit b, the forward NTT of the converted limbs and then key_switch_inner_prod over all digits
// (src/rns_bconv.cu:530-627, src/eval_key_switch.cu:14-92): the transformed digits (beta x (l + alpha) limbs) are written and read
// back once, and the digit's own limbs are copied verbatim first.  Here ONE wavefront owns a 512-coefficient tile of limb j for
// ALL digits: it runs the contiguous pass on digit 0's tile, multiplies the outputs -- still in registers -- by the two key words,
// does the same for digit 1, ... and stores only the two sums.  A digit's own limb is not transformed at all: its NTT-form value
// is the input c2 itself, read where it lies (so the conversion need not copy it).  Per key switch at C3 that removes 67.5 MiB of
// transformed-digit stores, 90 MiB of digit loads, the 45 MiB own-limb copy and one launch.
// Accumulation: limbs on the FP64 back end add centred residues as doubles (fp_mulmod of the lazy transform output with the key
// word as a double: |sum| <= beta (q/2 + 1)); integer limbs add Barrett-reduced products modulo q.  Both equal
// (sum_b x_b k_b) mod q, the value the 128-bit accumulate + Barrett of inner_prod_kernel stores.
template <class C, int BETA, bool FP>
__device__ __forceinline__ void modup_ip_body(const NttKArgs &k, const ModupIpArgs &ip, uint32_t twr, uint32_t prime, uint32_t tile,
                                              u64 *lds, int tid) {
    constexpr int RL = C::NR - 1, r = C::r(RL), K = 1 << r, G = C::EPT >> r;
    static_assert(C::LOGT - C::s0(RL) - r == 0, "the last round holds runs of K consecutive coefficients");
    using Prog = PassProgram<C, true, EPI_NONE, false, 0, false>;
    const size_t n = (size_t)1 << k.log_n;
    const DModulus m = k.mod[prime];
    const u64 q = m.value;
    FpMod fm{};
    u64 accb[C::EPT], acca[C::EPT];      // FP: doubles (bit patterns); integer: residues
#pragma unroll
    for (int i = 0; i < C::EPT; i++) accb[i] = acca[i] = FP ? as_u64(0.0) : 0;
    size_t g0[G];                        // first coefficient of each run inside the limb
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        int v, hi, lo;
        decode_group<C, RL>(tid + C::THREADS * gi, v, hi, lo);
        g0[gi] = ((size_t)tile * C::V + v) * C::T + ((size_t)hi << r);
    }
#if defined(PHA_IP_UNROLLED)   // (r03 A/B: the unrolled digit loop overlaps consecutive digits, needs 210 VGPRs = two wavefronts per
#pragma unroll                 //  SIMD, and is 3 % slower per key switch than the rolled one at 164 VGPRs = three)
#else
#pragma unroll 1
#endif
    for (int b = 0; b < BETA; b++) {
        PassArgs a;
        full_tile_args<C, true, EPI_NONE, false>(k, twr, (uint32_t)b, tile, a);
        a.fp = FP;
        fm = a.fpm;
        u64 reg[C::EPT];
        if (ip.own && limb_excluded(k, twr, (uint32_t)b)) {   // (uniform) digit b's own limb: the NTT-form input itself
#pragma unroll
            for (int gi = 0; gi < G; gi++)
#pragma unroll
                for (int kk = 0; kk < K; kk += 2) {
                    const u64x2 w = *reinterpret_cast<const u64x2 *>(ip.own + (size_t)twr * n + g0[gi] + kk);
                    reg[gi * K + kk] = FP ? as_u64(fp_from_canon(w.x)) : w.x;
                    reg[gi * K + kk + 1] = FP ? as_u64(fp_from_canon(w.y)) : w.y;
                }
        } else {
            u64x2 twreg[C::TW_TOTAL];
            Prog::load_twiddles(a, tid, twreg);
            Prog::template run<0>(a, lds, tid, reg, twreg);
            tile_sync<C>();
            if constexpr (Prog::NSEG == 4) {
                Prog::template run<1>(a, lds, tid, reg, twreg);
                tile_sync<C>();
                Prog::template run<2>(a, lds, tid, reg, twreg);
                tile_sync<C>();
                Prog::template run_keep<3>(a, lds, tid, reg, twreg);
            } else if constexpr (Prog::NSEG == 3) {
                Prog::template run<1>(a, lds, tid, reg, twreg);
                tile_sync<C>();
                Prog::template run_keep<2>(a, lds, tid, reg, twreg);
            } else {
                Prog::template run_keep<1>(a, lds, tid, reg, twreg);
            }
            tile_sync<C>();   // the next digit reuses the LDS words
            if (!FP) {
#pragma unroll
                for (int i = 0; i < C::EPT; i++) reg[i] = csub(csub(csub(reg[i], q << 2), q << 1), q);
            }
        }
        const u64 *key = ip.evks[b];
#pragma unroll
        for (int gi = 0; gi < G; gi++)
#pragma unroll
            for (int kk = 0; kk < K; kk += 2) {
                const size_t id = (size_t)prime * n + g0[gi] + kk;
                const u64x2 kb = *reinterpret_cast<const u64x2 *>(key + id);
                const u64x2 ka = *reinterpret_cast<const u64x2 *>(key + id + ip.qp_n);
                const int i0 = gi * K + kk;
                if (FP) {
                    // r04: the lazy outputs of the last round are below M q (M from the pass's compile-time schedule; 2.13 for the
                    // 8-8-4 rounds), so a LIGHT product is below q (0.5 + 0.375 M) and BETA of them stay exact integers below 8 q:
                    // 6 instead of 9 operations per product where that holds (every plan at beta <= 3, all but N = 2^14 at beta = 4)
                    constexpr double mlast = Prog::fp_sched().after[C::NR - 1];
                    constexpr bool light = (0.5 + 0.375 * (mlast > 1.0 ? mlast : 1.0)) * BETA < 7.5;
                    const double x0 = as_f64(reg[i0]), x1 = as_f64(reg[i0 + 1]);
                    auto prod = [&](double x, u64 kw) __attribute__((always_inline)) {
                        return light ? fp_mulmod_light(x, fp_from_canon(kw), fm) : fp_mulmod(x, fp_from_canon(kw), fm);
                    };
                    accb[i0] = as_u64(as_f64(accb[i0]) + prod(x0, kb.x));
                    accb[i0 + 1] = as_u64(as_f64(accb[i0 + 1]) + prod(x1, kb.y));
                    acca[i0] = as_u64(as_f64(acca[i0]) + prod(x0, ka.x));
                    acca[i0 + 1] = as_u64(as_f64(acca[i0 + 1]) + prod(x1, ka.y));
                } else {
                    // (128-bit accumulators with one Barrett at the end, as inner_prod_kernel has them, cost 32 more VGPRs across the
                    //  transforms: 256+ registers, one wavefront per SIMD; measured r03)
                    accb[i0] = add_mod(accb[i0], mul_mod(reg[i0], kb.x, m), q);
                    accb[i0 + 1] = add_mod(accb[i0 + 1], mul_mod(reg[i0 + 1], kb.y, m), q);
                    acca[i0] = add_mod(acca[i0], mul_mod(reg[i0], ka.x, m), q);
                    acca[i0 + 1] = add_mod(acca[i0 + 1], mul_mod(reg[i0 + 1], ka.y, m), q);
                }
            }
    }
    const bool fix = twr == ip.fix_limb;   // (uniform) pha_keyswitch_rescale: ct_last + cx_last * P^-1
    // (uniform) this limb goes back to coefficient form next: run the inverse transform's contiguous pass here (ModupIpArgs::inv_from)
    const bool inv = ip.inv_from != 0xffffffffu && (twr >= ip.inv_from || twr == ip.inv_lead);
#pragma unroll
    for (int gi = 0; gi < G; gi++)
#pragma unroll
        for (int kk = 0; kk < K; kk += 2) {
            const int i0 = gi * K + kk;
            u64x2 rb, ra;
            if (FP) {
                rb = u64x2{fp_to_canon(as_f64(accb[i0]), fm), fp_to_canon(as_f64(accb[i0 + 1]), fm)};
                ra = u64x2{fp_to_canon(as_f64(acca[i0]), fm), fp_to_canon(as_f64(acca[i0 + 1]), fm)};
            } else {
                rb = u64x2{accb[i0], accb[i0 + 1]};
                ra = u64x2{acca[i0], acca[i0 + 1]};
            }
            const size_t id = (size_t)twr * n + g0[gi] + kk;
            if (fix) {
                const u64x2 c0 = *reinterpret_cast<const u64x2 *>(ip.fix_ct + id);
                const u64x2 c1 = *reinterpret_cast<const u64x2 *>(ip.fix_ct + ip.fix_ct_stride + id);
                rb.x = add_mod(c0.x, shoup(rb.x, ip.fix_cst, q), q);
                rb.y = add_mod(c0.y, shoup(rb.y, ip.fix_cst, q), q);
                ra.x = add_mod(c1.x, shoup(ra.x, ip.fix_cst, q), q);
                ra.y = add_mod(c1.y, shoup(ra.y, ip.fix_cst, q), q);
            }
            if (inv) {   // canonical residues, in the layout the inverse pass's first round loads
                accb[i0] = rb.x; accb[i0 + 1] = rb.y;
                acca[i0] = ra.x; acca[i0 + 1] = ra.y;
            } else {
                *reinterpret_cast<u64x2 *>(ip.cx + id) = rb;
                *reinterpret_cast<u64x2 *>(ip.cx + ip.qlp_n + id) = ra;
            }
        }
    if (!inv) return;
    // nwt_2d_radix8_backward's first pass (intt_2d.cu:9-104) on the rows this wavefront owns, from registers: the pass stores what the
    // stand-alone launch would (lazy integers / centred doubles) and the caller launches the strided pass alone
    using InvProg = PassProgram<C, false, EPI_NONE, false, 0, false>;
    NttKArgs ki = k;
    ki.tw = ip.itw;
    ki.twf = ip.itwf;
    ki.in = ki.out = ip.cx;
    ki.batch = 2;
    ki.in_stride = ki.out_stride = ip.qlp_n;
    ki.pro_src = nullptr;
    auto inverse_rows = [&](u64 *r, uint32_t z) __attribute__((always_inline)) {
        PassArgs ai;
        full_tile_args<C, false, EPI_NONE, false>(ki, twr, z, tile, ai);
        ai.fp = FP;
        u64x2 twreg[C::TW_TOTAL];
        InvProg::template run_prefetched<0>(ai, lds, tid, r, twreg);
        tile_sync<C>();
        InvProg::template run_prefetched<1>(ai, lds, tid, r, twreg);
        if constexpr (InvProg::NSEG >= 3) {
            tile_sync<C>();
            InvProg::template run_prefetched<2>(ai, lds, tid, r, twreg);
        }
        if constexpr (InvProg::NSEG == 4) {
            tile_sync<C>();
            InvProg::template run_prefetched<3>(ai, lds, tid, r, twreg);
        }
        tile_sync<C>();   // the second sum reuses the LDS words
    };
    inverse_rows(accb, 0);
    inverse_rows(acca, 1);
}

// Limb order: blockIdx.y walks the special (P) limbs first -- 60-bit primes on the integer back end, the longest wavefronts of
// the launch -- then the data limbs, so that the long poles start at once and the FP64 limbs fill in behind them.
#if defined(PHA_IP_WAVES)
#define PHA_IP_BOUNDS __launch_bounds__(C::THREADS, PHA_IP_WAVES)
#else
#define PHA_IP_BOUNDS __launch_bounds__(C::THREADS)
#endif
template <class C, int BETA>
__global__ PHA_IP_BOUNDS void modup_ip_kernel(const NttKArgs k, const ModupIpArgs ip) {
    static_assert(C::WAVE_LOCAL && !C::STRIDED && !C::WHOLE, "the fused inner product rides on the one-wavefront contiguous pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);
    const uint32_t tile = blockIdx.x;
    const uint32_t n_special = k.sel.remap_from <= k.sel.start + k.sel.count ? k.sel.start + k.sel.count - k.sel.remap_from : 0;
    const uint32_t y = blockIdx.y < n_special ? k.sel.count - n_special + blockIdx.y : blockIdx.y - n_special;
    const uint32_t twr = k.sel.start + y;
    const uint32_t prime = twr >= k.sel.remap_from ? twr + k.sel.remap_add : twr;
    const bool fp = k.fpinfo && k.fpinfo[prime].ok;   // uniform
    if (fp) modup_ip_body<C, BETA, true>(k, ip, twr, prime, tile, lds, threadIdx.x);
    else modup_ip_body<C, BETA, false>(k, ip, twr, prime, tile, lds, threadIdx.x);
}

template <int LOGN>
static void launch_modup_ip(NttKArgs k, uint32_t beta, const ModupIpArgs &ip, hipStream_t s) {
    constexpr int V = (LOGN >= 14 && LOGN <= 16) ? PHA_IP_PLAN : 3;
    using P1 = typename NttPlan<LOGN, V>::P1;
    using P2 = typename NttPlan<LOGN, V>::P2;
    k.t1 = P1::T;
    k.t2 = P2::T;
    k.in = k.mid;                 // the contiguous pass reads what the strided pass left in the digits
    k.in_stride = k.poly_stride;
    k.pro_src = nullptr;
    k.zfast_tiles = 0;
    const dim3 grid((unsigned)(((size_t)1 << LOGN) >> P2::LOGTILE), k.sel.count, 1), block(P2::THREADS);
    const size_t lds_bytes = (size_t)P2::LDS_WORDS * sizeof(u64);
    switch (beta) {
        case 1: hipLaunchKernelGGL((modup_ip_kernel<P2, 1>), grid, block, lds_bytes, s, k, ip); break;
        case 2: hipLaunchKernelGGL((modup_ip_kernel<P2, 2>), grid, block, lds_bytes, s, k, ip); break;
        case 3: hipLaunchKernelGGL((modup_ip_kernel<P2, 3>), grid, block, lds_bytes, s, k, ip); break;
        default: hipLaunchKernelGGL((modup_ip_kernel<P2, 4>), grid, block, lds_bytes, s, k, ip); break;
    }
    check_launch();
}

// What a launch of `count` limbs x `batch` polynomials takes (measured rules):
//   * N = 4096: the whole transform in one launch; N = 8192: the same from 64 limb-polynomials per launch (one 512-thread
//     workgroup per limb: 1 / 10 / 60 / 240 / 1020 limbs 10.4 / 10.8 / 11.5 / 13.7 / 53.8 us against 8.7 / 9.4 / 12.0 / 26.2 / 58.2
//     in two passes);
//   * otherwise two passes, 8 coefficients per thread, one-wavefront workgroups in the contiguous pass;
//   * launches of >= 8 polynomials and >= 8192 tiles run the polynomials of a (tile, limb) back to back on one XCD, which then
//     fetches the twiddle rows they share once (r02: +4.6 % on 16 x 45 limbs at N = 2^16; -3 % on the 16-polynomial launches of
//     config 4 at N = 2^15 and on the 2- / 3-polynomial launches of a key switch, hence the size rule);
//   * other launches of >= 1024 tiles (the memory-bound throughput regime) form the last round's twiddles on the fly; small ones
//     are latency-bound and keep the table-driven last round (r01c).
#ifndef PHA_EPT4_MAX_WAVES
#define PHA_EPT4_MAX_WAVES 8192   // = 8 wavefronts per SIMD: 32 limb-polynomials at N = 2^16, 64 at 2^15, 128 at 2^14 (0 = never)
#endif
#ifndef PHA_SMALL_PLAN
#define PHA_SMALL_PLAN 5     // NttPlan variant of small launches (5: four coefficients per thread in the contiguous pass; 6 / 7: r03 experiments)
#endif
struct NttChoice {
    int v;            // NttPlan variant of the two-pass form (product: 3 or 4)
    int whole;        // 0: two passes; 12 / 13 / 14: the one-workgroup plan of that degree
    bool zfast;       // polynomial-fastest block order in the contiguous pass
    Context *fused;   // experiments: both passes in one launch
};
static inline bool has(int vv, int mask) { return (vv & mask) != 0; }
static NttChoice choose_plan(Context &c, const LimbSel &sel, const NttExtra &x) {
    const int vv = ntt_variant();
    const size_t batch = x.batch ? x.batch : 1;
    const size_t tiles = ((size_t)c.n / kTileElems) * sel.count * batch, limb_polys = (size_t)sel.count * batch;
    const bool shared_tables = has(vv, 2048) && batch >= 8 && tiles >= 8192;
    const bool ot = has(vv, 16) || (has(vv, 32) && has(vv, 1) && tiles >= 1024 && !shared_tables), wave = has(vv, 64) && has(vv, 1);
    NttChoice ch{ot ? (wave ? 4 : 2) : wave ? 3 : has(vv, 1), 0, shared_tables, nullptr};
    // launches that fit ONE co-resident generation of 256-coefficient wavefronts: four coefficients per thread in the contiguous
    // pass, i.e. twice the wavefronts with half the serial work each (r03; N = 2^14 .. 2^16).  Same-box A/B (profiles/r03_experiments.md;
    // kernel traces): a win of 0.4-1.1 us per launch pair up to 8 wavefronts per SIMD (2^16: 1-24 limbs, the 2 x 16-limb
    // inverse of key switch + rescale 14.9 -> 13.6 us; 2^15: every size up to 60 limbs), a loss of 1-2.5 us beyond (2^16: 40-60 limbs)
    if (ch.v == 3 && c.log_n >= 14 && c.log_n <= 16 && limb_polys * (c.n >> 8) <= (size_t)PHA_EPT4_MAX_WAVES) ch.v = 5;
    // r04: N = 2^16 as 64 x 1024 for every launch that is not the first half of the fused mod-up (whose contiguous pass, with the key
    // inner product as its epilogue, is the 256-point one-wavefront pass of plan 3)
    if (c.log_n == 16 && (ch.v == 3 || ch.v == 4) && !x.first_pass_only && !x.second_pass_only) {
        if (has(vv, 4096)) ch.v = 10;
#if defined(PHA_EXPERIMENTS)
        else if (has(vv, 8192)) ch.v = 12;
        else if (has(vv, 16384)) ch.v = 8;
#endif
    }
    if (c.log_n == 12 && !has(vv, 128)) ch.whole = 12;
    if (c.log_n == 13 && !has(vv, 128) && (has(vv, 256) || limb_polys >= 64)) ch.whole = 13;
#if defined(PHA_EXPERIMENTS)
    if (c.log_n == 14 && !has(vv, 128) && (has(vv, 256) || limb_polys >= (size_t)g_whole14_min.load(std::memory_order_relaxed))) ch.whole = 14;
    // both passes in one launch (hand-off through the XCD's L2): launches of >= g_fused_min_tiles tiles (bit 9: every
    // launch; bit 10: never), only where workgroup b is known to run on XCD b % 8
    if (c.xcd_placement_round_robin() && wave && !has(vv, 1024) &&
        (has(vv, 512) || tiles >= (size_t)g_fused_min_tiles.load(std::memory_order_relaxed)))
        ch.fused = &c;
#endif
    if (x.second_pass_only) {   // the other half ran in the fused mod-up: two launches' worth of plan, the split of plan 3
        ch.whole = 0;
        ch.fused = nullptr;
        if (ch.v > 5) ch.v = 3;
    }
    return ch;
}

// the two-pass plans of one degree: the product library instantiates variants 3 and 4 only (and nothing for N = 4096, which
// always takes its one-launch plan there)
template <int LOGN>
static void forward_two_pass(NttKArgs k, int epi, const NttChoice &ch, hipStream_t s) {
#if defined(PHA_EXPERIMENTS)
    switch (ch.v) {
        case 8: if constexpr (LOGN == 16) { forward_impl<16, 8>(k, epi, s); return; }
        case 10: if constexpr (LOGN == 16) { forward_impl<16, 10>(k, epi, s); return; }
        case 12: if constexpr (LOGN == 16) { forward_impl<16, 12>(k, epi, s); return; }
        case 5: if constexpr (LOGN >= 14 && LOGN <= 16) { forward_impl<LOGN, 5>(k, epi, s); return; }
        case 4: forward_impl<LOGN, 4>(k, epi, s, ch.fused); return;
        case 3: forward_impl<LOGN, 3>(k, epi, s, ch.fused); return;
        case 2: forward_impl<LOGN, 2>(k, epi, s); return;
        case 1: forward_impl<LOGN, 1>(k, epi, s); return;
        default: forward_impl<LOGN, 0>(k, epi, s); return;
    }
#else
    if constexpr (LOGN == 12) throw std::logic_error("N = 4096 has no two-pass plan in the product library");
    else if constexpr (LOGN == 13) forward_impl<LOGN, 3>(k, epi, s);   // (below 64 limb-polynomials: never 1024 tiles)
    else if (ch.v == 10) {
        if constexpr (LOGN == 16) forward_impl<16, 10>(k, epi, s);
    }
    else if (ch.v == 4) forward_impl<LOGN, 4>(k, epi, s);
    else if (ch.v == 5) {
        if constexpr (LOGN >= 14 && LOGN <= 16) forward_impl<LOGN, PHA_SMALL_PLAN>(k, epi, s);
    } else forward_impl<LOGN, 3>(k, epi, s);
#endif
}
template <int LOGN>
static void inverse_two_pass(NttKArgs k, int epi, const NttChoice &ch, hipStream_t s) {
#if defined(PHA_EXPERIMENTS)
    switch (ch.v) {
        case 8: if constexpr (LOGN == 16) { inverse_impl<16, 8>(k, epi, s); return; }
        case 10: if constexpr (LOGN == 16) { inverse_impl<16, 10>(k, epi, s); return; }
        case 12: if constexpr (LOGN == 16) { inverse_impl<16, 12>(k, epi, s); return; }
        case 5: if constexpr (LOGN >= 14 && LOGN <= 16) { inverse_impl<LOGN, 5>(k, epi, s); return; }
        case 4: inverse_impl<LOGN, 4>(k, epi, s, ch.fused); return;
        case 3: inverse_impl<LOGN, 3>(k, epi, s, ch.fused); return;
        case 2: inverse_impl<LOGN, 2>(k, epi, s); return;
        case 1: inverse_impl<LOGN, 1>(k, epi, s); return;
        default: inverse_impl<LOGN, 0>(k, epi, s); return;
    }
#else
    if constexpr (LOGN == 12) throw std::logic_error("N = 4096 has no two-pass plan in the product library");
    else if constexpr (LOGN == 13) inverse_impl<LOGN, 3>(k, epi, s);
    else if (ch.v == 10) {
        if constexpr (LOGN == 16) inverse_impl<16, 10>(k, epi, s);
    }
    else if (ch.v == 4) inverse_impl<LOGN, 4>(k, epi, s);
    else if (ch.v == 5) {
        if constexpr (LOGN >= 14 && LOGN <= 16) inverse_impl<LOGN, PHA_SMALL_PLAN>(k, epi, s);
    } else inverse_impl<LOGN, 3>(k, epi, s);
#endif
}

void ntt_forward(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, true);
    const NttChoice ch = choose_plan(c, sel, x);
    k.zfast_tiles = ch.zfast ? 1u : 0u;   // request: launch_pass turns it into the tile count of the contiguous pass
    if (ch.whole == 12) return forward_whole<WholePlan12>(k, epi, s);
    if (ch.whole == 13) return forward_whole<WholePlan13>(k, epi, s);
#if defined(PHA_EXPERIMENTS)
    if (ch.whole == 14) return forward_whole<WholePlan14>(k, epi, s);
#endif
    switch (c.log_n) {
        case 12: forward_two_pass<12>(k, epi, ch, s); break;
        case 13: forward_two_pass<13>(k, epi, ch, s); break;
        case 14: forward_two_pass<14>(k, epi, ch, s); break;
        case 15: forward_two_pass<15>(k, epi, ch, s); break;
        case 16: forward_two_pass<16>(k, epi, ch, s); break;
        case 17: forward_two_pass<17>(k, epi, ch, s); break;
        default: throw std::invalid_argument("unsupported polynomial degree");
    }
}

void ntt_inverse(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s) {
    if (sel.count == 0) return;
    check_sel(c, sel);
    NttKArgs k = make_args(c, in, mid, out, sel, x, false);
    const NttChoice ch = choose_plan(c, sel, x);
    k.zfast_tiles = ch.zfast ? 1u : 0u;
    if (ch.whole == 12) return inverse_whole<WholePlan12>(k, epi, s);
    if (ch.whole == 13) return inverse_whole<WholePlan13>(k, epi, s);
#if defined(PHA_EXPERIMENTS)
    if (ch.whole == 14) return inverse_whole<WholePlan14>(k, epi, s);
#endif
    switch (c.log_n) {
        case 12: inverse_two_pass<12>(k, epi, ch, s); break;
        case 13: inverse_two_pass<13>(k, epi, ch, s); break;
        case 14: inverse_two_pass<14>(k, epi, ch, s); break;
        case 15: inverse_two_pass<15>(k, epi, ch, s); break;
        case 16: inverse_two_pass<16>(k, epi, ch, s); break;
        case 17: inverse_two_pass<17>(k, epi, ch, s); break;
        default: throw std::invalid_argument("unsupported polynomial degree");
    }
}

bool modup_ntt_inner_prod(Context &c, u64 *digits, const LimbSel &sel, const NttExtra &x, uint32_t beta, const ModupIpArgs &ip,
                          hipStream_t s) {
    if (c.log_n < 14 || c.log_n > 17 || beta < 1 || beta > 4 || sel.count == 0) return false;   // (N <= 8192 takes the one-launch plans)
    if (choose_plan(c, sel, x).whole) return false;
    check_sel(c, sel);
    NttExtra x1 = x;
    x1.first_pass_only = true;
    ntt_forward(c, digits, digits, digits, sel, EPI_FWD_CANON, x1, s);   // strided pass of every digit, in place
    NttKArgs k = make_args(c, digits, digits, digits, sel, x, true);
    switch (c.log_n) {
        case 14: launch_modup_ip<14>(k, beta, ip, s); break;
        case 15: launch_modup_ip<15>(k, beta, ip, s); break;
        case 16: launch_modup_ip<16>(k, beta, ip, s); break;
        default: launch_modup_ip<17>(k, beta, ip, s); break;
    }
    return true;
}


// ---- r05: the mod-up's base conversion fused with the strided pass of the forward transform -------------------
// bconv_kernel (pha_rns.hip) writes beta x (l + alpha - alpha_b) converted limbs in coefficient form and the strided pass of the
// forward transform reads them back: 67.5 + 67.5 MiB per key switch at C3, plus a launch.  Here the conversion is the LOAD of that
// pass: a workgroup owns a strided-pass tile (T1 rows x V adjacent columns) of ONE digit polynomial, each thread reads the digit's
// <= 16 scaled input residues of ITS coefficients once (the first round's register layout: EPT coefficients per thread, so the
// inputs are EPT x isz words = 30 registers per coefficient as 30-bit halves -- hence four coefficients per thread and radix-4
// rounds, where the stand-alone pass has eight and radix-8), and then walks the digit's output limbs: convert (the carry-free
// 30 / 30 MAC against the LDS-staged Montgomery rows, REDC: the arithmetic of bconv_kernel, the same canonical residues), run
// the pass from registers (PassProgram::run_prefetched), store the pass's output where the contiguous pass expects it.
// Per output limb everything uniform (modulus, REDC constant, destination limb, table row, FP64 constants) comes from an LDS record
// staged once per workgroup, so the loop has no dependent scalar-memory round trip.
// LOGTILE 8: one wavefront per workgroup (64 rows x 4 columns: 32-byte runs, no workgroup barrier); 10: 64 x 16 (128-byte runs,
// 256 threads); 11: 64 x 32.
struct ConvLimbRec {      // 56 bytes; one per output limb of the digit
    u64 p, oninv;
    double fq, fqinv;
    uint32_t jo, prime, fpok, pad;
    u64 ratio1;           // floor(2^64 / p): the rescale form reduces c_last modulo this limb's prime (barrett64)
};
constexpr int kMcsMaxOut = 64;
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 uni64(u64 x) { return ((u64)uni32((uint32_t)(x >> 32)) << 32) | uni32((uint32_t)x); }

// The geometry below is the one kept out of the 12 that r05 measured (profiles/r05_experiments.md §2: batched HomMul + relinearize +
// rescale, us per op at B = 8 / 32; separate kernels 279.5 / 275.0): every round's twiddles requested before the conversion, all four
// coefficients converted side by side (a matrix row is read from LDS once), 8 of the 15 input limbs in registers (226 registers, two
// 256-thread workgroups per CU with 76 KB of LDS each): 275.1 / 275.2.  The losers (10 limbs in registers 290 / 286; 7: 322;
// per-round twiddles 284; 64 x 4 tiles 365; 64 x 8 295; 64 x 32 292; one wavefront per SIMD 339; staggered workgroups, raised
// priority for the rounds: no gain) are in that log, not in this source.
constexpr int kMcsWaves = 2;      // wavefronts per SIMD the register allocation aims at (the inputs alone are 120 registers)
constexpr int kMcsRegLimbs = 8;   // input limbs held in registers; the rest of the digit's inputs in LDS
#ifndef PHA_RESC_REGLIMBS
#define PHA_RESC_REGLIMBS 15   // r06: the rescale form has no copy loop and fewer live pointers (174 registers with 8 limbs in registers): all 15 inputs fit, no LDS input reads (B = 8: 256.6 -> 251.6 us per op, B = 32: 254.0 -> 252.0)
#endif
#ifndef PHA_DOWN_REGLIMBS
#define PHA_DOWN_REGLIMBS 8    // the plain mod-down form spills at 12 (76 bytes of scratch) and 15 (260 bytes: 280 instead of 230 us per key switch)
#endif
constexpr int mcs_reg_limbs(int form) { return form == 1 ? PHA_RESC_REGLIMBS : form == 2 ? PHA_DOWN_REGLIMBS : kMcsRegLimbs; }
constexpr int kMcsLogTile = 10;   // 64 rows x 16 columns: 128-byte runs, 256 threads
// RESC (r06): the same kernel as the load of the forward transform that ends key switch + rescale (pha_keyswitch_rescale_batched): the
// conversion is bconv_rescale_kernel's (pha_rns.hip) -- inputs = the P limbs of cx in coefficient form (x phat_i^-1), ONE converter
// (P -> Ql with P^-1 in its rows) whose LAST row belongs to the last data limb: c_last = t_last - convert(last row) mod q_last once per
// coefficient, then v_j = convert(row j) + (c_last mod q_j) for j < Ql - 1 goes straight into the strided pass's registers.  The
// 2 x (Ql - 1) limbs of v are then never written and read back in coefficient form (92 MB per ciphertext at C3) and one launch goes.
// FORM 2 (r06): the plain mod-down form -- the same without the last-limb fold: every row of the P -> Ql converter is an output limb (the CKKS
// mod-down of a batched key switch WITHOUT rescale; the transform's contiguous pass then carries the (cx - NTT delta) P^-1 epilogue)
template <class C, int ISZ_PAD, int FORM = 0>
__global__ __launch_bounds__(C::THREADS) __attribute__((amdgpu_waves_per_eu(kMcsWaves, kMcsWaves)))
void modup_conv_s1_kernel(const NttKArgs k, const ModupConvArgs m) {
    constexpr bool RESC = FORM == 1;
    static_assert(C::STRIDED && C::EPT == 4 && C::NR == 3 && C::r(0) == 2, "radix-4 strided pass, four coefficients per thread");
    constexpr int K = 4, LOGD = C::LOGT - 2;
    using Prog = PassProgram<C, true, EPI_NONE, false, 1, false>;   // HOIST 1: a limb's twiddles are requested before its conversion
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *lds = reinterpret_cast<u64 *>(smem);                                                  // the pass's tile
    uint2 *s_rows = reinterpret_cast<uint2 *>(smem + (size_t)C::LDS_WORDS * sizeof(u64));      // [max_osz][kBcRowPad]
    ConvLimbRec *s_rec = reinterpret_cast<ConvLimbRec *>(s_rows + m.max_osz * kBcRowPad);      // [max_osz]
    const int tid = threadIdx.x;
    const uint32_t z = blockIdx.z, tile = blockIdx.x;
    const uint32_t ci = z % m.beta, grp = z / m.beta;
    const BConvDev &d = m.convs[ci];
    const uint32_t n = 1u << k.log_n, isz = d.isz, osz = d.osz, pitch = d.row_pad;
    const bool r90 = uni32(d.r90) != 0;
    for (uint32_t e = tid; e < osz * kBcRowPad; e += C::THREADS) {
        const uint32_t j = e / kBcRowPad, i = e % kBcRowPad;
        s_rows[e] = i < pitch ? reinterpret_cast<const uint2 *>(d.mat30)[j * pitch + i] : uint2{0u, 0u};
    }
    for (uint32_t j = tid; j < osz; j += C::THREADS) {
        ConvLimbRec r;
        r.jo = j + (j >= d.pad_start ? d.pad_len : 0);                                   // limb of the [Ql | P] digit buffer
        r.prime = r.jo >= k.sel.remap_from ? r.jo + k.sel.remap_add : r.jo;              // its row of the QP tables (= d.oprime[j])
        r.p = k.mod[r.prime].value;
        r.oninv = d.oninv[j];
        r.fpok = 0; r.fq = 0.0; r.fqinv = 0.0; r.pad = 0;
        r.ratio1 = k.mod[r.prime].ratio1;
        if (k.fpinfo) {
            const FpInfo fi = k.fpinfo[r.prime];
            r.fpok = fi.ok; r.fq = fi.q; r.fqinv = fi.qinv;
        }
        s_rec[j] = r;
    }
    // this thread's coefficients: the first round's layout (round_load<C, 0>): rows lo + (kk << LOGD) of column tile * V + v
    int v, hi, lo;
    decode_group<C, 0>(tid, v, hi, lo);
    uint32_t off[K];
#pragma unroll
    for (int kk = 0; kk < K; kk++) off[kk] = (uint32_t)(lo + (kk << LOGD)) * k.t2 + tile * C::V + (uint32_t)v;
    const u64 *src = m.src + (size_t)grp * m.src_group_stride + (size_t)d.src_limb * n;
    u64 *dig = k.out + (size_t)z * k.poly_stride;
    // the first RL input limbs live in registers (8 per limb: four coefficients x two 30-bit halves), the others in LDS, [limb][kk][thread]
    // as (lo, hi) pairs: conflict-free 8-byte reads.  All in registers is 120 + the pass's ~130: spills at two wavefronts per SIMD.
    constexpr int RL = mcs_reg_limbs(FORM) < ISZ_PAD ? mcs_reg_limbs(FORM) : ISZ_PAD;
    uint2 *s_in = reinterpret_cast<uint2 *>(s_rec + m.max_osz);                                // [ISZ_PAD - RL][K][THREADS]
    u32 ylo[RL][K], yhi[RL][K];
#pragma unroll
    for (int i = 0; i < ISZ_PAD; i++)
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
            const u64 y = i < (int)isz ? src[(size_t)i * n + off[kk]] : 0;
            if (i < RL) {
                ylo[i][kk] = (u32)y & 0x3fffffffu;
                yhi[i][kk] = (u32)(y >> 30);
            } else {
                s_in[((i - RL) * K + kk) * C::THREADS + tid] = uint2{(u32)y & 0x3fffffffu, (u32)(y >> 30)};
            }
        }
    if (d.copy_own && m.own) {   // modup_copy_partQl_kernel rns_bconv.cu:522-528
        const u64 *own = m.own + (size_t)grp * m.own_group_stride + (size_t)d.src_limb * n;
        for (uint32_t i = 0; i < isz; i++)
#pragma unroll
            for (int kk = 0; kk < K; kk++) dig[(size_t)(d.src_limb + i) * n + off[kk]] = own[(size_t)i * n + off[kk]];
    }
    __syncthreads();
    // out[kk] = REDC(sum_i y_i[kk] * row_e[i]) for the thread's four coefficients: sum_i y_i qhat_i mod p, canonical (the arithmetic of
    // bconv_kernel, SPLIT 30 / 30, Montgomery rows; all four coefficients side by side, so a matrix row is read from LDS once)
    auto convert4 = [&](uint32_t e, u64 p, u64 oninv, u64 *out) __attribute__((always_inline)) {
        const uint2 *row = s_rows + e * kBcRowPad;
        u64 ll[K], lh[K], hl[K], hh[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) ll[kk] = lh[kk] = hl[kk] = hh[kk] = 0;
#pragma unroll
        for (int i = 0; i < ISZ_PAD; i++) {
            const uint2 mm = row[i];
#pragma unroll
            for (int kk = 0; kk < K; kk++) {
                u32 y0, y1;
                if (i < RL) {
                    y0 = ylo[i < RL ? i : 0][kk];
                    y1 = yhi[i < RL ? i : 0][kk];
                } else {
                    const uint2 yy = s_in[((i - RL) * K + kk) * C::THREADS + threadIdx.x];
                    y0 = yy.x;
                    y1 = yy.y;
                }
                ll[kk] = (u64)y0 * mm.x + ll[kk];
                lh[kk] = (u64)y0 * mm.y + lh[kk];
                hl[kk] = (u64)y1 * mm.x + hl[kk];
                hh[kk] = (u64)y1 * mm.y + hh[kk];
            }
        }
        if (r90) {   // (uniform) rows carry 2^90: word-wise REDC from the accumulators (pha_arith.h)
            const u32 p0 = (u32)p & 0x3fffffffu, p1 = (u32)(p >> 30);
#pragma unroll
            for (int kk = 0; kk < K; kk++) out[kk] = mont_redc90_split(ll[kk], lh[kk], hl[kk], hh[kk], p, p0, p1, (u32)oninv);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
            const u64 mid = lh[kk] + hl[kk], mid_c = mid < lh[kk] ? 1 : 0;
            u64 rl = ll[kk], rh = 0;
            const u64 t1 = mid << 30;
            rl += t1;
            rh += (rl < t1) + (mid >> 34) + (mid_c << 30);
            const u64 t2 = hh[kk] << 60;
            rl += t2;
            rh += (rl < t2) + (hh[kk] >> 4);
            out[kk] = mont_redc128(rl, rh, p, oninv);
        }
    };
    u64 c_last[K];                       // RESC: t_last - dP_last mod q_last of the thread's four coefficients
    uint32_t n_out = osz;
    if (RESC) {
        n_out = osz - 1;                 // the converter's last row is the last data limb's
        const ConvLimbRec rl = s_rec[n_out];
        const u64 ql_ = uni64(rl.p);
        u64 dl[K];
        convert4(n_out, ql_, uni64(rl.oninv), dl);
        const u64 *last = m.resc_last + (size_t)grp * m.src_group_stride;
#pragma unroll
        for (int kk = 0; kk < K; kk++) c_last[kk] = sub_mod(last[off[kk]], dl[kk], ql_);
    }
#pragma unroll 1
    for (uint32_t e = 0; e < n_out; e++) {
        const ConvLimbRec rc = s_rec[e];
        const u64 p = uni64(rc.p), oninv = uni64(rc.oninv);
        const uint32_t jo = uni32(rc.jo), prime = uni32(rc.prime), fpok = uni32(rc.fpok);
        PassArgs a;
        a.in = nullptr;
        a.out = dig + (size_t)jo * n;
        a.tw = k.tw + (size_t)prime * n;
        a.twd = k.twf + (size_t)prime * n;
        a.q = p;
        a.tile = tile;
        a.rho0 = k.t1;
        a.stride = k.t2;
        a.aux = a.aux2 = nullptr;
        a.pro_reduce = false;
        a.pro_ratio1 = 0;
        a.fp = fpok != 0;
        a.fpm = FpMod{as_f64(uni64(as_u64(rc.fq))), as_f64(uni64(as_u64(rc.fqinv))), (fpok & 2) != 0, (fpok & 4) != 0};
        u64 reg[K];
        u64x2 twreg[C::TW_TOTAL];
        // the pass's LDS / global offsets are functions of the thread index alone: recomputed per limb (a few dozen scalar-like
        // vector instructions against ~2500) instead of living in ~40 registers next to the 120 of the inputs
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        auto limb = [&](const PassArgs &pa) __attribute__((always_inline)) {
            const int tid = tid_e;
            Prog::load_twiddles(pa, tid, twreg);
            convert4(e, p, oninv, reg);
            if (RESC) {   // v_j = dP_j + (c_last mod q_j)   (bconv_rescale_kernel; divide_and_round_reduce_q_last_kernel rns.cu:1128-1139)
                const u64 ratio1 = uni64(rc.ratio1);
#pragma unroll
                for (int kk = 0; kk < K; kk++) reg[kk] = add_mod(reg[kk], barrett64(c_last[kk], p, ratio1), p);
            }
            Prog::template run_prefetched<0>(pa, lds, tid, reg, twreg);
            tile_sync<C>();
            Prog::template run_prefetched<1>(pa, lds, tid, reg, twreg);
            tile_sync<C>();
            Prog::template run_prefetched<2>(pa, lds, tid, reg, twreg);
            tile_sync<C>();   // the next limb's first round writes the same LDS words
        };
        if (a.fp) {   // (uniform) one specialised body per butterfly back end, as exec_pass has them
            PassArgs b = a;
            b.fp = true;
            limb(b);
        } else {
            PassArgs b = a;
            b.fp = false;
            limb(b);
        }
    }
}

#ifndef PHA_MODUP_CONV_FUSE
#define PHA_MODUP_CONV_FUSE 1
#endif
#ifndef PHA_MCS_MIN_WORKGROUPS
#if defined(PHA_EXPERIMENTS)
#define PHA_MCS_MIN_WORKGROUPS 1      // the test-only library: every N = 2^16 mod-up takes the fused form, so that the key-switch parity tests reach it with ONE ciphertext (tests/test_gpu_ntt_variants.py)
#else
#define PHA_MCS_MIN_WORKGROUPS 1024
#endif
#endif
bool modup_conv_strided(Context &c, u64 *digits, const LimbSel &sel, const NttExtra &x, const ModupConvArgs &m, hipStream_t s) {
#if !PHA_MODUP_CONV_FUSE
    return false;
#else
    // the fused pass replaces the strided pass of NttPlan<16, 10> (64 rows x 1024-point rows): N = 2^16 launches that take that plan
    if (c.log_n != 16 || !m.mont_split || m.max_isz > 16 || m.max_osz > (uint32_t)kMcsMaxOut || sel.count == 0 || sel.start != 0) return false;
    NttExtra xd = x;
    xd.first_pass_done = true;
    const NttChoice ch = choose_plan(c, sel, xd);
    if (ch.whole || ch.fused || ch.v != 10) return false;
    using P1 = NttPlan<16, 10>::P1;
    using P2 = NttPlan<16, 10>::P2;
    using C = PassCfg<6, true, 2, 2, 2, 4, false, kMcsLogTile>;    // = NttPlan<16, 13>::P1 (the CPU replay runs that plan)
    static_assert(C::LOGT == P1::LOGT, "the fused pass is the strided pass of the plan whose contiguous pass follows");
    static_assert(std::is_same<C, NttPlan<16, 13>::P1>::value, "NttPlan<16, 13> names this pass");
    check_sel(c, sel);
    NttKArgs k = make_args(c, digits, digits, digits, sel, x, true);
    k.t1 = P1::T;
    k.t2 = P2::T;
    auto lds_for = [&](int isz_pad, uint32_t max_osz) {
        const int rl = mcs_reg_limbs(m.resc_last ? 1 : m.moddown ? 2 : 0), in_lds = isz_pad > rl ? isz_pad - rl : 0;
        return (size_t)C::LDS_WORDS * sizeof(u64) + (size_t)max_osz * kBcRowPad * sizeof(uint2) + (size_t)max_osz * sizeof(ConvLimbRec) +
               (size_t)in_lds * 4 * C::THREADS * sizeof(uint2);
    };
    const dim3 grid((unsigned)(c.n >> C::LOGTILE), 1, k.batch), block(C::THREADS);
    // a workgroup walks all output limbs of its tile (45 at the top level: ~200 us), so the form only pays when the launch fills the
    // device several times over: two workgroups per CU are resident, four generations of them = batches of >= 6 ciphertexts at
    // beta = 3 (measured: equal to the separate kernels at B = 32, 1.5 % ahead at B = 8; a single ciphertext would wait ~200 us for
    // 192 workgroups where the separate kernels take 67)
    if ((size_t)grid.x * grid.z < (size_t)PHA_MCS_MIN_WORKGROUPS) return false;
    auto go = [&](auto kern, int isz_pad) {
        // the request grows with the level's output-limb count (176 B per limb of max_osz), so the limit is raised ONCE per kernel and
        // device to the WORST case (kMcsMaxOut output limbs), not to the first call's size (a later, larger level would
        // otherwise ask for more than the recorded limit)
        const size_t lds_bytes = lds_for(isz_pad, m.max_osz), lds_worst = lds_for(isz_pad, kMcsMaxOut);
        static_assert(kMcsMaxOut <= 64, "worst-case LDS request must stay inside the CU's 160 KiB");
        if (lds_worst > 160 * 1024) throw std::logic_error("modup_conv_strided: worst-case LDS request exceeds the CU");
        if (lds_worst > 64 * 1024) {   // beyond the default dynamic LDS limit
            static std::atomic<uint64_t> raised[6] = {{0}, {0}, {0}, {0}, {0}, {0}};
            int dev = 0;
            PHA_HIP(hipGetDevice(&dev));
            const uint64_t bit = 1ull << (dev & 63);
            std::atomic<uint64_t> &r = raised[(isz_pad == 15 ? 0 : 1) + (m.resc_last ? 2 : m.moddown ? 4 : 0)];
            if (!(r.load(std::memory_order_acquire) & bit)) {
                PHA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_worst));
                r.fetch_or(bit, std::memory_order_release);
            }
        }
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, k, m);
    };
    if (m.resc_last) {   // rescale form: one converter, its last row is the last data limb's (and yields no output limb)
        if (m.beta != 1 || sel.count + 1 != m.max_osz) throw std::logic_error("modup_conv_strided: rescale form needs the P -> Ql converter of this level");
        if (m.max_isz == 15) go(modup_conv_s1_kernel<C, 15, 1>, 15);
        else if (m.max_isz > 8) go(modup_conv_s1_kernel<C, 16, 1>, 16);
        else return false;
    } else if (m.moddown) {   // plain mod-down form: one converter, every row an output limb
        if (m.beta != 1 || sel.count != m.max_osz) throw std::logic_error("modup_conv_strided: mod-down form needs the P -> Ql converter of this level");
        if (m.max_isz == 15) go(modup_conv_s1_kernel<C, 15, 2>, 15);
        else if (m.max_isz > 8) go(modup_conv_s1_kernel<C, 16, 2>, 16);
        else return false;
    } else if (m.max_isz == 15) go(modup_conv_s1_kernel<C, 15>, 15);
    else if (m.max_isz > 8) go(modup_conv_s1_kernel<C, 16>, 16);
    else return false;
    check_launch();
    return true;
#endif
}

}  // namespace pha

using namespace pha;

static void need(const void *p) {
    if (!p) throw std::invalid_argument("null device pointer");
}

extern "C" {

int pha_nwt_2d_radix8_forward_inplace(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                          size_t start, size_t size_QP, size_t size_P,
                                                          void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    ntt_forward(ctx->c, inout, inout, inout, special_sel(start, cms, size_QP, size_P), EPI_FWD_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(pha_context_t ctx, uint64_t *inout,
                                                                        size_t cms, size_t start, size_t size_QP,
                                                                        size_t size_P, size_t ex_start,
                                                                        size_t ex_end, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    LimbSel sel = special_sel(start, cms, size_QP, size_P);
    sel.excl_start = (uint32_t)ex_start;
    sel.excl_end = (uint32_t)ex_end;
    ntt_forward(ctx->c, inout, inout, inout, sel, EPI_FWD_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_fuse_moddown(pha_context_t ctx, uint64_t *ct, const uint64_t *cx,
                                                   const uint64_t *pinv, const uint64_t *pinv_shoup,
                                                   uint64_t *delta, size_t cms, size_t start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(ct); need(cx); need(pinv); need(pinv_shoup); need(delta);
    NttExtra x;
    x.scale = pinv;
    x.scale_shoup = pinv_shoup;
    x.aux = cx;
    // pass 1 in place on delta; pass 2 reads delta, fuses (cx - NTT(delta)) * PInv and writes ct
    // (ntt_moddown.cu:106-261).  ct may alias cx: every thread reads cx[i] before it writes ct[i].
    ntt_forward(ctx->c, delta, delta, ct, plain_sel(start, cms), EPI_FWD_MODDOWN, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_inverse(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms, size_t start,
                               void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(out); need(in);
    ntt_inverse(ctx->c, in, out, out, plain_sel(start, cms), EPI_INV_CANON, NttExtra{}, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_scale(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t cms,
                                     size_t start, const uint64_t *scale, const uint64_t *scale_shoup,
                                     void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(out); need(in); need(scale); need(scale_shoup);
    NttExtra x;
    x.scale = scale;
    x.scale_shoup = scale_shoup;
    ntt_inverse(ctx->c, in, out, out, plain_sel(start, cms), EPI_INV_SCALE, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_scale(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                             const uint64_t *scale, const uint64_t *scale_shoup, void *stream) {
    return pha_nwt_2d_radix8_backward_scale(ctx, inout, inout, cms, start, scale, scale_shoup, stream);
}

int pha_nwt_2d_radix8_backward_inplace_include_special_mod(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                           size_t start, size_t size_QP, size_t size_P,
                                                           void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (size_P > cms) throw std::invalid_argument("size_P exceeds coeff_modulus_size");
    ntt_inverse(ctx->c, inout, inout, inout, special_sel(start, cms, size_QP, size_P), EPI_INV_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

// BEHZ base Bsk = B u {m_sk} (src/evaluate.cu:434,528).  The reference keeps the tables of Bsk u {m_tilde} in their own
// DNTTTable and sends the last data limb (m_sk) to its last row (fntt_2d.cu:226, intt_2d.cu:334); here those primes
// are auxiliary rows of the context's one table set, so the selector is a plain offset.
static LimbSel temp_mod_sel(Context &c, size_t cms, size_t start, size_t total) {
    Behz &b = c.behz();
    if (start != 0 || cms != b.size_bsk || total != (size_t)b.size_bsk + 1)
        throw std::invalid_argument("include_temp_mod transforms a whole Bsk buffer: coeff_modulus_size = |Bsk|, total = |Bsk| + 1");
    LimbSel s = plain_sel(0, cms);
    s.remap_from = 0;
    s.remap_add = b.aux0;
    return s;
}

int pha_nwt_2d_radix8_forward_inplace_include_temp_mod(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                                       size_t total_modulus_size, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    ntt_forward(ctx->c, inout, inout, inout, temp_mod_sel(ctx->c, cms, start, total_modulus_size), EPI_FWD_CANON, NttExtra{},
                as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale(pha_context_t ctx, uint64_t *inout, size_t cms,
                                                              size_t start, size_t total_modulus_size,
                                                              const uint64_t *scale, const uint64_t *scale_shoup,
                                                              void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout); need(scale); need(scale_shoup);
    NttExtra x;
    x.scale = scale;
    x.scale_shoup = scale_shoup;
    ntt_inverse(ctx->c, inout, inout, inout, temp_mod_sel(ctx->c, cms, start, total_modulus_size), EPI_INV_SCALE, x,
                as_stream(stream));
    PHA_API_END
}

// out[limb] = NTT modulo q_{modulus_index} of in[limb] for the limbs [start, start + cms): the reference lifts a
// plaintext (coefficients below t < q) into one RNS limb at a time this way (ntt_keyswitch_old.cu:225-265, callers
// evaluate.cu:1152,1210,1321).  Inputs are reduced modulo the prime as they are loaded (the reference leaves
// that reduction commented out, :47-49: same results for its inputs below q).
int pha_nwt_2d_radix8_forward_modup_fuse(pha_context_t ctx, uint64_t *out, const uint64_t *in, size_t modulus_index,
                                         size_t cms, size_t start, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(out); need(in);
    Context &c = ctx->c;
    if (modulus_index >= c.size_qp) throw std::invalid_argument("modulus_index out of range");
    if (cms == 0) return 0;
    if (cms > 65535) throw std::invalid_argument("coeff_modulus_size out of range");
    // ONE launch pair for all limbs: limb start + z is "polynomial" z of a batch of one-limb polynomials that sit n
    // coefficients apart, all of them transformed with table row modulus_index
    LimbSel sel = plain_sel(start, 1);
    sel.remap_from = (uint32_t)start;
    sel.remap_add = (uint32_t)modulus_index - (uint32_t)start;
    NttExtra x;
    x.batch = (uint32_t)cms;
    x.poly_stride = c.n;
    x.pro_src = in + start * c.n;
    x.pro_stride = c.n;
    ntt_forward(c, out, out, out, sel, EPI_FWD_CANON, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_forward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                              size_t batch, size_t poly_stride, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (batch == 0 || batch > 65535) throw std::invalid_argument("batch out of range");
    NttExtra x;
    x.batch = (uint32_t)batch;
    x.poly_stride = poly_stride;
    ntt_forward(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_FWD_CANON, x, as_stream(stream));
    PHA_API_END
}

int pha_nwt_2d_radix8_backward_inplace_batched(pha_context_t ctx, uint64_t *inout, size_t cms, size_t start,
                                               size_t batch, size_t poly_stride, void *stream) {
    PHA_CTX_BEGIN(ctx)
    need(inout);
    if (batch == 0 || batch > 65535) throw std::invalid_argument("batch out of range");
    NttExtra x;
    x.batch = (uint32_t)batch;
    x.poly_stride = poly_stride;
    ntt_inverse(ctx->c, inout, inout, inout, plain_sel(start, cms), EPI_INV_CANON, x, as_stream(stream));
    PHA_API_END
}

#if defined(PHA_FUSED_DEBUG)
int pha_fused_debug_read(uint32_t *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_dbg), sizeof(uint32_t) * 8); }
#endif
#if defined(PHA_EXP_STAMPS)
int pha_exp_read_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
}
int pha_exp_read_wg_times(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 2048);
}
#endif

#if defined(PHA_EXPERIMENTS)
int pha_set_tuning(int key, int value) {
    PHA_API_BEGIN
    if (key == 0) {
        if (value < 0 || value > 32767 || (value & 6)) throw std::invalid_argument("unknown NTT variant");
        g_ntt_variant.store(value);
    } else if (key == 1) {
        g_bconv_split.store(value ? 1 : 0);
    } else if (key == 2) {
        if (value < 1) throw std::invalid_argument("threshold must be positive");
        g_whole14_min.store(value);
    } else if (key == 3) {
        if (value < 0 || value > 64) throw std::invalid_argument("lag out of range");
        g_fused_lag.store(value);
    } else if (key == 5) {
        g_fused_split.store(value ? 1 : 0);
    } else if (key == 4) {
        if (value < 1) throw std::invalid_argument("threshold must be positive");
        g_fused_min_tiles.store(value);
    } else {
        throw std::invalid_argument("unknown tuning key");
    }
    PHA_API_END
}
#endif  // PHA_EXPERIMENTS

}  // extern "C"
