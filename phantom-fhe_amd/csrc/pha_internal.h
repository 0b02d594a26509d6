// pha_internal.h -- host-side state behind the C ABI (include/phantom_amd.h).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "pha_arith.h"

namespace pha {

// ---- errors: the C ABI turns these into status codes (phantom_amd.h) -------------------------
void set_last_error(const char *what);
int translate_exception();  // call inside catch(...)

#define PHA_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            throw std::runtime_error(std::string("HIP Runtime Error: ") + hipGetErrorString(e__) + \
                                     " at " #expr);                                      \
    } while (0)

#define PHA_API_BEGIN try {
// entry points that take a context run on the context's device whatever the calling thread's current device is, and
// leave that current device as they found it
#define PHA_CTX_BEGIN(ctx) \
    try {                  \
        if (!(ctx)) throw std::invalid_argument("null context"); \
        pha::DeviceGuard device_guard__((ctx)->c.device);
#define PHA_API_END                  \
    return 0;                        \
    }                                \
    catch (...) {                    \
        return pha::translate_exception(); \
    }

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            PHA_HIP(hipSetDevice(dev));
            changed = true;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// ---- device buffer (RAII) --------------------------------------------------------------------
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), count(o.count) { o.p = nullptr; o.count = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; count = o.count; o.p = nullptr; o.count = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n) PHA_HIP(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
        count = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    void upload(const std::vector<T> &h) {
        if (h.size() != count) alloc(h.size());
        if (count) PHA_HIP(hipMemcpy(p, h.data(), count * sizeof(T), hipMemcpyHostToDevice));
    }
    // keep the old contents, append `tail` (table rows of auxiliary primes)
    void append(const std::vector<T> &tail) {
        DevBuf<T> bigger(count + tail.size());
        if (count) PHA_HIP(hipMemcpy(bigger.p, p, count * sizeof(T), hipMemcpyDeviceToDevice));
        if (!tail.empty())
            PHA_HIP(hipMemcpy(bigger.p + count, tail.data(), tail.size() * sizeof(T), hipMemcpyHostToDevice));
        *this = std::move(bigger);
    }
};

constexpr int kBcRowPad = 16;  // row pitch (entries) of the split base-conversion matrix

// per-prime constants of the FP64 NTT path
struct FpInfo {
    double q, qinv;
    uint32_t ok, pad;
};

// ---- fast base converter constants (DBaseConverter, include/rns_bconv.cuh:3-87) --------------
// Device-resident description of one converter (uploaded once per Tool); a launch indexes an array
// of these with blockIdx.z (one per mod-up digit) or uses the same one for every z (mod-down polys).
struct BConvDev {
    const u64x2 *hat_inv;        // [isz] qhat_i^-1 mod q_i (+Shoup), used when SCALE_IN
    const uint32_t *iprime;      // [isz] rows of the QP table
    const uint32_t *oprime;      // [osz]
    const u64 *mat;              // [osz][isz] qhat_i mod p_j
    const uint32_t *mat30;       // [osz][kBcRowPad][2] 30-bit halves, rows zero-padded, of qhat_i mod p_j -- or, when oninv is set,
                                 //   of its Montgomery form qhat_i * 2^64 mod p_j (the split kernel then reduces with REDC)
    const u64 *oninv;            // [osz] -p_j^-1 mod 2^64, or null (an even output modulus: Barrett)
    uint32_t isz, osz;
    uint32_t pad_start, pad_len; // output j goes to limb j + (j >= pad_start ? pad_len : 0)
    uint32_t src_limb;           // first input limb inside the source polynomial
    uint32_t copy_own;           // mod-up: also copy the digit's own limbs [src_limb, src_limb+isz) verbatim
    uint32_t row_pad;            // row pitch (entries) of mat30: kBcRowPad, or 32 for 17..32 inputs (split kinds 2, 3)
    uint32_t r90;                // r06: the Montgomery rows carry 2^90 instead of 2^64 (<= 15 inputs, 30 / 30 cuts): mont_redc90_split
};


struct BConv {
    uint32_t isz = 0, osz = 0;
    std::vector<uint32_t> iprime, oprime;  // indices into the QP table
    DevBuf<u64x2> hat_inv;                 // [isz]  qhat_i^-1 mod q_i (+Shoup)
    DevBuf<u64> mat;                       // [osz][isz]  qhat_i mod p_j
    DevBuf<uint32_t> mat30;                // [osz][kBcRowPad][2] 30-bit halves of mat (Montgomery form when mont), zero-padded rows
    DevBuf<u64> oninv;                     // [osz] -p_j^-1 mod 2^64
    bool mont = false;                     // every output modulus is odd: the split kernel reduces with Montgomery
    bool r90 = false;                      // r06: Montgomery radix 2^90, reduced word by word from the split accumulators (pha_arith.h mont_redc90_split)
    // which carry-free split the constants in mat30 were cut for (0: none): 1 = inputs and matrix entries in 30-bit halves
    // (<= 16 inputs, every prime <= 60 bits); 2 = inputs cut at 30 bits, entries at 31 (<= 32 inputs of <= 60 bits, outputs
    // <= 62 bits: Q -> Bsk / Q -> R of the BFV multiply, whose auxiliary primes are 61 bits wide); 3 = the mirror (31, 30)
    int split_kind = 0;
    uint32_t row_pad = kBcRowPad;
    DevBuf<uint32_t> d_iprime, d_oprime;
};

// ---- DRNSTool of one level (src/rns.cu:11-200), hot-path constants only ----------------------
struct Tool {
    uint32_t size_ql = 0, size_qlp = 0, alpha = 0, beta = 0;
    std::vector<uint32_t> qlp_prime;  // limb of [Ql || P] -> index in the QP table
    DevBuf<uint32_t> d_qlp_prime;
    DevBuf<u64> part_hat_inv, part_hat_inv_shoup;  // partQlHatInv_mod_Ql_concat (rns.cu:152-182)
    std::vector<BConv> digit;                      // part Ql -> complement of QlP, per digit
    BConv p_to_ql;                                 // base_P_to_Ql_conv (rns.cu:196-198)
    BConv p_to_ql_pinv;                            // the same with every output row pre-multiplied by P^-1 mod q_j
    DevBuf<BConvDev> d_digit_convs, d_p_to_ql_conv, d_p_to_ql_pinv_conv; // device descriptors used by the batched launches
    bool split_ok = false;                         // every prime <= 60 bits: carry-free split MAC is valid
    int modup_split = 0, moddown_split = 0;        // BConv::split_kind of the digit converters (if common) and of P -> Ql
    DevBuf<u64> pinv, pinv_shoup;                  // bigPInv_mod_q (rns.cu:110-123)
    DevBuf<u64> inv_q_last, inv_q_last_shoup;      // rns.cu:66-80
    DevBuf<u64x2> inv_q_last2;                     // same, interleaved
    DevBuf<u64x2> pinv2;
    DevBuf<u64> p_hat_inv_by_limb, p_hat_inv_by_limb_shoup;  // [QlP]: phat_i^-1 mod p_i at limb Ql + i, 1 elsewhere
    // plain-modulus constants of the BGV branches (rns.cu:196-285); built when the context has a plain modulus
    bool bgv_ready = false;
    DModulus t_mod{};
    u64x2 inv_q_last_mod_t{}, pinv_mod_t{};        // (value, Shoup) modulo t
    u64x2 neg_ql_mod_t{};                          // t - (Ql mod t) (rns.cu:292-306), BFV add/sub plain
    DevBuf<u64> t_inv_mod_q, t_inv_mod_q_shoup;    // [Ql] t^-1 mod q_i (rns.cu:308-324)
    DevBuf<u64> plain_upper_half_increment;        // [Ql] q_i - t (context.cu: plain lift of multiply_plain)
    DevBuf<u64x2> q_last_mod_q2, p_mod_q2;         // [ql-1], [ql]  (value, Shoup) modulo q_i
    DevBuf<u64> p_hat_mod_t;                       // [alpha] row of base_P_to_t_conv
};

// ---- BFV BEHZ multiply at the top data level (src/rns.cu:392-560): auxiliary base Bsk = B u {m_sk}, m_tilde = 2^32.
//      The auxiliary moduli live in rows [aux0, aux0 + size_bsk] of the context's prime / table arrays
//      (B primes, m_sk, then m_tilde, which has a modulus entry but no NTT table).
struct Behz {
    uint32_t size_q = 0, size_b = 0, size_bsk = 0, aux0 = 0;
    u64 plain_t = 0, m_sk = 0;
    BConv q_to_bskmt, q_to_bsk, b_to_q, b_to_msk;      // Q -> Bsk u {m_tilde}; Q -> Bsk; B -> Q; B -> {m_sk}
    DevBuf<BConvDev> d_q_to_bskmt, d_q_to_bsk, d_b_to_q, d_b_to_msk;
    DevBuf<u64x2> mt_qhatinv;                            // [Q]   m_tilde * qhat_i^-1 mod q_i
    DevBuf<u64x2> inv_prod_q_mod_bsk, inv_mt_mod_bsk;    // [Bsk]
    DevBuf<u64> prod_q_mod_bsk;                          // [Bsk]
    DevBuf<u64> prod_b_mod_q;                            // [Q]
    DevBuf<u64> t_q, t_q_shoup, t_bsk, t_bsk_shoup;      // t with its Shoup quotient per limb (iNTT scale)
    u64x2 neg_inv_prod_q_mod_mt{}, inv_prod_b_mod_msk{};
};

// ---- BFV HPS multiply (mul_tech hps) at the top data level (src/rns.cu:687-790): auxiliary base R = |Q| + 1 primes
//      below the smallest q_i, in table rows [aux0, aux0 + size_r)
struct Hps {
    uint32_t size_q = 0, size_r = 0, aux0 = 0;
    BConv q_to_r, r_to_q;
    DevBuf<BConvDev> d_q_to_r, d_r_to_q;
    DevBuf<double> q_inv, r_inv;                 // 1 / q_i, 1 / r_j (src/host/rns.cu:321-323)
    DevBuf<u64> alpha_q_mod_r, alpha_r_mod_q;    // [|ibase| + 1][|obase|] alpha * prod(ibase) mod p_j (:459-466)
    DevBuf<double> frac;                         // [Q]        tRSHatInvModsDivsFrac
    DevBuf<u64> div_mod_r;                       // [R][Q + 1] tRSHatInvModsDivsModr
};

// ---- BFV HPS-over-Q multiply (mul_tech hps_overq, no levels dropped; src/rns.cu:792-885): base Rl = the first |Q|
//      primes below the smallest q_i = the first |Q| rows of the HPS base R
struct HpsQ {
    uint32_t size_q = 0, size_r = 0, aux0 = 0;
    BConv q_to_r, r_to_q, q_to_r_var1;           // exact Q -> Rl, exact Rl -> Q, quotient-style Q -> Rl (bConv_BEHZ_var1)
    DevBuf<BConvDev> d_q_to_r, d_r_to_q, d_q_to_r_var1;
    DevBuf<double> q_inv, r_inv;
    DevBuf<u64> alpha_q_mod_r, alpha_r_mod_q;
    DevBuf<double> frac;                         // [Rl]        tQlSlHatInvModsDivsFrac
    DevBuf<u64> div_mod_q;                       // [Q][Rl + 1] tQlSlHatInvModsDivsModq
    // hps_overq_leveled with levels dropped (src/rns.cu:897-975): size_q above is |Ql|; q_to_r_var1 then converts from the
    // FULL base Q (base_Q_to_Rl_conv)
    uint32_t size_q_full = 0, drop = 0;
    DevBuf<double> frac_drop;                    // [drop]          QlQHatInvModqDivqFrac
    DevBuf<u64> div_mod_q_drop;                  // [Ql][drop + 1]  QlQHatInvModqDivqModq
    DevBuf<u64> drop_mod_q, drop_mod_q_shoup;    // [Ql]            product of the dropped primes mod q_i
};

// ---- scratch arena: one per (context, stream), grown on demand, never freed until destroy ----
struct Arena {
    DevBuf<u64> buf;
    u64 *get(size_t words) {
        if (buf.count < words) buf.alloc(words);
        return buf.p;
    }
};

// two internal streams that a batched entry point may spread independent sub-batches over (forked from and joined back into the
// caller's stream with events; created on first use)
struct Lanes {
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    std::mutex mu;                                 // one batched call at a time enqueues on them
    void ensure();
    ~Lanes();
};

struct Context {
    int device = 0;
    int num_cus = 256;
    uint32_t log_n = 0, size_qp = 0, size_p = 0, size_q = 0;
    size_t n = 0;
    u64 plain_t = 0;  // 0 = none (ckks); set by pha_context_set_plain_modulus
    std::vector<u64> primes, roots, n_inv;
    std::vector<DModulus> mods;
    // device tables
    DevBuf<DModulus> d_mod;   // [size_qp]
    DevBuf<u64x2> d_tw;       // [size_qp][n] forward (psi^brev(k), Shoup)
    DevBuf<u64x2> d_itw;      // [size_qp][n] inverse (psi^-brev(k), Shoup), slot 1 NOT folded
    DevBuf<u64x2> d_ninv;     // [size_qp] (N^-1, Shoup)
    DevBuf<u64x2> d_w1ninv;   // [size_qp] (itw[1] * N^-1, Shoup)
    // FP64 path for primes below 2^50: the same tables as doubles W (bit patterns, 8 bytes per entry), rows of the
    // other primes are zero; d_fpinfo[prime] = (q, 1/q, usable flag)
    DevBuf<u64> d_twf, d_itwf;        // [size_qp][n] doubles W (bit patterns)
    DevBuf<u64x2> d_ninvf, d_w1ninvf;  // .x = the constant as a double
    DevBuf<FpInfo> d_fpinfo;
    // host copies kept for pha_context_download_twiddle and tool construction
    std::mutex mu;
    std::map<uint32_t, std::unique_ptr<Tool>> tools;
    std::unique_ptr<Behz> behz_tool;
    std::unique_ptr<Hps> hps_tool;
    std::map<uint32_t, std::unique_ptr<HpsQ>> hpsq_tools;   // by |Ql| (|Q| = plain hps_overq)
    uint32_t rows = 0;  // table rows = size_qp + auxiliary moduli
    // keyed by (stream handle, host thread): hipStreamPerThread and the null stream are sentinels that name a different
    // real stream (or none) in every host thread, so those two handles get one arena per calling thread
    typedef std::pair<void *, size_t> ArenaKey;
    static ArenaKey arena_key(void *stream);
    void release_thread_arenas(size_t thread_number, std::vector<std::unique_ptr<Arena>> &dead);   // a host thread exits (pha_context.hip: ThreadReaper)
    std::map<ArenaKey, std::unique_ptr<Arena>> arenas;
    std::map<ArenaKey, std::unique_ptr<Arena>> outer_arenas;   // for entry points that call other entry points (which use `arenas`)
    u64 *scratch_outer(void *stream, size_t words);
    std::map<uint32_t, DevBuf<uint32_t>> galois_tables;  // NTT-domain permutation per galois_elt
    Lanes lanes;

    Tool &tool(uint32_t size_ql);
    Behz &behz();                                             // built on first use; needs the plain modulus
    Hps &hps();
    HpsQ &hps_overq(uint32_t size_ql = 0);
    uint32_t add_aux_moduli(const std::vector<u64> &ntt_primes, u64 plain_modulus_like);
    u64 *scratch(void *stream, size_t words);
    // hand-off counters of the one-launch NTT (pha_ntt.hip): 16 header words + two words per (polynomial, limb) unit, zero
    // between launches (the kernel cleans up after itself); one array per (stream, thread) like the scratch arenas
    // (allocated once with the context, so that a launch never allocates: it may be inside a stream capture)
    static constexpr size_t kFlagUnits = 65536, kFlagArenas = 16;
    DevBuf<uint32_t> flag_pool;                    // [kFlagArenas][16 + 2 kFlagUnits]
    std::map<ArenaKey, uint32_t> flag_arenas;      // (stream, thread) -> arena of the pool
    uint32_t *ntt_flags(void *stream, size_t units);   // null: no arena left or too many units (caller takes two launches)
    // do all workgroups b of a 1-D grid with the same b % 8 run on one XCD?  (observed placement, checked once per
    // context by a census launch and again inside every one-launch NTT, which needs it and is disabled without it)
    bool xcd_round_robin = false, xcd_checked = false;
    bool xcd_placement_round_robin();              // (experiments library) census + counter pool on first use
    const uint32_t *galois_table(uint32_t elt);
};

// host number theory (independent of oracle/)
u64 h_mulmod(u64 a, u64 b, u64 q);
u64 h_powmod(u64 a, u64 e, u64 q);
u64 h_invmod(u64 a, u64 q);
u64 h_shoup(u64 w, u64 q);
bool h_is_prime(u64 n);
u64 h_minimal_primitive_root(u64 degree, u64 q);
DModulus h_modulus(u64 q);
uint32_t h_brev(uint32_t x, int bits);

// ---- launch descriptors shared by the .hip files ---------------------------------------------
struct LimbSel {
    uint32_t start;       // first processed limb (absolute index in the buffer)
    uint32_t count;       // number of processed limbs
    uint32_t remap_from;  // limbs >= remap_from use table row limb + remap_add (special primes)
    uint32_t remap_add;
    uint32_t excl_start, excl_end;  // limbs in [excl_start, excl_end) are skipped
};
inline LimbSel plain_sel(size_t start, size_t count) {
    return LimbSel{(uint32_t)start, (uint32_t)count, 0xffffffffu, 0, 0, 0};
}
// twr_idx2 of src/ntt/fntt_2d.cu:434-437
inline LimbSel special_sel(size_t start, size_t count, size_t size_QP, size_t size_P) {
    LimbSel s = plain_sel(start, count);
    s.remap_from = (uint32_t)(start + count - size_P);
    s.remap_add = (uint32_t)(size_QP - (start + count));
    return s;
}

// pha_check.hip: the canonical-operand precondition, countable; strict_* are no-ops unless strict mode is on (PHA_STRICT=1 / pha_set_strict)
struct Context;
struct RowMap {          // limb y of a buffer (relative to the pointer handed over) -> row of the prime table
    uint32_t row0, count, tail_from, tail_row0;   // y < tail_from: row0 + y; else tail_row0 + (y - tail_from)
};
inline RowMap rows_plain(size_t row0, size_t count) { return RowMap{(uint32_t)row0, (uint32_t)count, 0xffffffffu, 0}; }
// a [Q_l || P] buffer: size_ql data rows, then the special rows [size_q, size_q + size_p)
inline RowMap rows_qlp(size_t size_ql, size_t size_q, size_t size_p) {
    return RowMap{0, (uint32_t)(size_ql + size_p), (uint32_t)size_ql, (uint32_t)size_q};
}
bool strict_mode();
u64 count_noncanonical(Context &c, const u64 *data, const RowMap &rows, uint32_t polys, size_t poly_stride, hipStream_t s);
u64 count_noncanonical_keys(Context &c, const u64 *const *keys, uint32_t n_keys, uint32_t size_Ql, hipStream_t s);
void strict_operand(Context &c, const char *what, const u64 *data, const RowMap &rows, uint32_t polys, size_t poly_stride, hipStream_t s);
void strict_keys(Context &c, const char *what, const u64 *const *keys, uint32_t n_keys, uint32_t size_Ql, hipStream_t s);

// NTT drivers (pha_ntt.hip): pass 1 reads `in` and writes `mid`, pass 2 reads `mid` and writes `out`
struct NttExtra {
    const u64 *scale = nullptr, *scale_shoup = nullptr;  // indexed by absolute limb
    const u64 *aux = nullptr;                            // fuse_moddown: cx base
    const u64 *aux2 = nullptr;                           // EPI_FWD_KSRESCALE: ct base, with its own polynomial stride
    const u64 *scale2 = nullptr, *scale2_shoup = nullptr; //   and PInv per limb
    size_t aux2_stride = 0;
    uint32_t batch = 1;                                  // polynomials per launch (blockIdx.z)
    size_t poly_stride = 0;                              // elements between consecutive polynomials of in / mid
    size_t out_stride = 0, aux_stride = 0;               // same for out and aux (0 = poly_stride)
    size_t in_stride = 0;                                // polynomials of `in` when it is strided differently from mid (0 = poly_stride)
    bool first_pass_only = false;                        // forward: stop after the strided pass (the caller runs a fused second pass)
    bool second_pass_only = false;                       // inverse: the contiguous pass already ran (folded into the fused mod-up, ModupIpArgs::inv_from)
    bool first_pass_done = false;                        // forward: the strided pass already ran (r05: fused into the mod-up's base conversion, modup_conv_strided)
    const u64 *pro_src = nullptr;                        // forward only: every limb of polynomial z transforms
    size_t pro_stride = 0;                               //   (pro_src + z * pro_stride) mod its own prime; `in` unused
    uint32_t excl_step = 0;                              // polynomial z skips [excl_start + z*step, min(+len, limit))
    uint32_t excl_limit = 0xffffffffu;
    uint32_t excl_mod = 0;                               // != 0: the skipped range follows z % excl_mod (batches of ciphertexts)
};
void ntt_forward(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s);
void ntt_inverse(Context &c, const u64 *in, u64 *mid, u64 *out, const LimbSel &sel, int epi, const NttExtra &x,
                 hipStream_t s);
// The mod-up's forward transform with the key inner product as the epilogue of its contiguous pass (pha_ntt.hip): digits
// [beta][QlP][N] hold the converted limbs in coefficient form (own limbs untouched); cx [2][QlP][N] = sum_b NTT(digit_b) * key_b,
// with digit b's own limbs taken from `own` (the NTT-form input) instead.  Returns false when the shape has no fused form (the
// caller then runs the transform and the inner product separately).
struct ModupIpArgs {
    u64 *cx;
    const u64 *own;                   // c2 in NTT form [Ql][N], or null: no digit has untransformed own limbs (coefficient-form input)
    const u64 *const *evks;           // device array [beta] of keys [2][QP][N]
    size_t qlp_n, qp_n;
    uint32_t fix_limb;                // pha_keyswitch_rescale: limb whose cx receives ct + cx * P^-1 (0xffffffff = off)
    u64x2 fix_cst;
    const u64 *fix_ct;
    size_t fix_ct_stride;
    // r04: the limbs the mod-down transforms back first (the special limbs, and `inv_lead` -- the last data limb -- for the fused
    // rescale) leave the kernel with the inverse transform's contiguous pass already applied to both sums: the wavefront holds the
    // whole rows that pass works on, in its first round's register layout.  The caller then runs the inverse with
    // NttExtra::second_pass_only.  inv_from: first such limb, inv_lead: the extra one (limb indices in the [Ql | P] buffer, as
    // fix_limb; 0xffffffff = off / none).
    uint32_t inv_from = 0xffffffffu, inv_lead = 0xffffffffu;
    const u64x2 *itw = nullptr;       // inverse twiddle tables [prime][n] (integer / FP64 back end)
    const u64 *itwf = nullptr;
};
bool modup_ntt_inner_prod(Context &c, u64 *digits, const LimbSel &sel, const NttExtra &x, uint32_t beta, const ModupIpArgs &ip,
                          hipStream_t s);

// r05: the mod-up's base conversion fused with the STRIDED pass of the forward transform of the converted limbs ("from the
// conversion side", pha_ntt.hip): a workgroup owns a tile of T1 rows x V adjacent columns of one digit, reads the digit's (scaled,
// coefficient-form) input limbs for those coefficients ONCE, and for every output limb of the digit converts straight into the
// registers the strided pass starts from, runs the pass and stores its output -- the converted digits are never written or read back
// in coefficient form (src/rns_bconv.cu:455-485 + the first half of src/ntt/ntt_modup.cu:395-657).  `digits` [batch][QlP][N]
// receives what ntt_forward(..., first_pass_only) would have left there; the caller then runs the transform with
// NttExtra::first_pass_done.  Returns false (nothing launched) when the shape has no fused form.
struct ModupConvArgs {
    const BConvDev *convs;            // device array [beta]: the digit converters (Tool::d_digit_convs)
    uint32_t beta;                    // polynomial z of the launch = digit z % beta of ciphertext z / beta
    const u64 *src;                   // [ciphertexts][Ql][N]: c2 in coefficient form, scaled by partQlHatInv (the inverse transform's epilogue)
    size_t src_group_stride;
    const u64 *own;                   // NTT-form c2 for the verbatim copy of each digit's own limbs (modup_copy_partQl_kernel), or null
    size_t own_group_stride;
    uint32_t max_isz, max_osz;        // over the converters
    bool mont_split;                  // every converter: 30 / 30 cuts, Montgomery rows (BConv::split_kind 1 with oninv)
    // r06, rescale form (beta = 1, ONE converter P -> Ql whose last row is the last data limb's; `src` = cx, the converter's src_limb
    // points at its P limbs): the coefficient-form last data limb of polynomial z at resc_last + z * src_group_stride; the kernel
    // forms v_j = convert(row j) + (c_last mod q_j), j < osz - 1 (bconv_rescale_kernel's arithmetic).  null = the mod-up form
    const u64 *resc_last = nullptr;
    bool moddown = false;             // r06, plain mod-down form: as the rescale form without the last-limb fold (every row is an output limb)
};
bool modup_conv_strided(Context &c, u64 *digits, const LimbSel &sel, const NttExtra &x, const ModupConvArgs &m, hipStream_t s);

// shared launchers (pha_rns.hip / pha_poly.hip)
// optional epilogue of a conversion: store dst_j (+)= (cx_j - converted_j) * cst_j instead of converted_j (BFV mod-down)
struct BConvEpilogue {
    const u64 *cx;            // BFV mod-down: dst (+)= (cx - converted) * cst; null = off
    u64 *dst;
    const u64x2 *cst;
    size_t cx_stride, dst_stride;
    bool accumulate;
    // bConv_HPS (rns_bconv.cu:248-372): converted_j -= alpha_mod[v][j], v = round(sum_i y_i / q_i) from the scaled inputs the
    // kernel already holds; null = off
    const double *hps_inv = nullptr;      // [isz] 1 / q_i
    const u64 *hps_alpha = nullptr;       // [isz + 1][osz]
};
void launch_bconv(Context &c, const BConvDev *convs, uint32_t conv_step, uint32_t batch, uint32_t max_isz,
                  uint32_t max_osz, int split_kind, u64 *dst, size_t dst_stride, const u64 *src, size_t src_stride,
                  const u64 *own, bool scale_in, hipStream_t s, uint32_t conv_count = 0, size_t group_stride = 0,
                  const BConvEpilogue *epi = nullptr, size_t own_group_stride = 0);
// converter constants for arbitrary bases given as rows of the context's prime table (pha_context.hip)
void build_bconv(Context &c, BConv &b, const std::vector<uint32_t> &ip, const std::vector<uint32_t> &op,
                 const std::vector<u64> *out_scale = nullptr, uint32_t family_isz = 0);
void build_bconv_var1(Context &c, BConv &b, const std::vector<uint32_t> &ip, const std::vector<uint32_t> &op);
void describe_conv(const BConv &b, DevBuf<BConvDev> &out);
void launch_tensor(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, bool square,
                   hipStream_t s, size_t poly_limbs = 0);
void launch_add(Context &c, const u64 *a, const u64 *b, u64 *r, size_t limbs, size_t mod_start, hipStream_t s);

void register_context(Context *c, bool alive);   // live-context list walked by the per-thread arena reaper

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline void check_launch() { PHA_HIP(hipGetLastError()); }

}  // namespace pha

struct pha_context {
    pha::Context c;
};
