// phantom.h -- C++ host mirror of the reference's hot-path API over the C ABI (include/phantom_amd.h).
//
// The reference is compiled C++ (CUDA); its boundary for the accelerated path is the set of public
// headers include/context.cuh, ciphertext.h, secretkey.h (key containers) and evaluate.cuh.  This header
// keeps those names, members, argument meaning, pre-condition checks and exception types for everything
// that sits on the RNS hot path, so code written against the reference's evaluate.* API compiles against
// it for: negate / add / sub / multiply (CKKS, BGV tensor) / relinearize / multiply_and_relin /
// rescale_to_next / mod_switch_to_next / apply_galois / rotate / hoisting / keyswitch_inplace.
// Out of scope here exactly as in SURVEY.md section 8: key generation, encryption, decryption, encoders,
// BFV BEHZ/HPS multiply (those callers are not on the accelerated path yet and throw).
//
// Header-only; needs the HIP runtime for device memory (hipMallocAsync / hipFreeAsync on the
// per-thread stream, like include/cuda_wrapper.cuh:65-189) and libphantom_amd.so for every kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <phantom_amd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <istream>
#include <limits>
#include <ostream>
#include <vector>

// the reference spells the stream type cudaStream_t in every signature
using cudaStream_t = hipStream_t;
#ifndef cudaStreamPerThread
#define cudaStreamPerThread hipStreamPerThread
#endif

namespace phantom {

enum class scheme_type : std::uint8_t { none = 0x0, bfv = 0x1, ckks = 0x2, bgv = 0x3 };  // encryptionparams.h:19-27
enum class mul_tech_type : std::uint8_t { none = 0x0, behz = 0x1, hps = 0x2, hps_overq = 0x3, hps_overq_leveled = 0x4 };

namespace util {

inline void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("CUDA Runtime Error: ") + hipGetErrorString(e) + " (" + what + ")");
}
inline void check_pha(int status) {  // C-ABI status -> the reference's exception types
    if (status == PHA_OK) return;
    if (status == PHA_ERR_INVALID_ARGUMENT) throw std::invalid_argument(pha_last_error());
    if (status == PHA_ERR_LOGIC) throw std::logic_error(pha_last_error());
    throw std::runtime_error(pha_last_error());
}

// cuda_stream_wrapper (include/cuda_wrapper.cuh:47-63): an owned non-blocking stream
class cuda_stream_wrapper {
    cudaStream_t stream_{};

public:
    cuda_stream_wrapper() { check_hip(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreateWithFlags"); }
    ~cuda_stream_wrapper() { (void)hipStreamDestroy(stream_); }
    cuda_stream_wrapper(const cuda_stream_wrapper &) = delete;
    cuda_stream_wrapper &operator=(const cuda_stream_wrapper &) = delete;
    [[nodiscard]] const cudaStream_t &get_stream() const { return stream_; }
};

// cuda_auto_ptr<T> (include/cuda_wrapper.cuh:65-189): stream-ordered RAII buffer; move steals, copy is a
// deep device copy on the source's stream.
template <class T>
class cuda_auto_ptr {
    T *ptr_ = nullptr;
    size_t n_ = 0;
    cudaStream_t stream_ = nullptr;

public:
    cuda_auto_ptr() = default;
    cuda_auto_ptr(size_t n, const cudaStream_t &stream) : n_(n), stream_(stream) {
        if (n) check_hip(hipMallocAsync(reinterpret_cast<void **>(&ptr_), n * sizeof(T), stream), "hipMallocAsync");
    }
    cuda_auto_ptr(const cuda_auto_ptr &o) : n_(o.n_), stream_(o.stream_) {
        if (n_) {
            check_hip(hipMallocAsync(reinterpret_cast<void **>(&ptr_), n_ * sizeof(T), stream_), "hipMallocAsync");
            check_hip(hipMemcpyAsync(ptr_, o.ptr_, n_ * sizeof(T), hipMemcpyDeviceToDevice, stream_), "hipMemcpyAsync");
        }
    }
    cuda_auto_ptr(cuda_auto_ptr &&o) noexcept : ptr_(o.ptr_), n_(o.n_), stream_(o.stream_) { o.ptr_ = nullptr; o.n_ = 0; }
    cuda_auto_ptr &operator=(const cuda_auto_ptr &o) {
        if (this != &o) { cuda_auto_ptr tmp(o); swap(tmp); }
        return *this;
    }
    cuda_auto_ptr &operator=(cuda_auto_ptr &&o) noexcept {
        if (this != &o) { reset(); ptr_ = o.ptr_; n_ = o.n_; stream_ = o.stream_; o.ptr_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~cuda_auto_ptr() { reset(); }
    void swap(cuda_auto_ptr &o) noexcept { std::swap(ptr_, o.ptr_); std::swap(n_, o.n_); std::swap(stream_, o.stream_); }
    void reset() {
        if (ptr_) (void)hipFreeAsync(ptr_, stream_);
        ptr_ = nullptr;
        n_ = 0;
    }
    // free on ANOTHER stream than the allocating one: ordered after whatever that stream still does with the buffer
    void release_on(const cudaStream_t &stream) {
        stream_ = stream;
        reset();
    }
    [[nodiscard]] T *get() const { return ptr_; }
    [[nodiscard]] size_t get_n() const { return n_; }
};
template <class T>
inline cuda_auto_ptr<T> make_cuda_auto_ptr(size_t n, const cudaStream_t &stream) { return cuda_auto_ptr<T>(n, stream); }

// get_elt_from_step (include/galois.cuh:16-49): generator 5, step 0 = conjugation (2N - 1)
[[nodiscard]] inline uint32_t get_elt_from_step(int step, size_t coeff_count) {
    const auto n = static_cast<uint32_t>(coeff_count);
    const uint32_t m32 = n * 2;
    const auto m = static_cast<uint64_t>(m32);
    if (step == 0) return static_cast<uint32_t>(m - 1);
    const bool sign = step < 0;
    auto pos_step = static_cast<uint32_t>(std::abs(step));
    if (pos_step >= (n >> 1)) throw std::invalid_argument("step count too large");
    pos_step &= m32 - 1;
    step = sign ? static_cast<int>(n >> 1) - static_cast<int>(pos_step) : static_cast<int>(pos_step);
    uint64_t galois_elt = 1;
    while (step--) {
        galois_elt *= 5;
        galois_elt &= m - 1;
    }
    return static_cast<uint32_t>(galois_elt);
}
[[nodiscard]] inline std::vector<uint32_t> get_elts_from_steps(const std::vector<int> &steps, size_t coeff_count) {
    std::vector<uint32_t> elts;
    for (int s : steps) elts.push_back(get_elt_from_step(s, coeff_count));
    return elts;
}
// non-adjacent form of a rotation step (src/host/numth.cu naf)
[[nodiscard]] inline std::vector<int> naf(int value) {
    std::vector<int> res;
    const bool sign = value < 0;
    value = std::abs(value);
    for (int i = 0; value; i++) {
        const int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) res.push_back((sign ? -zi : zi) * (1 << i));
    }
    return res;
}

}  // namespace util

namespace arith {

// Modulus (include/host/modulus.h): value + Barrett const_ratio floor(2^128 / value)
class Modulus {
    uint64_t value_ = 0;
    std::array<uint64_t, 3> const_ratio_{{0, 0, 0}};
    int bit_count_ = 0;

public:
    Modulus() = default;
    explicit Modulus(uint64_t value) : value_(value) {
        if (value >> 61) throw std::invalid_argument("value can be at most 61-bit");
        if (value == 1) throw std::invalid_argument("value can not be 1");
        if (value) {
            const unsigned __int128 r = (~static_cast<unsigned __int128>(0)) / value;
            const_ratio_ = {static_cast<uint64_t>(r), static_cast<uint64_t>(r >> 64),
                            static_cast<uint64_t>((~static_cast<unsigned __int128>(0)) - r * value + 1)};
            bit_count_ = 64 - __builtin_clzll(value);
        }
    }
    [[nodiscard]] uint64_t value() const { return value_; }
    [[nodiscard]] const std::array<uint64_t, 3> &const_ratio() const { return const_ratio_; }
    [[nodiscard]] int bit_count() const { return bit_count_; }
    [[nodiscard]] bool is_zero() const { return value_ == 0; }
};

// sec_level_type (include/host/modulus.h:217-232)
enum class sec_level_type : int { none = 0, tc128 = 128, tc192 = 192, tc256 = 256 };

namespace default_tables {
struct DefaultModuliRow {
    int level;
    size_t degree;
    int max_bit_count;
    std::vector<uint64_t> primes;
};
// the reference's literal tables (src/host/globals.cu:51-361, include/host/hestdparms.h), as data
inline const std::vector<DefaultModuliRow> &default_moduli() {
    static const std::vector<DefaultModuliRow> rows = {
#include "default_coeff_modulus.inc"
    };
    return rows;
}
}  // namespace default_tables

// CoeffModulus::Create (src/host/modulus.cu:82-111), computed by the library; BFVDefault / MaxBitCount
// (src/host/modulus.cu:57-80, include/host/modulus.h:244-276) from the reference's tables
struct CoeffModulus {
    [[nodiscard]] static int MaxBitCount(size_t poly_modulus_degree, sec_level_type sec_level = sec_level_type::tc128) {
        if (sec_level == sec_level_type::none) return (std::numeric_limits<int>::max)();
        for (const auto &r : default_tables::default_moduli())
            if (r.level == static_cast<int>(sec_level) && r.degree == poly_modulus_degree) return r.max_bit_count;
        return 0;
    }
    [[nodiscard]] static std::vector<Modulus> BFVDefault(size_t poly_modulus_degree, sec_level_type sec_level = sec_level_type::tc128) {
        if (!MaxBitCount(poly_modulus_degree, sec_level)) throw std::invalid_argument("non-standard poly_modulus_degree");
        if (sec_level == sec_level_type::none) throw std::invalid_argument("invalid security level");
        for (const auto &r : default_tables::default_moduli())
            if (r.level == static_cast<int>(sec_level) && r.degree == poly_modulus_degree) {
                std::vector<Modulus> out;
                for (uint64_t q : r.primes) out.emplace_back(q);
                return out;
            }
        throw std::runtime_error("invalid security level");
    }
    static std::vector<Modulus> Create(size_t poly_modulus_degree, const std::vector<int> &bit_sizes) {
        std::vector<uint64_t> v(bit_sizes.size());
        util::check_pha(pha_coeff_modulus_create(poly_modulus_degree, bit_sizes.data(), bit_sizes.size(), v.data()));
        std::vector<Modulus> out;
        for (uint64_t q : v) out.emplace_back(q);
        return out;
    }
};

// PlainModulus::Batching (include/host/modulus.h:301-320): a prime = 1 mod 2N of the given size
struct PlainModulus {
    [[nodiscard]] static Modulus Batching(size_t poly_modulus_degree, int bit_size) {
        return CoeffModulus::Create(poly_modulus_degree, {bit_size})[0];
    }
};

}  // namespace arith

// EncryptionParameters (include/host/encryptionparams.h:29-246), the members the hot path reads
class EncryptionParameters {
    scheme_type scheme_;
    mul_tech_type mul_tech_ = mul_tech_type::hps;
    size_t poly_modulus_degree_ = 0;
    std::vector<arith::Modulus> coeff_modulus_;
    size_t special_modulus_size_ = 1;  // default 1 (:235)
    arith::Modulus plain_modulus_;
    std::vector<uint32_t> galois_elts_;

public:
    explicit EncryptionParameters(scheme_type scheme = scheme_type::none) : scheme_(scheme) {}
    void set_poly_modulus_degree(size_t n) { poly_modulus_degree_ = n; }
    void set_coeff_modulus(const std::vector<arith::Modulus> &m) { coeff_modulus_ = m; }
    void set_special_modulus_size(size_t s) { special_modulus_size_ = s; }
    void set_plain_modulus(const arith::Modulus &t) { plain_modulus_ = t; }
    void set_mul_tech(mul_tech_type t) { mul_tech_ = t; }
    void set_galois_elts(const std::vector<uint32_t> &e) { galois_elts_ = e; }
    [[nodiscard]] scheme_type scheme() const { return scheme_; }
    [[nodiscard]] mul_tech_type mul_tech() const { return mul_tech_; }
    [[nodiscard]] size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    [[nodiscard]] const std::vector<arith::Modulus> &coeff_modulus() const { return coeff_modulus_; }
    [[nodiscard]] size_t special_modulus_size() const { return special_modulus_size_; }
    [[nodiscard]] const arith::Modulus &plain_modulus() const { return plain_modulus_; }
    [[nodiscard]] const std::vector<uint32_t> &galois_elts() const { return galois_elts_; }
};

}  // namespace phantom

// DModulus (include/ntt.cuh:6-32): value + Barrett ratio; here a host-side copy (the device table lives in the library)
class DModulus {
    uint64_t value_ = 0;
    uint64_t const_ratio_[2] = {0, 0};

public:
    DModulus() = default;
    DModulus(uint64_t value, uint64_t ratio0, uint64_t ratio1) : value_(value), const_ratio_{ratio0, ratio1} {}
    [[nodiscard]] uint64_t value() const { return value_; }
    [[nodiscard]] const uint64_t (&const_ratio() const)[2] { return const_ratio_; }
};

// DNTTTable (include/ntt.cuh:34-129): the NTT tables of the QP primes.  Here a thin handle on the library's context,
// which owns one table set per prime (own layout: interleaved (w, w') pairs, FP64 copies for primes below 2^50);
// the launchers of include/ntt.cuh take it exactly as in the reference: nwt_2d_radix8_forward_inplace(p, ctx.gpu_rns_tables(), ...).
class DNTTTable {
    pha_context_t ctx_ = nullptr;
    uint64_t n_ = 0, size_ = 0;

public:
    DNTTTable() = default;
    DNTTTable(pha_context_t ctx, uint64_t n, uint64_t size) : ctx_(ctx), n_(n), size_(size) {}
    [[nodiscard]] uint64_t n() const { return n_; }
    [[nodiscard]] uint64_t size() const { return size_; }
    [[nodiscard]] pha_context_t amd() const { return ctx_; }
    // host copies of what the reference keeps on the device (twiddle(): [size][n], which = 0 forward, 1 forward Shoup, 2 inverse, 3 inverse Shoup)
    [[nodiscard]] DModulus modulus(size_t index) const {
        uint64_t v = 0, ratio[2] = {0, 0}, root = 0, ninv = 0;
        phantom::util::check_pha(pha_context_prime_info(ctx_, static_cast<uint32_t>(index), &v, ratio, &root, &ninv));
        return DModulus(v, ratio[0], ratio[1]);
    }
    [[nodiscard]] std::vector<uint64_t> twiddle_row(size_t index, int which) const {
        std::vector<uint64_t> row(n_);
        phantom::util::check_pha(pha_context_download_twiddle(ctx_, static_cast<uint32_t>(index), which, row.data()));
        return row;
    }
};

// DBaseConverter (include/rns_bconv.cuh:3-87) between two bases given as rows of the context's prime table
class DBaseConverter {
    pha_base_converter_t conv_ = nullptr;

public:
    DBaseConverter() = default;
    DBaseConverter(pha_context_t ctx, const std::vector<uint32_t> &ibase, const std::vector<uint32_t> &obase) {
        phantom::util::check_pha(pha_base_converter_create(ctx, ibase.data(), ibase.size(), obase.data(), obase.size(), &conv_));
    }
    // a converter to ONE raw output modulus (base_q_to_t_conv_ of src/rns.cu:283-284: the plain modulus is no row of the table);
    // it serves exact_convert_array only
    DBaseConverter(pha_context_t ctx, const std::vector<uint32_t> &ibase, uint64_t out_modulus) {
        phantom::util::check_pha(pha_base_converter_create_modulus(ctx, ibase.data(), ibase.size(), out_modulus, &conv_));
    }
    DBaseConverter(const DBaseConverter &) = delete;
    DBaseConverter &operator=(const DBaseConverter &) = delete;
    DBaseConverter(DBaseConverter &&o) noexcept : conv_(o.conv_) { o.conv_ = nullptr; }
    DBaseConverter &operator=(DBaseConverter &&o) noexcept {
        if (this != &o) { if (conv_) pha_base_converter_destroy(conv_); conv_ = o.conv_; o.conv_ = nullptr; }
        return *this;
    }
    ~DBaseConverter() { if (conv_) pha_base_converter_destroy(conv_); }
    // (n = poly degree: the reference passes it along; the converter knows its context's)
    void bConv_BEHZ(uint64_t *dst, const uint64_t *src, size_t /*n*/, const cudaStream_t &stream) const {
        phantom::util::check_pha(pha_bConv_BEHZ(conv_, dst, src, stream));
    }
    void bConv_HPS(uint64_t *dst, const uint64_t *src, size_t /*n*/, const cudaStream_t &stream) const {
        phantom::util::check_pha(pha_bConv_HPS(conv_, dst, src, stream));
    }
    void bConv_BEHZ_var1(uint64_t *dst, const uint64_t *src, size_t /*n*/, const cudaStream_t &stream) const {   // rns_bconv.cuh:64
        phantom::util::check_pha(pha_bConv_BEHZ_var1(conv_, dst, src, stream));
    }
    void exact_convert_array(uint64_t *dst, const uint64_t *src, uint64_t /*poly_degree*/, const cudaStream_t &stream) const {   // :68
        phantom::util::check_pha(pha_exact_convert_array(conv_, dst, src, stream));
    }
};

namespace phantom {

// DRNSTool (include/rns.cuh:13-236) of one level: a handle (context, |Ql|) on the library's lazily built per-level
// constants; the hot-path methods with the reference's signatures (include/rns.cuh:156-205)
class DRNSTool {
    pha_context_t ctx_ = nullptr;
    size_t size_Ql_ = 0;

public:
    DRNSTool() = default;
    DRNSTool(pha_context_t ctx, size_t size_Ql) : ctx_(ctx), size_Ql_(size_Ql) {}
    [[nodiscard]] pha_context_t amd() const { return ctx_; }
    [[nodiscard]] size_t size_Ql() const { return size_Ql_; }
    [[nodiscard]] size_t v_base_part_Ql_to_compl_part_QlP_conv_size() const {   // beta (rns.cu:152)
        uint32_t beta = 0;
        util::check_pha(pha_tool_beta(ctx_, static_cast<uint32_t>(size_Ql_), &beta));
        return beta;
    }
    void modup(uint64_t *dst, const uint64_t *cks, const DNTTTable &, const scheme_type &scheme, const cudaStream_t &stream) const {
        util::check_pha(pha_modup(ctx_, size_Ql_, dst, cks, static_cast<int>(scheme), stream));
    }
    void moddown(uint64_t *ct_i, uint64_t *cx_i, const DNTTTable &, const scheme_type &scheme, const cudaStream_t &stream) const {   // rns.cuh:159-160
        util::check_pha(pha_moddown(ctx_, size_Ql_, ct_i, cx_i, static_cast<int>(scheme), stream));
    }
    void moddown_from_NTT(uint64_t *ct_i, uint64_t *cx_i, const DNTTTable &, const scheme_type &scheme, const cudaStream_t &stream) const {
        util::check_pha(pha_moddown_from_NTT(ctx_, size_Ql_, ct_i, cx_i, static_cast<int>(scheme), stream));
    }
    // the BFV multiply steps (include/rns.cuh:167-200; callers src/evaluate.cu:404-548, :674-818, :1014-1016)
    void scaleAndRound_HPS_QR_R(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_scaleAndRound_HPS_QR_R(ctx_, size_Ql_, dst, src, stream));
    }
    void scaleAndRound_HPS_QlRl_Ql(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_scaleAndRound_HPS_QlRl_Ql(ctx_, size_Ql_, dst, src, stream));
    }
    void ExpandCRTBasis_Ql_Q_add_to_ct(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_ExpandCRTBasis_Ql_Q_add_to_ct(ctx_, size_Ql_, dst, src, stream));
    }
    void fastbconv_m_tilde(uint64_t *dst, uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_fastbconv_m_tilde(ctx_, size_Ql_, dst, src, stream));
    }
    void sm_mrq(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_sm_mrq(ctx_, size_Ql_, dst, src, stream));
    }
    void fast_floor(uint64_t *input_base_q, uint64_t *input_base_Bsk, uint64_t *out_base_Bsk, const cudaStream_t &stream) const {
        util::check_pha(pha_fast_floor(ctx_, size_Ql_, input_base_q, input_base_Bsk, out_base_Bsk, stream));
    }
    void fastbconv_sk(uint64_t *input_base_Bsk, uint64_t *out_base_q, const cudaStream_t &stream) const {
        util::check_pha(pha_fastbconv_sk(ctx_, size_Ql_, input_base_Bsk, out_base_q, stream));
    }
    // sizes of the auxiliary bases of this level's tool (base_Bsk().size(), base_Rl().size() of include/rns.cuh)
    [[nodiscard]] size_t base_Bsk_size() const { uint32_t v = 0; util::check_pha(pha_tool_aux_sizes(ctx_, size_Ql_, &v, nullptr, nullptr)); return v; }
    [[nodiscard]] size_t base_R_size() const { uint32_t v = 0; util::check_pha(pha_tool_aux_sizes(ctx_, size_Ql_, nullptr, &v, nullptr)); return v; }
    [[nodiscard]] size_t base_Rl_size() const { uint32_t v = 0; util::check_pha(pha_tool_aux_sizes(ctx_, size_Ql_, nullptr, nullptr, &v)); return v; }
    void divide_and_round_q_last(const uint64_t *src, size_t cipher_size, uint64_t *dst, const cudaStream_t &stream) const {
        util::check_pha(pha_divide_and_round_q_last(ctx_, size_Ql_, src, cipher_size, dst, stream));
    }
    void divide_and_round_q_last_ntt(uint64_t *src, size_t cipher_size, const DNTTTable &, uint64_t *dst, const cudaStream_t &stream) const {
        util::check_pha(pha_divide_and_round_q_last_ntt(ctx_, size_Ql_, src, cipher_size, dst, stream));
    }
    void mod_t_and_divide_q_last_ntt(uint64_t *src, size_t cipher_size, const DNTTTable &, uint64_t *dst, const cudaStream_t &stream) const {
        util::check_pha(pha_mod_t_and_divide_q_last_ntt(ctx_, size_Ql_, src, cipher_size, dst, stream));
    }
    void scaleAndRound_HPS_Q_Ql(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_scaleAndRound_HPS_Q_Ql(ctx_, size_Ql_, dst, src, stream));
    }
    void ExpandCRTBasis_Ql_Q(uint64_t *dst, const uint64_t *src, const cudaStream_t &stream) const {
        util::check_pha(pha_ExpandCRTBasis_Ql_Q(ctx_, size_Ql_, dst, src, stream));
    }
};

// ContextData (include/context.cuh:19-131): the parameters of one level of the modulus chain and its DRNSTool
class ContextData {
    EncryptionParameters parms_;
    size_t chain_index_ = 0;
    DRNSTool gpu_rns_tool_;

public:
    ContextData(EncryptionParameters parms, size_t chain_index) : parms_(std::move(parms)), chain_index_(chain_index) {}
    void bind_tool(pha_context_t ctx, size_t size_Ql) { gpu_rns_tool_ = DRNSTool(ctx, size_Ql); }
    [[nodiscard]] const EncryptionParameters &parms() const { return parms_; }
    [[nodiscard]] size_t chain_index() const { return chain_index_; }
    [[nodiscard]] const DRNSTool &gpu_rns_tool() const noexcept { return gpu_rns_tool_; }
};

}  // namespace phantom

// PhantomContext (include/context.cuh:133-273, src/context.cu:121-232): the modulus chain
// (context_data_[0] = key level QP, [1] = Q, [1 + l] drops l primes) plus the device tables, which
// live behind one pha_context_t (tables for every QP prime + a lazily built DRNSTool per level).
class PhantomContext {
    pha_context_t amd_ = nullptr;
    std::vector<phantom::ContextData> context_data_;
    bool using_keyswitching_ = false;
    size_t first_parm_index_ = 0;
    size_t poly_degree_ = 0;
    size_t coeff_mod_size_ = 0;
    DNTTTable gpu_rns_tables_;

public:
    explicit PhantomContext(const phantom::EncryptionParameters &params, int device = -1) {
        using namespace phantom;
        const auto &qp = params.coeff_modulus();
        const size_t n = params.poly_modulus_degree();
        if (n == 0 || (n & (n - 1))) throw std::invalid_argument("poly_modulus_degree is invalid");
        if (qp.empty()) throw std::invalid_argument("coeff_modulus is empty");
        const size_t size_p = params.special_modulus_size();
        if (size_p >= qp.size() && qp.size() > 1) throw std::invalid_argument("special_modulus_size is invalid");
        poly_degree_ = n;
        coeff_mod_size_ = qp.size();
        // a single-prime chain has no special prime and no key switching (src/context.cu:142-168)
        using_keyswitching_ = qp.size() > 1 && size_p > 0;
        const size_t sp = using_keyswitching_ ? size_p : 0;
        context_data_.emplace_back(params, 0);
        if (using_keyswitching_) {
            const size_t size_q = qp.size() - sp;
            for (size_t drop = 0; drop < size_q; drop++) {
                EncryptionParameters p = params;
                p.set_coeff_modulus(std::vector<arith::Modulus>(qp.begin(), qp.begin() + (size_q - drop)));
                context_data_.emplace_back(p, 1 + drop);
            }
            first_parm_index_ = 1;
        }
        if (device < 0) util::check_hip(hipGetDevice(&device), "hipGetDevice");
        std::vector<uint64_t> primes;
        for (const auto &m : qp) primes.push_back(m.value());
        int log_n = 0;
        while ((size_t(1) << log_n) < n) log_n++;
        util::check_pha(pha_context_create(&amd_, log_n, primes.data(), primes.size(), sp, device));
        // DRNSTool receives the plain modulus for BFV / BGV (src/context.cu:200-216 -> src/rns.cu:196-285)
        if ((params.scheme() == scheme_type::bgv || params.scheme() == scheme_type::bfv) && params.plain_modulus().value() != 0)
            util::check_pha(pha_context_set_plain_modulus(amd_, params.plain_modulus().value()));
        gpu_rns_tables_ = DNTTTable(amd_, n, qp.size());
        const size_t size_q_all = qp.size() - sp;
        for (auto &cd : context_data_)   // index 0 (key level) and 1 share |Q|; 1 + l drops l primes (context.cu:186-229)
            cd.bind_tool(amd_, cd.chain_index() <= 1 ? size_q_all : size_q_all - (cd.chain_index() - 1));
    }
    PhantomContext(const PhantomContext &) = delete;
    PhantomContext &operator=(const PhantomContext &) = delete;
    ~PhantomContext() { pha_context_destroy(amd_); }

    [[nodiscard]] pha_context_t amd() const { return amd_; }
    [[nodiscard]] const DNTTTable &gpu_rns_tables() const noexcept { return gpu_rns_tables_; }
    [[nodiscard]] const phantom::DRNSTool &get_context_data_rns_tool(size_t index) const { return get_context_data(index).gpu_rns_tool(); }
    [[nodiscard]] const phantom::ContextData &get_context_data(size_t index) const {
        if (index >= context_data_.size()) throw std::invalid_argument("index is out of range");
        return context_data_[index];
    }
    [[nodiscard]] size_t total_parm_size() const { return context_data_.size(); }
    [[nodiscard]] bool using_keyswitching() const { return using_keyswitching_; }
    [[nodiscard]] size_t get_first_index() const { return first_parm_index_; }
    [[nodiscard]] size_t get_next_index(size_t index) const {
        if (index + 1 >= context_data_.size()) throw std::invalid_argument("no next parameters");
        return index + 1;
    }
    [[nodiscard]] size_t poly_degree() const { return poly_degree_; }
    [[nodiscard]] size_t coeff_mod_size() const { return coeff_mod_size_; }
};

// PhantomCiphertext (include/ciphertext.h:7-171): metadata + one device buffer [poly][limb][coeff]
class PhantomCiphertext {
    size_t chain_index_ = 0, size_ = 0, poly_modulus_degree_ = 0, coeff_modulus_size_ = 0;
    double scale_ = 1.0;
    uint64_t correction_factor_ = 1;
    size_t noiseScaleDeg_ = 1;
    bool is_ntt_form_ = true, is_asymmetric_ = false;
    phantom::util::cuda_auto_ptr<uint64_t> data_;

public:
    void resize(const PhantomContext &context, size_t chain_index, size_t size, const cudaStream_t &stream) {
        const auto &parms = context.get_context_data(chain_index).parms();
        resize(size, parms.coeff_modulus().size(), parms.poly_modulus_degree(), stream);
        chain_index_ = chain_index;
    }
    // when the size changes the previous data is copied (ciphertext.h:44-72)
    void resize(size_t size, size_t coeff_modulus_size, size_t poly_modulus_degree, const cudaStream_t &stream) {
        const size_t old_size = size_ * coeff_modulus_size_ * poly_modulus_degree_;
        const size_t new_size = size * coeff_modulus_size * poly_modulus_degree;
        if (new_size == 0) {
            data_.reset();
            return;
        }
        if (new_size != old_size) {
            auto prev(std::move(data_));
            data_ = phantom::util::make_cuda_auto_ptr<uint64_t>(new_size, stream);
            const size_t copy = std::min(old_size, new_size);
            if (copy)
                phantom::util::check_hip(hipMemcpyAsync(data_.get(), prev.get(), copy * sizeof(uint64_t),
                                                        hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
        }
        size_ = size;
        coeff_modulus_size_ = coeff_modulus_size;
        poly_modulus_degree_ = poly_modulus_degree;
    }
    void set_scale(double scale) { scale_ = scale; }
    void set_chain_index(size_t i) { chain_index_ = i; }
    void set_ntt_form(bool f) { is_ntt_form_ = f; }
    void set_correction_factor(uint64_t c) { correction_factor_ = c; }
    void SetNoiseScaleDeg(size_t d) { noiseScaleDeg_ = d; }
    [[nodiscard]] size_t GetNoiseScaleDeg() const { return noiseScaleDeg_; }
    [[nodiscard]] bool is_asymmetric() const { return is_asymmetric_; }
    [[nodiscard]] bool is_ntt_form() const { return is_ntt_form_; }
    [[nodiscard]] size_t chain_index() const { return chain_index_; }
    [[nodiscard]] size_t size() const { return size_; }
    [[nodiscard]] size_t poly_modulus_degree() const { return poly_modulus_degree_; }
    [[nodiscard]] size_t coeff_modulus_size() const { return coeff_modulus_size_; }
    [[nodiscard]] double scale() const { return scale_; }
    [[nodiscard]] uint64_t correction_factor() const { return correction_factor_; }
    [[nodiscard]] uint64_t *data() const { return data_.get(); }
    // take over a device buffer of the same shape: lets an out-of-place kernel produce the new contents without a copy back.
    // The old buffer is released on `stream`, i.e. after the kernels already enqueued there that still read it (it may have
    // been allocated on another stream).  NOTE: data() changes -- unlike the reference's apply_galois_inplace, which copies
    // back into the same buffer (src/evaluate.cu:1597-1622), pointers obtained from data() before a rotation are stale after it.
    void replace_data(phantom::util::cuda_auto_ptr<uint64_t> &&fresh, const cudaStream_t &stream) {
        data_.swap(fresh);
        fresh.release_on(stream);
    }

    // On-disk format of include/ciphertext.h:173-214: the nine metadata fields as raw host-endian values, then
    // size * coeff_modulus_size * poly_modulus_degree words.  Files are interchangeable with the reference's.
    void save(std::ostream &stream) const {
        stream.write(reinterpret_cast<const char *>(&chain_index_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&size_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&poly_modulus_degree_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&coeff_modulus_size_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&scale_), sizeof(double));
        stream.write(reinterpret_cast<const char *>(&correction_factor_), sizeof(std::uint64_t));
        stream.write(reinterpret_cast<const char *>(&noiseScaleDeg_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&is_ntt_form_), sizeof(bool));
        stream.write(reinterpret_cast<const char *>(&is_asymmetric_), sizeof(bool));
        std::vector<uint64_t> host(size_ * coeff_modulus_size_ * poly_modulus_degree_);
        if (!host.empty()) store_to_host(host.data());
        stream.write(reinterpret_cast<const char *>(host.data()), static_cast<std::streamsize>(host.size() * sizeof(uint64_t)));
    }
    void load(std::istream &stream) {
        size_t size = 0, degree = 0, limbs = 0;
        stream.read(reinterpret_cast<char *>(&chain_index_), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&size), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&degree), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&limbs), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&scale_), sizeof(double));
        stream.read(reinterpret_cast<char *>(&correction_factor_), sizeof(std::uint64_t));
        stream.read(reinterpret_cast<char *>(&noiseScaleDeg_), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&is_ntt_form_), sizeof(bool));
        stream.read(reinterpret_cast<char *>(&is_asymmetric_), sizeof(bool));
        if (!stream || size > 16 || limbs > 4096 || degree > (size_t(1) << 17))
            throw std::invalid_argument("ciphertext stream is not valid");
        std::vector<uint64_t> host(size * limbs * degree);
        stream.read(reinterpret_cast<char *>(host.data()), static_cast<std::streamsize>(host.size() * sizeof(uint64_t)));
        if (!stream) throw std::invalid_argument("ciphertext stream is truncated");
        const auto &s = cudaStreamPerThread;
        data_.reset();
        size_ = coeff_modulus_size_ = poly_modulus_degree_ = 0;
        resize(size, limbs, degree, s);
        if (!host.empty()) {
            phantom::util::check_hip(hipMemcpyAsync(data_.get(), host.data(), host.size() * 8, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
            phantom::util::check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        }
    }

    // test / interop helpers (the reference fills ciphertexts by encryption, which is out of scope)
    void load_from_host(const PhantomContext &context, size_t chain_index, size_t size, const uint64_t *host,
                        const cudaStream_t &stream = cudaStreamPerThread) {
        resize(context, chain_index, size, stream);
        phantom::util::check_hip(hipMemcpyAsync(data_.get(), host, size_ * coeff_modulus_size_ * poly_modulus_degree_ * 8,
                                                hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    }
    void store_to_host(uint64_t *host, const cudaStream_t &stream = cudaStreamPerThread) const {
        phantom::util::check_hip(hipMemcpyAsync(host, data_.get(), size_ * coeff_modulus_size_ * poly_modulus_degree_ * 8,
                                                hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
        phantom::util::check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    }
};

// PhantomPlaintext (include/plaintext.h:8-116): metadata + one device buffer.  BFV / BGV plaintexts hold N coefficients
// modulo t (coeff_modulus_size 1); CKKS plaintexts hold [limb][coeff] in NTT form at their chain index.  The encoders
// that fill it are outside the accelerated path; load_from_host stands in for them.
class PhantomPlaintext {
    size_t chain_index_ = 0, poly_modulus_degree_ = 0, coeff_modulus_size_ = 0;
    double scale_ = 1.0;
    phantom::util::cuda_auto_ptr<uint64_t> data_;

public:
    void resize(size_t coeff_modulus_size, size_t poly_modulus_degree, const cudaStream_t &stream) {
        data_ = phantom::util::make_cuda_auto_ptr<uint64_t>(coeff_modulus_size * poly_modulus_degree, stream);
        coeff_modulus_size_ = coeff_modulus_size;
        poly_modulus_degree_ = poly_modulus_degree;
    }
    void set_chain_index(size_t chain_index) { chain_index_ = chain_index; }
    void set_scale(double scale) { scale_ = scale; }
    [[nodiscard]] size_t coeff_count() const noexcept { return poly_modulus_degree_ * coeff_modulus_size_; }
    [[nodiscard]] size_t chain_index() const noexcept { return chain_index_; }
    [[nodiscard]] size_t coeff_modulus_size() const noexcept { return coeff_modulus_size_; }
    [[nodiscard]] size_t poly_modulus_degree() const noexcept { return poly_modulus_degree_; }
    [[nodiscard]] double scale() const noexcept { return scale_; }
    [[nodiscard]] uint64_t *data() const noexcept { return data_.get(); }
    [[nodiscard]] phantom::util::cuda_auto_ptr<uint64_t> &data_ptr() noexcept { return data_; }

    // on-disk format of include/plaintext.h:72-114: four metadata fields, then the words
    void save(std::ostream &stream) const {
        stream.write(reinterpret_cast<const char *>(&chain_index_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&poly_modulus_degree_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&coeff_modulus_size_), sizeof(std::size_t));
        stream.write(reinterpret_cast<const char *>(&scale_), sizeof(double));
        std::vector<uint64_t> host(coeff_count());
        if (!host.empty()) store_to_host(host.data());
        stream.write(reinterpret_cast<const char *>(host.data()), static_cast<std::streamsize>(host.size() * sizeof(uint64_t)));
    }
    void load(std::istream &stream) {
        size_t degree = 0, limbs = 0;
        stream.read(reinterpret_cast<char *>(&chain_index_), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&degree), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&limbs), sizeof(std::size_t));
        stream.read(reinterpret_cast<char *>(&scale_), sizeof(double));
        if (!stream || limbs > 4096 || degree > (size_t(1) << 17)) throw std::invalid_argument("plaintext stream is not valid");
        std::vector<uint64_t> host(limbs * degree);
        stream.read(reinterpret_cast<char *>(host.data()), static_cast<std::streamsize>(host.size() * sizeof(uint64_t)));
        if (!stream) throw std::invalid_argument("plaintext stream is truncated");
        load_from_host(host.data(), limbs, degree, chain_index_, scale_);
    }
    void load_from_host(const uint64_t *host, size_t coeff_modulus_size, size_t poly_modulus_degree, size_t chain_index,
                        double scale = 1.0, const cudaStream_t &stream = cudaStreamPerThread) {
        resize(coeff_modulus_size, poly_modulus_degree, stream);
        chain_index_ = chain_index;
        scale_ = scale;
        if (coeff_count()) {
            phantom::util::check_hip(hipMemcpyAsync(data_.get(), host, coeff_count() * 8, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
            phantom::util::check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
        }
    }
    void store_to_host(uint64_t *host, const cudaStream_t &stream = cudaStreamPerThread) const {
        phantom::util::check_hip(hipMemcpyAsync(host, data_.get(), coeff_count() * 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
        phantom::util::check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    }
};

// PhantomRelinKey (include/secretkey.h:102-165): dnum public keys, each a size-2 ciphertext [2][#QP][N] at chain
// index 0, + the device pointer table the inner product reads
class PhantomRelinKey {
    std::vector<PhantomCiphertext> public_keys_;
    phantom::util::cuda_auto_ptr<uint64_t *> public_keys_ptr_;
    bool gen_flag_ = false;

    void allocate(const PhantomContext &context, size_t dnum, const cudaStream_t &stream) {
        public_keys_.clear();
        public_keys_.resize(dnum);
        for (auto &pk : public_keys_) {
            pk.resize(context, 0, 2, stream);
            pk.set_ntt_form(true);
        }
        upload_pointers(stream);
    }
    void upload_pointers(const cudaStream_t &stream) {
        std::vector<uint64_t *> ptrs;
        for (auto &pk : public_keys_) ptrs.push_back(pk.data());
        public_keys_ptr_ = phantom::util::make_cuda_auto_ptr<uint64_t *>(ptrs.size(), stream);
        phantom::util::check_hip(hipMemcpyAsync(public_keys_ptr_.get(), ptrs.data(), ptrs.size() * sizeof(uint64_t *),
                                                hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
        phantom::util::check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
    }

public:
    PhantomRelinKey() = default;
    // keys from outside: evk = [dnum][2][#QP][N] on the host
    void load_from_host(const PhantomContext &context, const uint64_t *evk, size_t dnum,
                        const cudaStream_t &stream = cudaStreamPerThread) {
        const size_t words = 2 * context.coeff_mod_size() * context.poly_degree();
        allocate(context, dnum, stream);
        for (size_t d = 0; d < dnum; d++)
            phantom::util::check_hip(hipMemcpyAsync(public_keys_[d].data(), evk + d * words, words * 8,
                                                    hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
        phantom::util::check_hip(hipStreamSynchronize(stream), "hipStreamSynchronize");
        gen_flag_ = true;
    }
    // PhantomSecretKey::generate_one_kswitch_key (src/secretkey.cu:297-341) with the randomness handed in (device
    // buffers, see pha_generate_one_kswitch_key): sk_ntt [QP][N], new_key_ntt [Q][N], a / e [dnum][QP][N]
    void generate(const PhantomContext &context, const uint64_t *sk_ntt, const uint64_t *new_key_ntt, const uint64_t *a,
                  uint64_t *e, const cudaStream_t &stream = cudaStreamPerThread) {
        const auto &parms = context.get_context_data(0).parms();
        const size_t size_p = parms.special_modulus_size();
        if (!context.using_keyswitching() || size_p == 0) throw std::invalid_argument("keyswitching is not supported by the context");
        allocate(context, (parms.coeff_modulus().size() - size_p) / size_p, stream);
        phantom::util::check_pha(pha_generate_one_kswitch_key(context.amd(), sk_ntt, new_key_ntt, a, e, public_keys_ptr_.get(),
                                                              static_cast<int>(parms.scheme()), stream));
        gen_flag_ = true;
    }
    // include/secretkey.h:129-163: dnum, then every public key as a ciphertext dump
    void save(std::ostream &stream) const {
        if (!gen_flag_) throw std::invalid_argument("PhantomRelinKey has not been generated");
        const size_t dnum = public_keys_.size();
        stream.write(reinterpret_cast<const char *>(&dnum), sizeof(std::size_t));
        for (const auto &pk : public_keys_) pk.save(stream);
    }
    void load(std::istream &stream) {
        size_t dnum = 0;
        stream.read(reinterpret_cast<char *>(&dnum), sizeof(std::size_t));
        if (!stream || dnum == 0 || dnum > 4096) throw std::invalid_argument("relin key stream is not valid");
        public_keys_.clear();
        public_keys_.resize(dnum);
        for (auto &pk : public_keys_) pk.load(stream);
        upload_pointers(cudaStreamPerThread);
        gen_flag_ = true;
    }
    // Extension (the reference is single-GPU): replicate the key generated on rank `root` on every rank of
    // `nccl_comm` (an ncclComm_t with one rank per GPU), in place, with ONE RCCL group over xGMI.  Ranks other than the root
    // call allocate_like() first, so that every rank holds buffers of the same shape.  Nothing else is ever communicated.
    void allocate_like(const PhantomContext &context, size_t dnum, const cudaStream_t &stream = cudaStreamPerThread) {
        allocate(context, dnum, stream);
        gen_flag_ = true;
    }
    void broadcast(const PhantomContext &context, int root, void *nccl_comm, const cudaStream_t &stream = cudaStreamPerThread) {
        if (!gen_flag_) throw std::invalid_argument("PhantomRelinKey has not been generated (non-root ranks: allocate_like)");
        std::vector<uint64_t *> ptrs;
        for (auto &pk : public_keys_) ptrs.push_back(pk.data());
        const size_t words = 2 * context.coeff_mod_size() * context.poly_degree();
        phantom::util::check_pha(pha_broadcast_keys(context.amd(), ptrs.data(), ptrs.size(), words, root, nccl_comm, stream));
    }
    [[nodiscard]] uint64_t **public_keys_ptr() const { return public_keys_ptr_.get(); }
    [[nodiscard]] const PhantomCiphertext &public_key(size_t d) const { return public_keys_.at(d); }
    [[nodiscard]] size_t dnum() const { return public_keys_.size(); }
    [[nodiscard]] bool generated() const { return gen_flag_; }
};

// PhantomGaloisKey (include/secretkey.h:170-220): one relin key per Galois element
class PhantomGaloisKey {
    std::vector<uint32_t> galois_elts_;
    std::vector<PhantomRelinKey> relin_keys_;

public:
    void add(uint32_t galois_elt, PhantomRelinKey &&key) {
        galois_elts_.push_back(galois_elt);
        relin_keys_.push_back(std::move(key));
    }
    [[nodiscard]] const std::vector<uint32_t> &galois_elts() const { return galois_elts_; }
    [[nodiscard]] const PhantomRelinKey &get_relin_keys(size_t index) const { return relin_keys_.at(index); }
    // Extension: every key of the set from rank `root` to all ranks of `nccl_comm` (see PhantomRelinKey::broadcast)
    void broadcast(const PhantomContext &context, int root, void *nccl_comm, const cudaStream_t &stream = cudaStreamPerThread) {
        for (auto &rlk : relin_keys_) rlk.broadcast(context, root, nccl_comm, stream);
    }
    // include/secretkey.h:196-220: the number of keys, then every relin key.  The file does not name the Galois
    // elements (the reference pairs key i with the context's i-th element), so load() takes them.
    void save(std::ostream &stream) const {
        if (relin_keys_.empty()) throw std::invalid_argument("PhantomGaloisKey has not been generated");
        const size_t rlk_num = relin_keys_.size();
        stream.write(reinterpret_cast<const char *>(&rlk_num), sizeof(std::size_t));
        for (const auto &rlk : relin_keys_) rlk.save(stream);
    }
    void load(std::istream &stream, const std::vector<uint32_t> &galois_elts) {
        size_t rlk_num = 0;
        stream.read(reinterpret_cast<char *>(&rlk_num), sizeof(std::size_t));
        if (!stream || rlk_num != galois_elts.size()) throw std::invalid_argument("galois key stream does not match the elements");
        galois_elts_ = galois_elts;
        relin_keys_.clear();
        relin_keys_.resize(rlk_num);
        for (auto &rlk : relin_keys_) rlk.load(stream);
    }
};

namespace phantom {

namespace detail {
inline size_t level_size_Ql(const PhantomContext &context, const PhantomCiphertext &ct) {
    // keyswitch_inplace src/eval_key_switch.cu:112-127: bfv -> level 1; ckks/bgv -> chain_index
    const auto scheme = context.get_context_data(0).parms().scheme();
    const size_t idx = scheme == scheme_type::bfv ? 1 : ct.chain_index();
    return context.get_context_data(idx).parms().coeff_modulus().size();
}
inline void same_shape(const PhantomCiphertext &a, const PhantomCiphertext &b) {
    if (a.chain_index() != b.chain_index()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (a.is_ntt_form() != b.is_ntt_form()) throw std::invalid_argument("NTT form mismatch");
    if (a.scale() != b.scale()) throw std::invalid_argument("scale mismatch");
    if (a.size() != b.size()) throw std::invalid_argument("poly number mismatch");
}
}  // namespace detail

// FindLevelsToDrop (src/evaluate.cu:551-647): how many primes of Q a BFV operation may ignore under hps_overq_leveled,
// from the noise estimate of a ciphertext of the given multiplicative depth (host arithmetic in double precision).
namespace detail {
inline size_t find_levels_to_drop(const PhantomContext &context, size_t multiplicativeDepth, double dcrtBits, bool isKeySwitch,
                                  bool isAsymmetric) {
    const auto &parms = context.get_context_data(0).parms();
    if (parms.mul_tech() != mul_tech_type::hps_overq_leveled)
        throw std::invalid_argument("FindLevelsToDrop is only used in HPS over Q Leveled");
    const auto n = static_cast<uint32_t>(parms.poly_modulus_degree());
    const double sigma = static_cast<double>(3.2f), alpha = 36.0;   // distributionParameter, assuranceMeasure (hestdparms.h:152-153)
    const double p = static_cast<double>(parms.plain_modulus().value());
    const size_t size_P = context.using_keyswitching() ? parms.special_modulus_size() : 0;
    const size_t size_Q = parms.coeff_modulus().size() - size_P;
    const auto k = static_cast<uint32_t>(size_P);
    const auto numPartQ = static_cast<uint32_t>(size_P ? (size_Q + size_P - 1) / size_P : 0);
    const double Bkey = 1.0;
    const double Berr = sigma * std::sqrt(alpha);
    const auto delta = [](uint32_t m) { return 2. * std::sqrt(static_cast<double>(m)); };
    const auto Vnorm = [&](uint32_t m) {
        if (isAsymmetric) return (1. + delta(m) * Bkey) / 2.;
        return Berr * (1. + 2. * delta(m) * Bkey);
    };
    const auto noiseKS = [&](uint32_t m) { return k * (numPartQ * delta(m) * Berr + delta(m) * Bkey + 1.0) / 2; };
    const auto C1 = [&](uint32_t m) { return delta(m) * delta(m) * p * Bkey; };
    const auto C2 = [&](uint32_t m) { return delta(m) * delta(m) * Bkey * Bkey / 2.0 + noiseKS(m); };
    const auto logqBFV = [&](uint32_t m) {
        if (multiplicativeDepth > 0)
            return std::log(4 * p) + (static_cast<double>(multiplicativeDepth) - 1) * std::log(C1(m)) +
                   std::log(C1(m) * Vnorm(m) + static_cast<double>(multiplicativeDepth) * C2(m));
        return std::log(p * (4 * Vnorm(m)));
    };
    double logqPrev = 6. * std::log(10.);
    double logq = logqBFV(n);
    while (std::fabs(logq - logqPrev) > std::log(1.001)) {
        logqPrev = logq;
        logq = logqBFV(n);
    }
    const double loge = logq / std::log(2.) - 2 - std::log2(p);
    const double logExtra = isKeySwitch ? std::log2(noiseKS(n)) : std::log2(delta(n));
    auto levels = static_cast<int32_t>(std::floor((loge - 2 * static_cast<double>(multiplicativeDepth) - 16 - logExtra) / dcrtBits));
    const auto sizeQ = static_cast<int32_t>(size_Q);
    if (levels < 0) levels = 0;
    else if (levels > sizeQ - 1) levels = sizeQ - 1;
    return static_cast<size_t>(levels);
}
// qMSB of the top level's DRNSTool (src/rns.cu:587): bit count of the largest prime of Q
inline double dcrt_bits(const PhantomContext &context) {
    const auto &parms = context.get_context_data(context.get_first_index()).parms();
    int bits = 0;
    for (const auto &m : parms.coeff_modulus()) bits = std::max(bits, m.bit_count());
    return static_cast<double>(bits);
}
}  // namespace detail

// phantom::keyswitch_inplace (include/evaluate.cuh:29-32, src/eval_key_switch.cu:95-182)
inline void keyswitch_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, uint64_t *c2,
                              const PhantomRelinKey &relin_keys, bool is_relin, const cudaStream_t &stream) {
    const auto &key_parms = context.get_context_data(0).parms();
    if (key_parms.scheme() == scheme_type::bfv && key_parms.mul_tech() == mul_tech_type::hps_overq_leveled) {
        // levels to ignore from the ciphertext's depth (:113-123); the key switch then runs at the lower level (:142-175)
        const size_t levelsDropped = detail::find_levels_to_drop(context, encrypted.GetNoiseScaleDeg() - 1, detail::dcrt_bits(context),
                                                                 !is_relin, encrypted.is_asymmetric());
        if (levelsDropped) {
            const size_t size_Q = context.get_context_data(1).parms().coeff_modulus().size();
            util::check_pha(pha_keyswitch_inplace_bfv_leveled(context.amd(), size_Q - levelsDropped, encrypted.data(), c2,
                                                              relin_keys.public_keys_ptr(), stream));
            return;
        }
    }
    util::check_pha(pha_keyswitch_inplace(context.amd(), detail::level_size_Ql(context, encrypted), encrypted.data(), c2,
                                          relin_keys.public_keys_ptr(), static_cast<int>(key_parms.scheme()), stream));
}

// negate / add / sub (src/evaluate.cu:80-343): residue-wise over size * coeff_modulus_size limbs
inline void negate_inplace(const PhantomContext &context, PhantomCiphertext &encrypted) {
    const auto &s = cudaStreamPerThread;
    const size_t L = encrypted.coeff_modulus_size(), n = encrypted.poly_modulus_degree();
    for (size_t i = 0; i < encrypted.size(); i++)
        util::check_pha(pha_negate_rns_poly(context.amd(), encrypted.data() + i * L * n, encrypted.data() + i * L * n, L, 0, s));
}
// balance_correction_factors (src/evaluate.cu:20-77): the pair (e1, e2) with e1 * f1 = e2 * f2 (mod t) of smallest
// |e1| + |e2| (centred) found along the extended Euclidean remainder sequence of (t, f2 / f1); returns (e1 * f1, e1, e2)
namespace detail {
inline std::tuple<uint64_t, uint64_t, uint64_t> balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t) {
    const auto mulmod = [t](uint64_t a, uint64_t b) { return static_cast<uint64_t>(static_cast<unsigned __int128>(a) * b % t); };
    const auto centred_abs = [t](uint64_t x) { return static_cast<int64_t>(x > t / 2 ? t - x : x); };
    const auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t r = a % b; a = b; b = r; } return a; };
    // factor1^-1 mod t
    __int128 r0 = t, r1 = factor1 % t, s0 = 0, s1 = 1;
    while (r1 != 0) {
        const __int128 k = r0 / r1, r2 = r0 - k * r1, s2 = s0 - k * s1;
        r0 = r1; r1 = r2; s0 = s1; s1 = s2;
    }
    if (r0 != 1) throw std::logic_error("invalid correction factor1");
    const uint64_t inv1 = static_cast<uint64_t>(((s0 % static_cast<__int128>(t)) + t) % t);
    const uint64_t ratio = mulmod(inv1, factor2 % t);
    uint64_t e1 = ratio, e2 = 1;
    int64_t best = centred_abs(e1) + centred_abs(e2);
    int64_t prev_a = static_cast<int64_t>(t), prev_b = 0, a = static_cast<int64_t>(ratio), b = 1;
    while (a != 0) {
        const int64_t q = prev_a / a, rem = prev_a % a;
        prev_a = a;
        a = rem;
        const int64_t nb = prev_b - b * q;
        prev_b = b;
        b = nb;
        const auto to_mod = [t](int64_t v) {
            const uint64_t m = static_cast<uint64_t>(v < 0 ? -v : v) % t;
            return (v < 0 && m) ? t - m : m;
        };
        const uint64_t a_mod = to_mod(a), b_mod = to_mod(b);
        if (a_mod != 0 && gcd(a_mod, t) == 1) {
            const int64_t sum = centred_abs(a_mod) + centred_abs(b_mod);
            if (sum < best) {
                best = sum;
                e1 = a_mod;
                e2 = b_mod;
            }
        }
    }
    return {mulmod(e1, factor1 % t), e1, e2};
}
// BGV: bring two ciphertexts to a common correction factor before adding / subtracting (src/evaluate.cu:148-168);
// returns the (possibly scaled) second operand
inline const PhantomCiphertext &balance(const PhantomContext &context, PhantomCiphertext &encrypted1,
                                        const PhantomCiphertext &encrypted2, PhantomCiphertext &scratch2) {
    if (encrypted1.correction_factor() == encrypted2.correction_factor()) return encrypted2;
    const auto &parms = context.get_context_data(encrypted1.chain_index()).parms();
    const auto f = balance_correction_factors(encrypted1.correction_factor(), encrypted2.correction_factor(), parms.plain_modulus().value());
    const auto &s = cudaStreamPerThread;
    const size_t L = encrypted1.coeff_modulus_size(), n = encrypted1.poly_modulus_degree();
    scratch2 = encrypted2;
    for (size_t i = 0; i < encrypted1.size(); i++)
        util::check_pha(pha_multiply_uniform_scalar_rns_poly(context.amd(), encrypted1.data() + i * L * n, std::get<1>(f),
                                                             encrypted1.data() + i * L * n, L, 0, s));
    for (size_t i = 0; i < scratch2.size(); i++)
        util::check_pha(pha_multiply_uniform_scalar_rns_poly(context.amd(), scratch2.data() + i * L * n, std::get<2>(f),
                                                             scratch2.data() + i * L * n, L, 0, s));
    encrypted1.set_correction_factor(std::get<0>(f));
    scratch2.set_correction_factor(std::get<0>(f));
    return scratch2;
}
}  // namespace detail

inline void add_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2) {
    detail::same_shape(encrypted1, encrypted2);
    const auto &s = cudaStreamPerThread;
    const size_t L = encrypted1.coeff_modulus_size(), n = encrypted1.poly_modulus_degree();
    PhantomCiphertext scaled;
    const PhantomCiphertext &rhs = detail::balance(context, encrypted1, encrypted2, scaled);
    for (size_t i = 0; i < encrypted1.size(); i++)
        util::check_pha(pha_add_rns_poly(context.amd(), encrypted1.data() + i * L * n, rhs.data() + i * L * n,
                                         encrypted1.data() + i * L * n, L, 0, s));
}
inline void sub_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2,
                        bool negate = false) {
    detail::same_shape(encrypted1, encrypted2);
    const auto &s = cudaStreamPerThread;
    const size_t L = encrypted1.coeff_modulus_size(), n = encrypted1.poly_modulus_degree();
    PhantomCiphertext scaled;
    const PhantomCiphertext &rhs = detail::balance(context, encrypted1, encrypted2, scaled);
    for (size_t i = 0; i < encrypted1.size(); i++) {
        uint64_t *a = encrypted1.data() + i * L * n;
        const uint64_t *b = rhs.data() + i * L * n;
        if (negate) util::check_pha(pha_sub_rns_poly(context.amd(), b, a, a, L, 0, s));
        else util::check_pha(pha_sub_rns_poly(context.amd(), a, b, a, L, 0, s));
    }
}

// add_many (src/evaluate.cu:202-262): BGV folds with add_inplace (correction factors), the others sum in one kernel
inline void add_many(const PhantomContext &context, const std::vector<PhantomCiphertext> &encrypteds, PhantomCiphertext &destination) {
    if (encrypteds.empty()) throw std::invalid_argument("encrypteds cannot be empty");
    for (const auto &e : encrypteds) {
        if (&e == &destination) throw std::invalid_argument("encrypteds must be different from destination");
        if (encrypteds[0].chain_index() != e.chain_index()) throw std::invalid_argument("encrypteds parameter mismatch");
        if (encrypteds[0].is_ntt_form() != e.is_ntt_form()) throw std::invalid_argument("NTT form mismatch");
        if (encrypteds[0].scale() != e.scale()) throw std::invalid_argument("scale mismatch");
        if (encrypteds[0].size() != e.size()) throw std::invalid_argument("poly number mismatch");
    }
    const auto &parms = context.get_context_data(encrypteds[0].chain_index()).parms();
    const auto &s = cudaStreamPerThread;
    if (parms.scheme() == scheme_type::bgv) {
        destination = encrypteds[0];
        for (size_t i = 1; i < encrypteds.size(); i++) add_inplace(context, destination, encrypteds[i]);
        return;
    }
    destination.resize(context, encrypteds[0].chain_index(), encrypteds[0].size(), s);
    destination.set_ntt_form(encrypteds[0].is_ntt_form());
    destination.set_scale(encrypteds[0].scale());
    std::vector<const uint64_t *> ptrs;
    for (const auto &e : encrypteds) ptrs.push_back(e.data());
    for (size_t i = 0; i < encrypteds[0].size(); i++)
        util::check_pha(pha_add_many_rns_poly(context.amd(), ptrs.data(), ptrs.size(), destination.data(), i,
                                              parms.coeff_modulus().size(), s));
}

// multiply_inplace (src/evaluate.cu:1030-1079 -> bgv_ckks_multiply :345-397)
inline void multiply_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1, const PhantomCiphertext &encrypted2) {
    const auto &parms = context.get_context_data(encrypted1.chain_index()).parms();
    if (parms.scheme() == scheme_type::bfv) {
        // bfv_multiply (src/evaluate.cu:962-982) -> bfv_multiply_behz (:447-548) or bfv_multiply_hps with mul_tech hps /
        // hps_overq / hps_overq_leveled (:674-818)
        const auto mul_tech = parms.mul_tech();
        if (mul_tech == mul_tech_type::none) throw std::invalid_argument("mul_tech not supported for bfv_multiply");
        if (encrypted1.is_ntt_form() || encrypted2.is_ntt_form())
            throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
        if (encrypted1.chain_index() != encrypted2.chain_index())
            throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (encrypted1.size() != 2 || encrypted2.size() != 2)
            throw std::invalid_argument("only 2x2 tensor products are on the accelerated path");
        if (encrypted1.chain_index() != context.get_first_index())
            throw std::invalid_argument("BFV multiply runs at the top data level");
        const auto &s = cudaStreamPerThread;
        const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree();
        auto out = util::make_cuda_auto_ptr<uint64_t>(3 * L * n, s);
        if (mul_tech == mul_tech_type::behz)
            util::check_pha(pha_bfv_multiply_behz(context.amd(), encrypted1.data(), encrypted2.data(), out.get(), s));
        else if (mul_tech == mul_tech_type::hps_overq)
            util::check_pha(pha_bfv_multiply_hps_overq(context.amd(), encrypted1.data(), encrypted2.data(), out.get(), s));
        else if (mul_tech == mul_tech_type::hps_overq_leveled) {   // levels from the operands' depth (:680-691)
            const size_t levels = std::max(encrypted1.GetNoiseScaleDeg(), encrypted2.GetNoiseScaleDeg()) - 1;
            const size_t dropped = detail::find_levels_to_drop(context, levels, detail::dcrt_bits(context), false, encrypted1.is_asymmetric());
            util::check_pha(pha_bfv_multiply_hps_overq_leveled(context.amd(), L - dropped, encrypted1.data(), encrypted2.data(), out.get(), s));
            encrypted1.SetNoiseScaleDeg(std::max(encrypted1.GetNoiseScaleDeg(), encrypted2.GetNoiseScaleDeg()) + 1);   // :798-800
        } else
            util::check_pha(pha_bfv_multiply_hps(context.amd(), encrypted1.data(), encrypted2.data(), out.get(), s));
        encrypted1.resize(context, encrypted1.chain_index(), 3, s);
        util::check_hip(hipMemcpyAsync(encrypted1.data(), out.get(), 3 * L * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        return;
    }
    if (!(encrypted1.is_ntt_form() && encrypted2.is_ntt_form()))
        throw std::invalid_argument("encrypted1 and encrypted2 must be in NTT form");
    if (encrypted1.chain_index() != encrypted2.chain_index())
        throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
    if (encrypted1.size() != 2 || encrypted2.size() != 2)
        throw std::invalid_argument("only 2x2 tensor products are on the accelerated path");
    const auto &s = cudaStreamPerThread;
    const size_t L = parms.coeff_modulus().size();
    const bool square = &encrypted1 == &encrypted2;
    encrypted1.resize(context, encrypted1.chain_index(), 3, s);
    if (square) util::check_pha(pha_tensor_square_2x2_rns_poly(context.amd(), encrypted1.data(), encrypted1.data(), L, s));
    else util::check_pha(pha_tensor_prod_2x2_rns_poly(context.amd(), encrypted1.data(), encrypted2.data(), encrypted1.data(), L, s));
    if (parms.scheme() == scheme_type::ckks) encrypted1.set_scale(encrypted1.scale() * encrypted2.scale());
    if (parms.scheme() == scheme_type::bgv)  // evaluate.cu:392-396
        encrypted1.set_correction_factor(static_cast<uint64_t>(
            (static_cast<unsigned __int128>(encrypted1.correction_factor()) * encrypted2.correction_factor()) %
            parms.plain_modulus().value()));
}

// relinearize_inplace (src/evaluate.cu:1342-1374)
inline void relinearize_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomRelinKey &relin_keys) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    const size_t decomp_modulus_size = parms.coeff_modulus().size();
    const size_t n = parms.poly_modulus_degree();
    const auto scheme = parms.scheme();
    if (encrypted.size() != 3) throw std::invalid_argument("destination_size must be 3");
    if (scheme == scheme_type::bfv && encrypted.is_ntt_form()) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
    if (scheme == scheme_type::ckks && !encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    if (scheme == scheme_type::bgv && !encrypted.is_ntt_form()) throw std::invalid_argument("BGV encrypted must be in NTT form");
    uint64_t *c2 = encrypted.data() + 2 * decomp_modulus_size * n;
    const auto &s = cudaStreamPerThread;
    keyswitch_inplace(context, encrypted, c2, relin_keys, true, s);
    encrypted.resize(2, decomp_modulus_size, n, s);
}
// multiply_and_relin_inplace (src/evaluate.cu:1064-1103): for the HPS variants of BFV the reference fuses the two
// (bfv_mul_relin_hps :822-1027).  Without dropped levels that is multiply followed by the ordinary key switch; with dropped
// levels c2 never leaves level l and the key switch reuses the multiply's level count (not a second estimate).
inline void multiply_and_relin_inplace(const PhantomContext &context, PhantomCiphertext &encrypted1,
                                       const PhantomCiphertext &encrypted2, const PhantomRelinKey &relin_keys) {
    const auto &parms = context.get_context_data(encrypted1.chain_index()).parms();
    if (parms.scheme() == scheme_type::bfv && parms.mul_tech() == mul_tech_type::hps_overq_leveled) {
        if (encrypted1.is_ntt_form() || encrypted2.is_ntt_form()) throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
        if (encrypted1.size() != 2 || encrypted2.size() != 2) throw std::logic_error("dest_size must be 3 when computing BFV multiplication using HPS");
        if (encrypted1.chain_index() != context.get_first_index() || encrypted2.chain_index() != encrypted1.chain_index())
            throw std::invalid_argument("BFV multiply runs at the top data level");
        const auto &s = cudaStreamPerThread;
        const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree();
        const size_t depth = std::max(encrypted1.GetNoiseScaleDeg(), encrypted2.GetNoiseScaleDeg());
        const size_t dropped = detail::find_levels_to_drop(context, depth - 1, detail::dcrt_bits(context), false, encrypted1.is_asymmetric());
        if (dropped) {
            auto out = util::make_cuda_auto_ptr<uint64_t>(2 * L * n, s);
            util::check_pha(pha_bfv_mul_relin_hps_overq_leveled(context.amd(), L - dropped, encrypted1.data(), encrypted2.data(),
                                                                relin_keys.public_keys_ptr(), out.get(), s));
            util::check_hip(hipMemcpyAsync(encrypted1.data(), out.get(), 2 * L * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        } else {
            auto out = util::make_cuda_auto_ptr<uint64_t>(3 * L * n, s);
            util::check_pha(pha_bfv_multiply_hps_overq_leveled(context.amd(), L, encrypted1.data(), encrypted2.data(), out.get(), s));
            util::check_pha(pha_keyswitch_inplace(context.amd(), L, out.get(), out.get() + 2 * L * n, relin_keys.public_keys_ptr(),
                                                  static_cast<int>(scheme_type::bfv), s));
            util::check_hip(hipMemcpyAsync(encrypted1.data(), out.get(), 2 * L * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        }
        encrypted1.SetNoiseScaleDeg(depth + 1);
        return;
    }
    multiply_inplace(context, encrypted1, encrypted2);
    relinearize_inplace(context, encrypted1, relin_keys);
}

// rescale_to_next (src/evaluate.cu:1545-1565 -> mod_switch_scale_to_next :1376-1427)
[[nodiscard]] inline PhantomCiphertext rescale_to_next(const PhantomContext &context, const PhantomCiphertext &encrypted) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    if (parms.scheme() != scheme_type::ckks) throw std::invalid_argument("unsupported operation for scheme type");
    if (!encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    const auto &s = cudaStreamPerThread;
    const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree(), size = encrypted.size();
    const size_t next = context.get_next_index(encrypted.chain_index());
    auto copy = util::make_cuda_auto_ptr<uint64_t>(size * L * n, s);  // the reference rescales a copy (:1392-1395)
    util::check_hip(hipMemcpyAsync(copy.get(), encrypted.data(), size * L * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    PhantomCiphertext destination;
    destination.resize(context, next, size, s);
    util::check_pha(pha_divide_and_round_q_last_ntt(context.amd(), L, copy.get(), size, destination.data(), s));
    destination.set_ntt_form(encrypted.is_ntt_form());
    destination.set_scale(encrypted.scale() / static_cast<double>(parms.coeff_modulus().back().value()));
    return destination;
}
inline void rescale_to_next_inplace(const PhantomContext &context, PhantomCiphertext &encrypted) {
    encrypted = rescale_to_next(context, encrypted);
}

// Extension (no reference function): relinearize_inplace (src/evaluate.cu:1028-1077) followed by rescale_to_next
// (:1376-1427) as ONE call -- the same ciphertext bit for bit (pha_keyswitch_rescale: NTT is linear, so the mod-down's and the
// rescale's forward transforms are one), 17 % less GPU time at N = 2^16 / 45 + 15 limbs.  ckks, size-3 input.
[[nodiscard]] inline PhantomCiphertext relinearize_rescale(const PhantomContext &context, const PhantomCiphertext &encrypted,
                                                           const PhantomRelinKey &relin_keys) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    if (parms.scheme() != scheme_type::ckks) throw std::invalid_argument("unsupported operation for scheme type");
    if (encrypted.size() != 3) throw std::invalid_argument("destination_size must be 3");
    if (!encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    if (!relin_keys.generated()) throw std::invalid_argument("PhantomRelinKey has not been generated");
    const auto &s = cudaStreamPerThread;
    const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree();
    const size_t next = context.get_next_index(encrypted.chain_index());
    PhantomCiphertext destination;
    destination.resize(context, next, 2, s);
    util::check_pha(pha_keyswitch_rescale(context.amd(), detail::level_size_Ql(context, encrypted), encrypted.data(),
                                          encrypted.data() + 2 * L * n, relin_keys.public_keys_ptr(), destination.data(), s));
    destination.set_ntt_form(true);
    destination.set_scale(encrypted.scale() / static_cast<double>(parms.coeff_modulus().back().value()));
    return destination;
}

// mod_switch_to_next (src/evaluate.cu:1506-1543): CKKS drops the last limb (mod_switch_drop_to_next
// :1429-1470); BFV / BGV divide by q_last (mod_switch_scale_to_next :1376-1427)
[[nodiscard]] inline PhantomCiphertext mod_switch_to_next(const PhantomContext &context, const PhantomCiphertext &encrypted) {
    const auto &first = context.get_context_data(context.get_first_index()).parms();
    const auto scheme = first.scheme();
    if (encrypted.chain_index() == first.coeff_modulus().size())
        throw std::invalid_argument("end of modulus switching chain reached");
    if (scheme == scheme_type::bfv && encrypted.is_ntt_form()) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
    if (scheme == scheme_type::bgv && !encrypted.is_ntt_form()) throw std::invalid_argument("BGV encrypted must be in NTT form");
    if (scheme == scheme_type::ckks && !encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    const auto &s = cudaStreamPerThread;
    const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree(), size = encrypted.size();
    const size_t next = context.get_next_index(encrypted.chain_index());
    PhantomCiphertext destination;
    destination.resize(context, next, size, s);
    destination.set_ntt_form(encrypted.is_ntt_form());
    destination.set_scale(encrypted.scale());
    if (scheme == scheme_type::ckks) {
        for (size_t i = 0; i < size; i++)
            util::check_hip(hipMemcpyAsync(destination.data() + i * (L - 1) * n, encrypted.data() + i * L * n, (L - 1) * n * 8,
                                           hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        return destination;
    }
    if (scheme == scheme_type::bfv) {
        util::check_pha(pha_divide_and_round_q_last(context.amd(), L, encrypted.data(), size, destination.data(), s));
        return destination;
    }
    if (scheme != scheme_type::bgv) throw std::invalid_argument("unsupported scheme");
    auto copy = util::make_cuda_auto_ptr<uint64_t>(size * L * n, s);  // the switch works on a copy (:1392-1395)
    util::check_hip(hipMemcpyAsync(copy.get(), encrypted.data(), size * L * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    util::check_pha(pha_mod_t_and_divide_q_last_ntt(context.amd(), L, copy.get(), size, destination.data(), s));
    // correction factor *= q_last^-1 mod t (:1421-1425)
    const uint64_t t = parms.plain_modulus().value();
    const uint64_t q_last_t = parms.coeff_modulus().back().value() % t;
    __int128 r0 = t, r1 = q_last_t, s0 = 0, s1 = 1;
    while (r1 != 0) {
        const __int128 k = r0 / r1, r2 = r0 - k * r1, s2 = s0 - k * s1;
        r0 = r1; r1 = r2; s0 = s1; s1 = s2;
    }
    const uint64_t inv = static_cast<uint64_t>(((s0 % static_cast<__int128>(t)) + t) % t);
    destination.set_correction_factor(
        static_cast<uint64_t>((static_cast<unsigned __int128>(encrypted.correction_factor()) * inv) % t));
    return destination;
}
inline void mod_switch_to_next_inplace(const PhantomContext &context, PhantomCiphertext &encrypted) {
    encrypted = mod_switch_to_next(context, encrypted);
}

// mod_switch_to (include/evaluate.cuh:153-177)
inline void mod_switch_to_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, size_t chain_index) {
    if (encrypted.chain_index() > chain_index) throw std::invalid_argument("cannot switch to higher level modulus");
    while (encrypted.chain_index() != chain_index) mod_switch_to_next_inplace(context, encrypted);
}
[[nodiscard]] inline PhantomCiphertext mod_switch_to(const PhantomContext &context, const PhantomCiphertext &encrypted, size_t chain_index) {
    PhantomCiphertext destination = encrypted;
    mod_switch_to_inplace(context, destination, chain_index);
    return destination;
}
// plaintext mod switch (src/evaluate.cu:1474-1503): keeps the leading limbs of the next level
inline void mod_switch_to_next_inplace(const PhantomContext &context, PhantomPlaintext &plain) {
    const auto &first = context.get_context_data(context.get_first_index()).parms();
    if (plain.chain_index() == first.coeff_modulus().size()) throw std::invalid_argument("end of modulus switching chain reached");
    const size_t next = plain.chain_index() + 1;
    const auto &next_parms = context.get_context_data(next).parms();
    const size_t limbs = next_parms.coeff_modulus().size(), n = next_parms.poly_modulus_degree();
    const auto &s = cudaStreamPerThread;
    auto old = std::move(plain.data_ptr());
    const double scale = plain.scale();
    plain.resize(limbs, n, s);
    plain.set_scale(scale);
    util::check_hip(hipMemcpyAsync(plain.data(), old.get(), limbs * n * 8, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    plain.set_chain_index(next);
}
[[nodiscard]] inline PhantomPlaintext mod_switch_to_next(const PhantomContext &context, const PhantomPlaintext &plain) {
    PhantomPlaintext destination = plain;
    mod_switch_to_next_inplace(context, destination);
    return destination;
}
inline void mod_switch_to_inplace(const PhantomContext &context, PhantomPlaintext &plain, size_t chain_index) {
    if (plain.chain_index() > chain_index) throw std::invalid_argument("cannot switch to higher level modulus");
    while (plain.chain_index() != chain_index) mod_switch_to_next_inplace(context, plain);
}
[[nodiscard]] inline PhantomPlaintext mod_switch_to(const PhantomContext &context, const PhantomPlaintext &plain, size_t chain_index) {
    PhantomPlaintext destination = plain;
    mod_switch_to_inplace(context, destination, chain_index);
    return destination;
}

// add_plain_inplace / sub_plain_inplace (src/evaluate.cu:1105-1226)
namespace detail {
inline void add_sub_plain(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain, bool subtract) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    const auto scheme = parms.scheme();
    if (scheme == scheme_type::bfv && encrypted.is_ntt_form()) throw std::invalid_argument("BFV encrypted cannot be in NTT form");
    if (scheme == scheme_type::ckks && !encrypted.is_ntt_form()) throw std::invalid_argument("CKKS encrypted must be in NTT form");
    if (scheme == scheme_type::bgv && !encrypted.is_ntt_form()) throw std::invalid_argument("BGV encrypted must be in NTT form");
    if (encrypted.scale() != plain.scale()) throw std::invalid_argument("scale mismatch");
    const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree();
    const auto &s = cudaStreamPerThread;
    switch (scheme) {
        case scheme_type::bfv:   // multiply_{add,sub}_plain_with_scaling_variant src/scalingvariant.cu:10-60
            util::check_pha(pha_bfv_add_plain(context.amd(), L, encrypted.data(), plain.data(), subtract ? 1 : 0, s));
            break;
        case scheme_type::ckks:  // (c0 +- pt, c1)
            if (plain.chain_index() != encrypted.chain_index() || plain.coeff_modulus_size() != L)
                throw std::invalid_argument("encrypted and plain parameter mismatch");
            if (subtract) util::check_pha(pha_sub_rns_poly(context.amd(), encrypted.data(), plain.data(), encrypted.data(), L, 0, s));
            else util::check_pha(pha_add_rns_poly(context.amd(), encrypted.data(), plain.data(), encrypted.data(), L, 0, s));
            break;
        case scheme_type::bgv: {  // lift t -> {q_i} in NTT form, then c0 +- correction_factor * pt
            auto lifted = util::make_cuda_auto_ptr<uint64_t>(L * n, s);
            util::check_pha(pha_bgv_lift_plain(context.amd(), L, plain.data(), lifted.get(), s));
            if (subtract)
                util::check_pha(pha_multiply_scalar_and_sub_rns_poly(context.amd(), encrypted.data(), lifted.get(),
                                                                     encrypted.correction_factor(), encrypted.data(), L, 0, s));
            else
                util::check_pha(pha_multiply_scalar_and_add_rns_poly(context.amd(), encrypted.data(), lifted.get(),
                                                                     encrypted.correction_factor(), encrypted.data(), L, 0, s));
            break;
        }
        default:
            throw std::invalid_argument("unsupported scheme");
    }
}
}  // namespace detail
inline void add_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain) {
    detail::add_sub_plain(context, encrypted, plain, false);
}
inline void sub_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain) {
    detail::add_sub_plain(context, encrypted, plain, true);
}

// multiply_plain_inplace (src/evaluate.cu:1228-1340)
inline void multiply_plain_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, const PhantomPlaintext &plain) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    const size_t L = parms.coeff_modulus().size(), n = parms.poly_modulus_degree();
    const auto &s = cudaStreamPerThread;
    const double new_scale = encrypted.scale() * plain.scale();
    switch (parms.scheme()) {
        case scheme_type::bfv:   // multiply_plain_normal :1256-1300
            util::check_pha(pha_bfv_multiply_plain(context.amd(), L, encrypted.data(), encrypted.size(), plain.data(), s));
            break;
        case scheme_type::ckks:  // multiply_plain_ntt :1228-1254
            if (encrypted.chain_index() != plain.chain_index()) throw std::invalid_argument("encrypted and plain parameter mismatch");
            if (encrypted.poly_modulus_degree() != plain.poly_modulus_degree() || plain.coeff_modulus_size() != L)
                throw std::invalid_argument("encrypted and plain parameter mismatch");
            for (size_t i = 0; i < encrypted.size(); i++)
                util::check_pha(pha_multiply_rns_poly(context.amd(), encrypted.data() + i * L * n, plain.data(),
                                                      encrypted.data() + i * L * n, L, 0, s));
            break;
        case scheme_type::bgv: {
            auto lifted = util::make_cuda_auto_ptr<uint64_t>(L * n, s);
            util::check_pha(pha_bgv_lift_plain(context.amd(), L, plain.data(), lifted.get(), s));
            for (size_t i = 0; i < encrypted.size(); i++)
                util::check_pha(pha_multiply_rns_poly(context.amd(), encrypted.data() + i * L * n, lifted.get(),
                                                      encrypted.data() + i * L * n, L, 0, s));
            break;
        }
        default:
            throw std::invalid_argument("unsupported scheme");
    }
    encrypted.set_scale(new_scale);
}

// apply_galois_inplace (src/evaluate.cu:1567-1630)
inline void apply_galois_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, size_t galois_elt,
                                 const PhantomGaloisKey &galois_keys) {
    const auto &parms = context.get_context_data(encrypted.chain_index()).parms();
    const size_t N = parms.poly_modulus_degree(), L = parms.coeff_modulus().size();
    if (encrypted.size() > 2) throw std::invalid_argument("encrypted size must be 2");
    const auto &elts = galois_keys.galois_elts();
    const auto it = std::find(elts.begin(), elts.end(), static_cast<uint32_t>(galois_elt));
    if (it == elts.end()) throw std::invalid_argument("Galois elt not present");
    const size_t idx = static_cast<size_t>(it - elts.begin());
    const auto &s = cudaStreamPerThread;
    if (parms.scheme() != scheme_type::bfv && parms.scheme() != scheme_type::ckks && parms.scheme() != scheme_type::bgv)
        throw std::logic_error("scheme not implemented");
    // the reference permutes c0 through a temporary, copies it back, permutes c1 into the temporary and clears c1
    // (:1597-1622); here one kernel writes (galois(c0), 0) into a fresh ciphertext buffer and galois(c1) into the key-switch
    // operand, and the ciphertext adopts the fresh buffer
    auto temp = util::make_cuda_auto_ptr<uint64_t>(L * N, s);
    auto fresh = util::make_cuda_auto_ptr<uint64_t>(2 * L * N, s);
    util::check_pha(pha_apply_galois_for_keyswitch(context.amd(), encrypted.data(), fresh.get(), temp.get(), static_cast<uint32_t>(galois_elt),
                                                   L, 1, parms.scheme() == scheme_type::bfv ? 0 : 1, s));
    encrypted.replace_data(std::move(fresh), s);
    keyswitch_inplace(context, encrypted, temp.get(), galois_keys.get_relin_keys(idx), false, s);
}

// rotate_inplace (src/evaluate.cu:1632-1668): direct key if present, else NAF decomposition of the step
inline void rotate_inplace(const PhantomContext &context, PhantomCiphertext &encrypted, int step,
                           const PhantomGaloisKey &galois_key) {
    const size_t coeff_count = context.get_context_data(encrypted.chain_index()).parms().poly_modulus_degree();
    const auto &elts = galois_key.galois_elts();
    const uint32_t elt = util::get_elt_from_step(step, coeff_count);
    if (std::find(elts.begin(), elts.end(), elt) != elts.end()) {
        apply_galois_inplace(context, encrypted, elt, galois_key);
        return;
    }
    const std::vector<int> naf_step = util::naf(step);
    if (naf_step.size() == 1) throw std::invalid_argument("Galois key not present");
    for (int t : naf_step)
        if (static_cast<size_t>(std::abs(t)) != (coeff_count >> 1)) rotate_inplace(context, encrypted, t, galois_key);
}

// hoisting_inplace (src/evaluate.cu:1670-1866): ct <- sum over `steps` of rotate(ct, step) with ONE shared mod-up
inline void hoisting_inplace(const PhantomContext &context, PhantomCiphertext &ct, const PhantomGaloisKey &glk,
                             const std::vector<int> &steps) {
    if (ct.size() > 2) throw std::invalid_argument("ciphertext size must be 2");
    const auto &key_parms = context.get_context_data(0).parms();
    const std::vector<uint32_t> elts = util::get_elts_from_steps(steps, key_parms.poly_modulus_degree());
    const auto &have = glk.galois_elts();
    std::vector<const uint64_t *const *> tables;
    for (uint32_t e : elts) {
        const auto it = std::find(have.begin(), have.end(), e);
        if (it == have.end()) throw std::logic_error("Galois key not present in hoisting");
        tables.push_back(glk.get_relin_keys(static_cast<size_t>(it - have.begin())).public_keys_ptr());
    }
    if (key_parms.scheme() == scheme_type::bfv && key_parms.mul_tech() == mul_tech_type::hps_overq_leveled) {
        // levels from the depth (:1689-1700); with dropped levels both polynomials are scaled down to Ql, rotated and summed
        // at that level, and expanded back to Q (:1732-1738, :1757-1763, :1845-1862)
        const size_t dropped = detail::find_levels_to_drop(context, ct.GetNoiseScaleDeg() - 1, detail::dcrt_bits(context), true, ct.is_asymmetric());
        if (dropped) {
            const auto &s = cudaStreamPerThread;
            const size_t size_Q = context.get_context_data(1).parms().coeff_modulus().size(), ql = size_Q - dropped;
            const size_t n = key_parms.poly_modulus_degree();
            auto low = util::make_cuda_auto_ptr<uint64_t>(2 * ql * n, s);
            for (size_t p = 0; p < 2; p++)
                util::check_pha(pha_scaleAndRound_HPS_Q_Ql(context.amd(), ql, low.get() + p * ql * n, ct.data() + p * size_Q * n, s));
            util::check_pha(pha_hoisting(context.amd(), ql, low.get(), elts.data(), elts.size(), tables.data(),
                                         static_cast<int>(scheme_type::bfv), s));
            for (size_t p = 0; p < 2; p++)
                util::check_pha(pha_ExpandCRTBasis_Ql_Q(context.amd(), ql, ct.data() + p * size_Q * n, low.get() + p * ql * n, s));
            return;
        }
    }
    util::check_pha(pha_hoisting(context.amd(), detail::level_size_Ql(context, ct), ct.data(), elts.data(), elts.size(),
                                 tables.data(), static_cast<int>(key_parms.scheme()), cudaStreamPerThread));
}
inline PhantomCiphertext hoisting(const PhantomContext &c, const PhantomCiphertext &e, const PhantomGaloisKey &k, const std::vector<int> &steps) { PhantomCiphertext d = e; hoisting_inplace(c, d, k, steps); return d; }

// out-of-place forms (include/evaluate.cuh): copy, then the in-place op
inline PhantomCiphertext negate(const PhantomContext &c, const PhantomCiphertext &e) { PhantomCiphertext d = e; negate_inplace(c, d); return d; }
inline PhantomCiphertext add(const PhantomContext &c, const PhantomCiphertext &a, const PhantomCiphertext &b) { PhantomCiphertext d = a; add_inplace(c, d, b); return d; }
inline PhantomCiphertext sub(const PhantomContext &c, const PhantomCiphertext &a, const PhantomCiphertext &b, bool negate = false) { PhantomCiphertext d = a; sub_inplace(c, d, b, negate); return d; }
inline PhantomCiphertext multiply(const PhantomContext &c, const PhantomCiphertext &a, const PhantomCiphertext &b) { PhantomCiphertext d = a; multiply_inplace(c, d, b); return d; }
// multiply + relinearize + rescale (examples/3_ckks.cu:496-498) with the fused second half
[[nodiscard]] inline PhantomCiphertext multiply_relin_rescale(const PhantomContext &context, const PhantomCiphertext &encrypted1,
                                                              const PhantomCiphertext &encrypted2, const PhantomRelinKey &relin_keys) {
    return relinearize_rescale(context, multiply(context, encrypted1, encrypted2), relin_keys);
}
inline PhantomCiphertext relinearize(const PhantomContext &c, const PhantomCiphertext &e, const PhantomRelinKey &k) { PhantomCiphertext d = e; relinearize_inplace(c, d, k); return d; }
inline PhantomCiphertext multiply_and_relin(const PhantomContext &c, const PhantomCiphertext &a, const PhantomCiphertext &b, const PhantomRelinKey &k) { PhantomCiphertext d = a; multiply_and_relin_inplace(c, d, b, k); return d; }
inline PhantomCiphertext apply_galois(const PhantomContext &c, const PhantomCiphertext &e, size_t elt, const PhantomGaloisKey &k) { PhantomCiphertext d = e; apply_galois_inplace(c, d, elt, k); return d; }
inline PhantomCiphertext rotate(const PhantomContext &c, const PhantomCiphertext &e, int step, const PhantomGaloisKey &k) { PhantomCiphertext d = e; rotate_inplace(c, d, step, k); return d; }
inline PhantomCiphertext add_plain(const PhantomContext &c, const PhantomCiphertext &e, const PhantomPlaintext &p) { PhantomCiphertext d = e; add_plain_inplace(c, d, p); return d; }
inline PhantomCiphertext sub_plain(const PhantomContext &c, const PhantomCiphertext &e, const PhantomPlaintext &p) { PhantomCiphertext d = e; sub_plain_inplace(c, d, p); return d; }
inline PhantomCiphertext multiply_plain(const PhantomContext &c, const PhantomCiphertext &e, const PhantomPlaintext &p) { PhantomCiphertext d = e; multiply_plain_inplace(c, d, p); return d; }

}  // namespace phantom

// ------------------------------------------------------------------------------------------------------------------------
// The launcher level of the reference (include/ntt.cuh:157-226, include/evaluate.cuh:25-27, include/polymath.cuh): the same
// names, argument order and meaning, forwarding to the C ABI.  `const DNTTTable &` is the handle PhantomContext::gpu_rns_tables().
// ------------------------------------------------------------------------------------------------------------------------
#define PHA_NTT_FWD_(call) phantom::util::check_pha(call)
inline void nwt_2d_radix8_forward_inplace(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                          size_t start_modulus_idx, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_inplace(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx, stream));
}
inline void nwt_2d_radix8_forward_inplace_fuse_moddown(uint64_t *ct, const uint64_t *cx, const uint64_t *bigPInv_mod_q,
                                                       const uint64_t *bigPInv_mod_q_shoup, uint64_t *delta, const DNTTTable &ntt_tables,
                                                       size_t coeff_modulus_size, size_t start_modulus_idx, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_inplace_fuse_moddown(ntt_tables.amd(), ct, cx, bigPInv_mod_q, bigPInv_mod_q_shoup, delta,
                                                                coeff_modulus_size, start_modulus_idx, stream));
}
inline void nwt_2d_radix8_forward_inplace_include_temp_mod(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                                           size_t start_modulus_idx, size_t total_modulus_size, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_inplace_include_temp_mod(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx,
                                                                    total_modulus_size, stream));
}
inline void nwt_2d_radix8_forward_inplace_include_special_mod(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                                              size_t start_modulus_idx, size_t size_QP, size_t size_P,
                                                              const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_inplace_include_special_mod(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx,
                                                                       size_QP, size_P, stream));
}
inline void nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(uint64_t *inout, const DNTTTable &ntt_tables,
                                                                            size_t coeff_modulus_size, size_t start_modulus_idx,
                                                                            size_t size_QP, size_t size_P, size_t excluded_range_start,
                                                                            size_t excluded_range_end, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_inplace_include_special_mod_exclude_range(
        ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx, size_QP, size_P, excluded_range_start, excluded_range_end, stream));
}
inline void nwt_2d_radix8_forward_modup_fuse(uint64_t *out, const uint64_t *in, size_t modulus_index, const DNTTTable &ntt_tables,
                                             size_t coeff_modulus_size, size_t start_modulus_idx, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_forward_modup_fuse(ntt_tables.amd(), out, in, modulus_index, coeff_modulus_size, start_modulus_idx, stream));
}
inline void nwt_2d_radix8_backward_inplace(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                           size_t start_modulus_idx, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward_inplace(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx, stream));
}
inline void nwt_2d_radix8_backward(uint64_t *out, const uint64_t *in, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                   size_t start_modulus_idx, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward(ntt_tables.amd(), out, in, coeff_modulus_size, start_modulus_idx, stream));
}
inline void nwt_2d_radix8_backward_scale(uint64_t *out, const uint64_t *in, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                         size_t start_modulus_idx, const uint64_t *scale, const uint64_t *scale_shoup,
                                         const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward_scale(ntt_tables.amd(), out, in, coeff_modulus_size, start_modulus_idx, scale, scale_shoup, stream));
}
inline void nwt_2d_radix8_backward_inplace_scale(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                                 size_t start_modulus_idx, const uint64_t *scale, const uint64_t *scale_shoup,
                                                 const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward_inplace_scale(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx, scale, scale_shoup, stream));
}
inline void nwt_2d_radix8_backward_inplace_include_special_mod(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                                               size_t start_modulus_idx, size_t size_QP, size_t size_P,
                                                               const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward_inplace_include_special_mod(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx,
                                                                        size_QP, size_P, stream));
}
inline void nwt_2d_radix8_backward_inplace_include_temp_mod_scale(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                                                  size_t start_modulus_idx, size_t total_modulus_size, const uint64_t *scale,
                                                                  const uint64_t *scale_shoup, const cudaStream_t &stream) {
    PHA_NTT_FWD_(pha_nwt_2d_radix8_backward_inplace_include_temp_mod_scale(ntt_tables.amd(), inout, coeff_modulus_size, start_modulus_idx,
                                                                           total_modulus_size, scale, scale_shoup, stream));
}
#undef PHA_NTT_FWD_

namespace phantom {

// phantom::key_switch_inner_prod (include/evaluate.cuh:25-27, src/eval_key_switch.cu:71-92).  modulus_QP is the device
// table the reference's kernel indexes (the library owns its own copy) and reduction_threshold its dead branch
// (SURVEY.md R6): both are accepted and unused.
inline void key_switch_inner_prod(uint64_t *p_cx, const uint64_t *p_t_mod_up, const uint64_t *const *rlk, const DRNSTool &rns_tool,
                                  const DModulus * /*modulus_QP*/, size_t /*reduction_threshold*/, const cudaStream_t &stream) {
    util::check_pha(pha_key_switch_inner_prod(rns_tool.amd(), rns_tool.size_Ql(), p_cx, p_t_mod_up, rlk, stream));
}

// The residue-wise kernels of include/polymath.cuh are `__global__` symbols the reference launches itself with
// <<<n * limbs / 128, 128>>>; here they are host launchers of the same names whose first argument is the table handle in
// place of the launch configuration: multiply_rns_poly<<<g, b, 0, s>>>(a, b, modulus, r, n, limbs) becomes
// multiply_rns_poly(tables, a, b, r, limbs, start_modulus_idx, s).
inline void add_rns_poly(const DNTTTable &t, const uint64_t *op1, const uint64_t *op2, uint64_t *result, size_t coeff_mod_size,
                         size_t start_modulus_idx, const cudaStream_t &stream) {
    util::check_pha(pha_add_rns_poly(t.amd(), op1, op2, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void sub_rns_poly(const DNTTTable &t, const uint64_t *op1, const uint64_t *op2, uint64_t *result, size_t coeff_mod_size,
                         size_t start_modulus_idx, const cudaStream_t &stream) {
    util::check_pha(pha_sub_rns_poly(t.amd(), op1, op2, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void negate_rns_poly(const DNTTTable &t, const uint64_t *operand, uint64_t *result, size_t coeff_mod_size, size_t start_modulus_idx,
                            const cudaStream_t &stream) {
    util::check_pha(pha_negate_rns_poly(t.amd(), operand, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void multiply_rns_poly(const DNTTTable &t, const uint64_t *op1, const uint64_t *op2, uint64_t *result, size_t coeff_mod_size,
                              size_t start_modulus_idx, const cudaStream_t &stream) {
    util::check_pha(pha_multiply_rns_poly(t.amd(), op1, op2, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void multiply_and_add_rns_poly(const DNTTTable &t, const uint64_t *op1, const uint64_t *op2, const uint64_t *op3, uint64_t *result,
                                      size_t coeff_mod_size, size_t start_modulus_idx, const cudaStream_t &stream) {
    util::check_pha(pha_multiply_and_add_rns_poly(t.amd(), op1, op2, op3, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void multiply_scalar_rns_poly(const DNTTTable &t, const uint64_t *operand, const uint64_t *scalar, const uint64_t *scalar_shoup,
                                     uint64_t *result, size_t coeff_mod_size, size_t start_modulus_idx, const cudaStream_t &stream) {
    util::check_pha(pha_multiply_scalar_rns_poly(t.amd(), operand, scalar, scalar_shoup, result, coeff_mod_size, start_modulus_idx, stream));
}
inline void tensor_prod_2x2_rns_poly(const DNTTTable &t, const uint64_t *op1, const uint64_t *op2, uint64_t *result, size_t coeff_mod_size,
                                     const cudaStream_t &stream) {
    util::check_pha(pha_tensor_prod_2x2_rns_poly(t.amd(), op1, op2, result, coeff_mod_size, stream));
}
inline void tensor_square_2x2_rns_poly(const DNTTTable &t, const uint64_t *op, uint64_t *result, size_t coeff_mod_size,
                                       const cudaStream_t &stream) {
    util::check_pha(pha_tensor_square_2x2_rns_poly(t.amd(), op, result, coeff_mod_size, stream));
}

}  // namespace phantom
