// pha_context.hip -- host precompute + device tables behind pha_context_t.
//
// Replaces, for the hot path only: PhantomContext's table upload (src/context.cu:170-183,
// include/ntt.cuh:34-129), the SEAL-derived host tables (src/host/ntt.cu:11-56, src/host/rns.cu:282-337,
// 438-497) and the hybrid key-switch part of the DRNSTool constructor (src/rns.cu:66-200).
// Written independently of oracle/ (different inverse / primality / root-search code paths) so the
// two can check each other.  Each prime's tables are computed once and shared by every level
// (the reference recomputes them per ContextData).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

#include "../../include/phantom_amd.h"
#include "pha_internal.h"

namespace pha {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_last_error(const char *what) { g_last_error = what ? what : ""; }
int translate_exception() {
    try {
        throw;
    } catch (const std::invalid_argument &e) {
        set_last_error(e.what());
        return PHA_ERR_INVALID_ARGUMENT;
    } catch (const std::logic_error &e) {
        set_last_error(e.what());
        return PHA_ERR_LOGIC;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return PHA_ERR_RUNTIME;
    } catch (...) {
        set_last_error("unknown error");
        return PHA_ERR_RUNTIME;
    }
}

// ------------------------------------------------------------------------------------------------
// host number theory
// ------------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;

u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
u64 h_powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod(r, a, q);
        a = h_mulmod(a, a, q);
    }
    return r;
}
// extended Euclid (the reference's try_invert_uint_mod is xgcd as well, src/host/numth.cu)
u64 h_invmod(u64 a, u64 q) {
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr) {
        __int128 k = r / nr;
        __int128 tmp = t - k * nt; t = nt; nt = tmp;
        tmp = r - k * nr; r = nr; nr = tmp;
    }
    if (r != 1) throw std::logic_error("value is not invertible modulo q");
    if (t < 0) t += q;
    return (u64)t;
}
u64 h_shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
DModulus h_modulus(u64 q) {
    u128 r = (~(u128)0) / q;  // floor(2^128/q) for q not a power of two
    return DModulus{q, (u64)r, (u64)(r >> 64)};
}
uint32_t h_brev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
bool h_is_prime(u64 n) {
    // deterministic Miller-Rabin, 7-base set valid for all n < 2^64
    if (n < 2) return false;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : small) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    static const u64 bases[] = {2, 325, 9375, 28178, 450775, 9780504, 1795265022};
    for (u64 a : bases) {
        u64 x = h_powmod(a % n, d, n);
        if (x == 0 || x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s; i++) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}
// minimal primitive degree-th root (degree = 2N a power of two): src/host/numth.cu:309-331.
u64 h_minimal_primitive_root(u64 degree, u64 q) {
    if ((q - 1) % degree) throw std::invalid_argument("invalid modulus in try_minimal_primitive_root");
    const u64 quo = (q - 1) / degree;
    u64 root = 0;
    for (u64 g = 3; g < 4096 && !root; g += 2) {  // any generator-ish start; the minimum is start-independent
        u64 r = h_powmod(g, quo, q);
        if (h_powmod(r, degree >> 1, q) == q - 1) root = r;
    }
    if (!root) throw std::invalid_argument("invalid modulus in try_minimal_primitive_root");
    const u64 sq = h_mulmod(root, root, q);
    u64 cur = root, best = root;
    for (u64 i = 0; i < degree / 2; i++) {  // the degree/2 odd powers are all the primitive roots
        if (cur < best) best = cur;
        cur = h_mulmod(cur, sq, q);
    }
    return best;
}

static void get_primes(u64 ntt_size, int bit_size, size_t count, std::vector<u64> &out) {
    // src/host/numth.cu:207-233
    if (bit_size < 2 || bit_size > 61) throw std::invalid_argument("bit_sizes is invalid");
    const u64 factor = 2 * ntt_size;
    u64 value = (u64)1 << bit_size;
    if (value < factor) throw std::logic_error("failed to find enough qualifying primes");
    value = value - factor + 1;
    const u64 lower = (u64)1 << (bit_size - 1);
    while (count > 0 && value > lower) {
        if (h_is_prime(value)) { out.push_back(value); count--; }
        value -= factor;
    }
    if (count > 0) throw std::logic_error("failed to find enough qualifying primes");
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct FpTables {
    std::vector<u64> tw, itw;
    std::vector<u64x2> ninv, w1ninv;
    std::vector<FpInfo> info;
};
static u64 fp_bits(u64 w) { return as_u64((double)w); }  // W as a double (exact: W < 2^50)

// Tables of prime c.primes[row] written at position `i` of the output vectors (i == row for the QP chain;
// auxiliary primes are built into short vectors and appended to the device arrays).
static void build_prime_tables(Context &c, uint32_t row, uint32_t i, std::vector<u64x2> &tw, std::vector<u64x2> &itw,
                               std::vector<u64x2> &ninv, std::vector<u64x2> &w1ninv, FpTables &fp) {
    const u64 q = c.primes[row];
    const size_t n = c.n;
    const u64 psi = h_minimal_primitive_root(2 * n, q);
    const u64 ipsi = h_invmod(psi, q);
    c.roots[row] = psi;
    u64x2 *t = tw.data() + (size_t)i * n, *it = itw.data() + (size_t)i * n;
    u64 pw = 1, ipw = 1;
    for (size_t e = 0; e < n; e++) {  // slot brev(e) holds psi^e (src/host/ntt.cu:27-33)
        const uint32_t k = h_brev((uint32_t)e, (int)c.log_n);
        t[k] = u64x2{pw, h_shoup(pw, q)};
        it[k] = u64x2{ipw, h_shoup(ipw, q)};
        pw = h_mulmod(pw, psi, q);
        ipw = h_mulmod(ipw, ipsi, q);
    }
    const u64 ni = h_invmod((u64)(n % q), q);
    c.n_inv[row] = ni;
    ninv[i] = u64x2{ni, h_shoup(ni, q)};
    const u64 w1 = h_mulmod(it[1].x, ni, q);  // what the reference stores in itwiddle[1] (ntt.cu:53-55)
    w1ninv[i] = u64x2{w1, h_shoup(w1, q)};
    // FP64 tables (q < 2^50 only; see pha_arith.h)
    const bool ok = (q >> 50) == 0;
    const FpMod fm = make_fpmod(q);  // ok: bit 0 usable, bit 1 light forward butterflies, bit 2 light inverse ones
    fp.info[i] = FpInfo{fm.q, fm.qinv, ok ? (1u | (fm.ct_light ? 2u : 0u) | (fm.gs_light ? 4u : 0u)) : 0u, 0u};
    if (ok) {
        u64 *tf = fp.tw.data() + (size_t)i * n, *itf = fp.itw.data() + (size_t)i * n;
        for (size_t k = 0; k < n; k++) {
            tf[k] = fp_bits(t[k].x);
            itf[k] = fp_bits(it[k].x);
        }
        fp.ninv[i] = u64x2{fp_bits(ni), 0};
        fp.w1ninv[i] = u64x2{fp_bits(w1), 0};
    }
}

#if defined(PHA_EXPERIMENTS)
// (experiments library only: the one-launch NTT needs it; the product library neither launches the census nor allocates counters)
// XCD placement census: the first workgroup of each class b % 8 records its XCC_ID, the others compare
__global__ void xcd_census_kernel(uint32_t *cls, uint32_t *mismatch) {
    if (threadIdx.x == 0) {
        uint32_t id;
#if defined(__gfx942__) || defined(__gfx950__)
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
#else
        id = blockIdx.x;   // no XCC_ID register: report the classes as if round-robin placement held (one die)
#endif
        const uint32_t mine = (id & 7u) + 1;
        const uint32_t seen = atomicCAS(cls + (blockIdx.x & 7u), 0u, mine);
        if (seen != 0 && seen != mine) atomicAdd(mismatch, 1u);
    }
}
static bool xcd_placement_is_round_robin() {
    DevBuf<uint32_t> d(9);
    for (int rep = 0; rep < 3; rep++) {
        PHA_HIP(hipMemset(d.p, 0, 8 * sizeof(uint32_t)));
        if (rep == 0) PHA_HIP(hipMemset(d.p + 8, 0, sizeof(uint32_t)));
        hipLaunchKernelGGL(xcd_census_kernel, dim3(4096 + 8 * rep + 3), dim3(64), 0, 0, d.p, d.p + 8);
        PHA_HIP(hipGetLastError());
        PHA_HIP(hipDeviceSynchronize());
    }
    uint32_t h[9];
    PHA_HIP(hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost));
    return h[8] == 0;
}

// Lazily, the first time the one-launch transform is asked for (pha_set_tuning key 0 bit 9 / key 4): three census launches on
// the null stream and the counter pool.  Not to be triggered inside a stream capture.
bool Context::xcd_placement_round_robin() {
    std::lock_guard<std::mutex> lk(mu);
    if (xcd_checked) return xcd_round_robin;
    DeviceGuard on_device(device);
    xcd_round_robin = xcd_placement_is_round_robin();
    if (xcd_round_robin) {
        flag_pool.alloc(Context::kFlagArenas * (2 * Context::kFlagUnits + 16));
        PHA_HIP(hipMemset(flag_pool.p, 0, flag_pool.count * sizeof(uint32_t)));
    }
    xcd_checked = true;
    return xcd_round_robin;
}
#endif  // PHA_EXPERIMENTS

static void context_init(Context &c, uint32_t log_n, const uint64_t *primes, uint32_t size_qp, uint32_t size_p,
                         int device) {
    if (log_n < 12 || log_n > 17) throw std::invalid_argument("poly_modulus_degree is invalid (2^12..2^17 supported)");
    if (size_qp < 1 || size_qp > 64 || size_p >= size_qp) throw std::invalid_argument("RNSBase is invalid");
    c.device = device;
    DeviceGuard on_device(device);   // the caller's current device is restored when the context has been built
    {
        hipDeviceProp_t prop;
        PHA_HIP(hipGetDeviceProperties(&prop, device));
        c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    c.log_n = log_n;
    c.n = (size_t)1 << log_n;
    c.size_qp = size_qp;
    c.size_p = size_p;
    c.size_q = size_qp - size_p;
    c.primes.assign(primes, primes + size_qp);
    c.roots.resize(size_qp);
    c.n_inv.resize(size_qp);
    c.mods.resize(size_qp);
    for (uint32_t i = 0; i < size_qp; i++) {
        const u64 q = c.primes[i];
        if (q >> 61) throw std::invalid_argument("modulus exceeds 61 bits");
        if (!h_is_prime(q) || (q - 1) % (2 * c.n)) throw std::invalid_argument("modulus is not an NTT prime");
        for (uint32_t j = 0; j < i; j++)
            if (c.primes[j] == q) throw std::invalid_argument("coeff_modulus is not coprime");
        c.mods[i] = h_modulus(q);
    }
    std::vector<u64x2> tw((size_t)size_qp * c.n), itw((size_t)size_qp * c.n), ninv(size_qp), w1ninv(size_qp);
    FpTables fp;
    fp.tw.assign((size_t)size_qp * c.n, 0);
    fp.itw.assign((size_t)size_qp * c.n, 0);
    fp.ninv.assign(size_qp, u64x2{0, 0});
    fp.w1ninv.assign(size_qp, u64x2{0, 0});
    fp.info.resize(size_qp);
    {
        const unsigned nthreads = std::max(1u, std::min(std::thread::hardware_concurrency(), size_qp));
        std::vector<std::thread> pool;
        std::vector<std::string> errs(nthreads);
        for (unsigned t = 0; t < nthreads; t++)
            pool.emplace_back([&, t] {
                try {
                    for (uint32_t i = t; i < size_qp; i += nthreads) build_prime_tables(c, i, i, tw, itw, ninv, w1ninv, fp);
                } catch (const std::exception &e) { errs[t] = e.what(); }
            });
        for (auto &th : pool) th.join();
        for (auto &e : errs)
            if (!e.empty()) throw std::invalid_argument(e);
    }
    c.d_mod.upload(c.mods);
    c.d_tw.upload(tw);
    c.d_itw.upload(itw);
    c.d_ninv.upload(ninv);
    c.d_w1ninv.upload(w1ninv);
    c.d_twf.upload(fp.tw);
    c.d_itwf.upload(fp.itw);
    c.d_ninvf.upload(fp.ninv);
    c.d_w1ninvf.upload(fp.w1ninv);
    c.d_fpinfo.upload(fp.info);
    c.rows = size_qp;
}

// Append auxiliary moduli to the context's prime / table arrays: `ntt_primes` get full NTT tables, then one more
// modulus without tables (m_tilde = 2^32 of BEHZ).  Returns the row of the first one.  Nothing may be in flight.
uint32_t Context::add_aux_moduli(const std::vector<u64> &ntt_primes, u64 table_less_modulus) {
    PHA_HIP(hipSetDevice(device));
    PHA_HIP(hipDeviceSynchronize());
    const uint32_t first = rows, cnt = (uint32_t)ntt_primes.size();
    for (u64 q : ntt_primes) {
        if (q >> 61) throw std::invalid_argument("modulus exceeds 61 bits");
        if (!h_is_prime(q) || (q - 1) % (2 * n)) throw std::invalid_argument("modulus is not an NTT prime");
        for (u64 p : primes)
            if (p == q) throw std::invalid_argument("coeff_modulus is not coprime");
        primes.push_back(q);
        mods.push_back(h_modulus(q));
    }
    const uint32_t extra = table_less_modulus ? 1u : 0u;
    if (extra) {
        primes.push_back(table_less_modulus);
        mods.push_back(h_modulus(table_less_modulus));
    }
    roots.resize(primes.size());
    n_inv.resize(primes.size());
    const uint32_t total = cnt + extra;
    std::vector<u64x2> tw((size_t)total * n, u64x2{0, 0}), itw((size_t)total * n, u64x2{0, 0}), ninv(total, u64x2{0, 0}),
        w1ninv(total, u64x2{0, 0});
    FpTables fp;
    fp.tw.assign((size_t)total * n, 0);
    fp.itw.assign((size_t)total * n, 0);
    fp.ninv.assign(total, u64x2{0, 0});
    fp.w1ninv.assign(total, u64x2{0, 0});
    fp.info.assign(total, FpInfo{0.0, 0.0, 0u, 0u});
    for (uint32_t i = 0; i < cnt; i++) build_prime_tables(*this, first + i, i, tw, itw, ninv, w1ninv, fp);
    d_mod.append(std::vector<DModulus>(mods.begin() + first, mods.end()));
    d_tw.append(tw);
    d_itw.append(itw);
    d_ninv.append(ninv);
    d_w1ninv.append(w1ninv);
    d_twf.append(fp.tw);
    d_itwf.append(fp.itw);
    d_ninvf.append(fp.ninv);
    d_w1ninvf.append(fp.w1ninv);
    d_fpinfo.append(fp.info);
    rows += total;
    return first;
}

void describe_conv(const BConv &b, DevBuf<BConvDev> &out) {
    out.upload({BConvDev{b.hat_inv.p, b.d_iprime.p, b.d_oprime.p, b.mat.p, b.mat30.p, b.mont ? b.oninv.p : nullptr, b.isz, b.osz, 0xffffffffu, 0, 0, 0, b.row_pad, b.r90 ? 1u : 0u}});
}

// DRNSTool constructor, HPS multiply part (src/rns.cu:687-790; converters src/host/rns.cu:282-337,438-466).
Hps &Context::hps() {
    std::lock_guard<std::mutex> lk(mu);
    if (hps_tool) return *hps_tool;
    if (!plain_t) throw std::invalid_argument("bfv multiply needs a plain modulus (pha_context_set_plain_modulus)");
    auto h = std::make_unique<Hps>();
    const uint32_t sq = size_q, sr = size_q + 1;
    h->size_q = sq;
    h->size_r = sr;
    // get_primes_below(n, min q, |R|) src/host/numth.cu:235-263
    u64 minq = primes[0];
    for (uint32_t i = 1; i < sq; i++) minq = std::min(minq, primes[i]);
    std::vector<u64> r;
    {
        const u64 factor = 2 * (u64)n, lower = (u64)1 << (63 - __builtin_clzll(minq));
        for (u64 v = minq - factor; r.size() < sr && v > lower; v -= factor)
            if (h_is_prime(v)) r.push_back(v);
        if (r.size() < sr) throw std::logic_error("failed to find enough qualifying primes 2");
    }
    h->aux0 = add_aux_moduli(r, 0);
    std::vector<uint32_t> iq, ir;
    for (uint32_t i = 0; i < sq; i++) iq.push_back(i);
    for (uint32_t j = 0; j < sr; j++) ir.push_back(h->aux0 + j);
    build_bconv(*this, h->q_to_r, iq, ir);
    build_bconv(*this, h->r_to_q, ir, iq);
    describe_conv(h->q_to_r, h->d_q_to_r);
    describe_conv(h->r_to_q, h->d_r_to_q);
    auto prod_mod = [&](const std::vector<uint32_t> &rows_, u64 m) {
        u64 p = 1 % m;
        for (uint32_t row : rows_) p = h_mulmod(p, primes[row] % m, m);
        return p;
    };
    {
        std::vector<double> qi(sq), ri(sr);
        for (uint32_t i = 0; i < sq; i++) qi[i] = 1.0 / (double)primes[i];
        for (uint32_t j = 0; j < sr; j++) ri[j] = 1.0 / (double)r[j];
        h->q_inv.upload(qi);
        h->r_inv.upload(ri);
        std::vector<u64> aq((size_t)(sq + 1) * sr), ar((size_t)(sr + 1) * sq);
        for (uint32_t j = 0; j < sr; j++) {
            const u64 qm = prod_mod(iq, r[j]);
            for (uint32_t a = 0; a <= sq; a++) aq[(size_t)a * sr + j] = h_mulmod(a, qm, r[j]);
        }
        for (uint32_t i = 0; i < sq; i++) {
            const u64 rm = prod_mod(ir, primes[i]);
            for (uint32_t a = 0; a <= sr; a++) ar[(size_t)a * sq + i] = h_mulmod(a, rm, primes[i]);
        }
        h->alpha_q_mod_r.upload(aq);
        h->alpha_r_mod_q.upload(ar);
    }
    {   // t/Q scale-and-round tables (rns.cu:727-790): x_i = t * R * (S / s_i)^-1 mod s_i as big integers, S = Q || R
        std::vector<u64> s_all(primes.begin(), primes.begin() + sq);
        s_all.insert(s_all.end(), r.begin(), r.end());
        std::vector<double> frac(sq);
        std::vector<u64> tab((size_t)sr * (sq + 1));
        auto mul_small = [](std::vector<u64> &b, u64 m) {
            u64 carry = 0;
            for (auto &w : b) { const u128 t = (u128)w * m + carry; w = (u64)t; carry = (u64)(t >> 64); }
            if (carry) b.push_back(carry);
        };
        auto mod_small = [](const std::vector<u64> &b, u64 m) {
            u128 rem = 0;
            for (size_t i = b.size(); i-- > 0;) rem = ((rem << 64) | b[i]) % m;
            return (u64)rem;
        };
        auto div_small = [](std::vector<u64> &b, u64 m) {
            u128 rem = 0;
            for (size_t i = b.size(); i-- > 0;) { const u128 cur = (rem << 64) | b[i]; b[i] = (u64)(cur / m); rem = cur % m; }
        };
        for (uint32_t i = 0; i < sq + sr; i++) {
            u64 hat = 1;
            for (uint32_t k = 0; k < sq + sr; k++)
                if (k != i) hat = h_mulmod(hat, s_all[k] % s_all[i], s_all[i]);
            const u64 shat_inv = h_invmod(hat, s_all[i]);
            std::vector<u64> x{1};
            for (u64 rj : r) mul_small(x, rj);
            mul_small(x, plain_t);
            mul_small(x, shat_inv);
            if (i < sq) frac[i] = (double)mod_small(x, s_all[i]) / (double)s_all[i];
            div_small(x, s_all[i]);
            if (i < sq) {
                for (uint32_t j = 0; j < sr; j++) tab[(size_t)j * (sq + 1) + i] = mod_small(x, r[j]);
            } else {
                tab[(size_t)(i - sq) * (sq + 1) + sq] = mod_small(x, r[i - sq]);
            }
        }
        h->frac.upload(frac);
        h->div_mod_r.upload(tab);
    }
    hps_tool = std::move(h);
    return *hps_tool;
}

// DRNSTool constructor, hps_overq part (src/rns.cu:792-885), and for size_ql < |Q| the hps_overq_leveled part with
// |Q| - size_ql levels dropped (:897-975).
HpsQ &Context::hps_overq(uint32_t size_ql) {
    if (size_ql == 0) size_ql = size_q;
    if (size_ql < 1 || size_ql > size_q) throw std::invalid_argument("RNSBase is invalid");
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = hpsq_tools.find(size_ql);
        if (it != hpsq_tools.end()) return *it->second;
    }
    Hps &base = hps();                       // its base R starts with the primes of every Rl (same get_primes_below walk)
    std::lock_guard<std::mutex> lk(mu);
    {
        auto it = hpsq_tools.find(size_ql);
        if (it != hpsq_tools.end()) return *it->second;
    }
    auto h = std::make_unique<HpsQ>();
    const uint32_t sq = size_ql, sr = size_ql, drop = size_q - size_ql;
    h->size_q = sq;
    h->size_r = sr;
    h->size_q_full = size_q;
    h->drop = drop;
    h->aux0 = base.aux0;
    std::vector<uint32_t> iq, ir, iq_full;
    std::vector<u64> r(sr);
    for (uint32_t i = 0; i < sq; i++) iq.push_back(i);
    for (uint32_t i = 0; i < size_q; i++) iq_full.push_back(i);
    for (uint32_t j = 0; j < sr; j++) { ir.push_back(h->aux0 + j); r[j] = primes[h->aux0 + j]; }
    build_bconv(*this, h->q_to_r, iq, ir);
    build_bconv(*this, h->r_to_q, ir, iq);
    build_bconv_var1(*this, h->q_to_r_var1, drop ? iq_full : iq, ir);
    describe_conv(h->q_to_r, h->d_q_to_r);
    describe_conv(h->r_to_q, h->d_r_to_q);
    describe_conv(h->q_to_r_var1, h->d_q_to_r_var1);
    auto prod_mod = [&](const std::vector<uint32_t> &rows_, u64 m) {
        u64 p = 1 % m;
        for (uint32_t row : rows_) p = h_mulmod(p, primes[row] % m, m);
        return p;
    };
    {
        std::vector<double> qi(sq), ri(sr);
        for (uint32_t i = 0; i < sq; i++) qi[i] = 1.0 / (double)primes[i];
        for (uint32_t j = 0; j < sr; j++) ri[j] = 1.0 / (double)r[j];
        h->q_inv.upload(qi);
        h->r_inv.upload(ri);
        std::vector<u64> aq((size_t)(sq + 1) * sr), ar((size_t)(sr + 1) * sq);
        for (uint32_t j = 0; j < sr; j++) {
            const u64 qm = prod_mod(iq, r[j]);
            for (uint32_t a = 0; a <= sq; a++) aq[(size_t)a * sr + j] = h_mulmod(a, qm, r[j]);
        }
        for (uint32_t i = 0; i < sq; i++) {
            const u64 rm = prod_mod(ir, primes[i]);
            for (uint32_t a = 0; a <= sr; a++) ar[(size_t)a * sq + i] = h_mulmod(a, rm, primes[i]);
        }
        h->alpha_q_mod_r.upload(aq);
        h->alpha_r_mod_q.upload(ar);
    }
    auto mul_small = [](std::vector<u64> &b, u64 m) {
        u64 carry = 0;
        for (auto &w : b) { const u128 t = (u128)w * m + carry; w = (u64)t; carry = (u64)(t >> 64); }
        if (carry) b.push_back(carry);
    };
    auto mod_small = [](const std::vector<u64> &b, u64 m) {
        u128 rem = 0;
        for (size_t i = b.size(); i-- > 0;) rem = ((rem << 64) | b[i]) % m;
        return (u64)rem;
    };
    auto div_small = [](std::vector<u64> &b, u64 m) {
        u128 rem = 0;
        for (size_t i = b.size(); i-- > 0;) { const u128 cur = (rem << 64) | b[i]; b[i] = (u64)(cur / m); rem = cur % m; }
    };
    // scale-and-round tables of the shape [Ql][extra + 1] + fractions [extra]: x_i = factor * prod(Ql) * (S / s_i)^-1 mod
    // s_i as big integers over S = Ql || extra primes
    auto scale_tables = [&](const std::vector<u64> &extra, u64 factor, std::vector<double> &frac, std::vector<u64> &tab) {
        const uint32_t ne = (uint32_t)extra.size();
        std::vector<u64> s_all(primes.begin(), primes.begin() + sq);
        s_all.insert(s_all.end(), extra.begin(), extra.end());
        frac.assign(ne, 0.0);
        tab.assign((size_t)sq * (ne + 1), 0);
        for (uint32_t i = 0; i < sq + ne; i++) {
            u64 hat = 1;
            for (uint32_t k = 0; k < sq + ne; k++)
                if (k != i) hat = h_mulmod(hat, s_all[k] % s_all[i], s_all[i]);
            const u64 shat_inv = h_invmod(hat, s_all[i]);
            std::vector<u64> x{1};
            for (uint32_t k = 0; k < sq; k++) mul_small(x, primes[k]);
            if (factor != 1) mul_small(x, factor);
            mul_small(x, shat_inv);
            if (i >= sq) frac[i - sq] = (double)mod_small(x, s_all[i]) / (double)s_all[i];
            div_small(x, s_all[i]);
            if (i >= sq) {
                for (uint32_t l = 0; l < sq; l++) tab[(size_t)l * (ne + 1) + (i - sq)] = mod_small(x, primes[l]);
            } else {
                tab[(size_t)i * (ne + 1) + ne] = mod_small(x, primes[i]);
            }
        }
    };
    {   // t/Rl scale-and-round (rns.cu:836-885)
        std::vector<double> frac;
        std::vector<u64> tab;
        scale_tables(r, plain_t, frac, tab);
        h->frac.upload(frac);
        h->div_mod_q.upload(tab);
    }
    if (drop) {   // Ql/Q scale-and-round (rns.cu:918-972) and the expansion constants (base_Ql_to_QlDrop_conv.PModq)
        std::vector<u64> dropped(primes.begin() + sq, primes.begin() + size_q);
        std::vector<double> frac;
        std::vector<u64> tab;
        scale_tables(dropped, 1, frac, tab);
        h->frac_drop.upload(frac);
        h->div_mod_q_drop.upload(tab);
        std::vector<u64> dm(sq), dms(sq);
        for (uint32_t i = 0; i < sq; i++) {
            u64 p = 1;
            for (u64 d : dropped) p = h_mulmod(p, d % primes[i], primes[i]);
            dm[i] = p;
            dms[i] = h_shoup(p, primes[i]);
        }
        h->drop_mod_q.upload(dm);
        h->drop_mod_q_shoup.upload(dms);
    }
    HpsQ &ref = *h;
    hpsq_tools[size_ql] = std::move(h);
    return ref;
}

// DRNSTool constructor, BEHZ part (src/rns.cu:392-560).  Needs the plain modulus; top data level only.
Behz &Context::behz() {
    std::lock_guard<std::mutex> lk(mu);
    if (behz_tool) return *behz_tool;
    if (!plain_t) throw std::invalid_argument("bfv multiply needs a plain modulus (pha_context_set_plain_modulus)");
    auto b = std::make_unique<Behz>();
    b->size_q = size_q;
    b->plain_t = plain_t;
    // |prod(q)| exactly (get_significant_bit_count_uint of the big modulus)
    std::vector<u64> acc{1};
    for (uint32_t i = 0; i < size_q; i++) {
        u64 carry = 0;
        for (auto &w : acc) {
            const u128 t = (u128)w * primes[i] + carry;
            w = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (carry) acc.push_back(carry);
    }
    const int total_bits = (int)(acc.size() - 1) * 64 + (64 - __builtin_clzll(acc.back()));
    const int t_bits = 64 - __builtin_clzll(plain_t);
    b->size_b = size_q + ((32 + t_bits + total_bits >= 61 * (int)size_q + 61) ? 1 : 0);  // rns.cu:398-406
    b->size_bsk = b->size_b + 1;
    // get_primes(n, 61, size_b + 1): m_sk first, then B (rns.cu:413-419); descending from 2^61 - 2n + 1 in steps of 2n
    std::vector<u64> found;
    for (u64 v = ((u64)1 << 61) - 2 * n + 1; found.size() < b->size_b + 1 && v > ((u64)1 << 60); v -= 2 * n)
        if (h_is_prime(v)) found.push_back(v);
    if (found.size() < b->size_b + 1) throw std::logic_error("failed to find enough qualifying primes");
    b->m_sk = found[0];
    std::vector<u64> bsk(found.begin() + 1, found.end());
    bsk.push_back(b->m_sk);
    const u64 m_tilde = (u64)1 << 32;
    b->aux0 = add_aux_moduli(bsk, m_tilde);
    const uint32_t mt_row = b->aux0 + b->size_bsk;
    std::vector<uint32_t> iq, ib, obskmt, obsk, omsk{b->aux0 + b->size_b};
    for (uint32_t i = 0; i < size_q; i++) iq.push_back(i);
    for (uint32_t i = 0; i < b->size_b; i++) ib.push_back(b->aux0 + i);
    for (uint32_t i = 0; i < b->size_bsk; i++) obsk.push_back(b->aux0 + i);
    obskmt = obsk;
    obskmt.push_back(mt_row);
    build_bconv(*this, b->q_to_bskmt, iq, obskmt);
    build_bconv(*this, b->q_to_bsk, iq, obsk);
    build_bconv(*this, b->b_to_q, ib, iq);
    build_bconv(*this, b->b_to_msk, ib, omsk);
    describe_conv(b->q_to_bskmt, b->d_q_to_bskmt);
    describe_conv(b->q_to_bsk, b->d_q_to_bsk);
    describe_conv(b->b_to_q, b->d_b_to_q);
    describe_conv(b->b_to_msk, b->d_b_to_msk);
    auto pair = [](u64 v, u64 q) { return u64x2{v, h_shoup(v, q)}; };
    auto prod_mod = [&](uint32_t first, uint32_t cnt, u64 m) {
        u64 p = 1 % m;
        for (uint32_t i = 0; i < cnt; i++) p = h_mulmod(p, primes[first + i] % m, m);
        return p;
    };
    {   // m_tilde * qhat_i^-1 mod q_i (:450-465)
        std::vector<u64x2> hi(size_q), v(size_q);
        PHA_HIP(hipMemcpy(hi.data(), b->q_to_bsk.hat_inv.p, size_q * sizeof(u64x2), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < size_q; i++) v[i] = pair(h_mulmod(m_tilde % primes[i], hi[i].x, primes[i]), primes[i]);
        b->mt_qhatinv.upload(v);
        // the Q -> Bsk u {m_tilde} converter is only ever fed m_tilde * x (BEHZ_mul_1): its phase-1 factors carry the m_tilde, so
        // the conversion kernel scales on load and no scaled copy of the input is written (describe_conv keeps this pointer)
        b->q_to_bskmt.hat_inv.upload(v);
    }
    {
        std::vector<u64x2> ipq(b->size_bsk), imt(b->size_bsk);
        std::vector<u64> pq(b->size_bsk), tb(b->size_bsk), tbs(b->size_bsk);
        for (uint32_t j = 0; j < b->size_bsk; j++) {
            const u64 p = primes[b->aux0 + j];
            pq[j] = prod_mod(0, size_q, p);                         // :548-553
            ipq[j] = pair(h_invmod(pq[j], p), p);                   // :508-518
            imt[j] = pair(h_invmod(m_tilde % p, p), p);             // :537-546
            tb[j] = plain_t;                                        // :467-479
            tbs[j] = h_shoup(plain_t, p);
        }
        b->inv_prod_q_mod_bsk.upload(ipq);
        b->inv_mt_mod_bsk.upload(imt);
        b->prod_q_mod_bsk.upload(pq);
        b->t_bsk.upload(tb);
        b->t_bsk_shoup.upload(tbs);
    }
    {
        std::vector<u64> pb(size_q), tq(size_q), tqs(size_q);
        for (uint32_t i = 0; i < size_q; i++) {
            pb[i] = prod_mod(b->aux0, b->size_b, primes[i]);
            tq[i] = plain_t;
            tqs[i] = h_shoup(plain_t, primes[i]);
        }
        b->prod_b_mod_q.upload(pb);
        b->t_q.upload(tq);
        b->t_q_shoup.upload(tqs);
    }
    const u64 inv_q_mt = h_invmod(prod_mod(0, size_q, m_tilde), m_tilde);
    b->neg_inv_prod_q_mod_mt = pair((m_tilde - inv_q_mt) % m_tilde, m_tilde);
    b->inv_prod_b_mod_msk = pair(h_invmod(prod_mod(b->aux0, b->size_b, b->m_sk), b->m_sk), b->m_sk);
    behz_tool = std::move(b);
    return *behz_tool;
}

// q-hat_i^-1 mod q_i and q-hat_i mod p_j for an (ibase -> obase) converter: src/host/rns.cu:282-337,438-457
// upload converter constants: hat_inv [isz] (value, Shoup), mat [osz][isz]
static void upload_bconv(Context &c, BConv &b, const std::vector<uint32_t> &ip, const std::vector<uint32_t> &op,
                         const std::vector<u64x2> &hat_inv, const std::vector<u64> &mat, uint32_t family_isz = 0) {
    b.isz = (uint32_t)ip.size();
    b.osz = (uint32_t)op.size();
    b.iprime = ip;
    b.oprime = op;
    // which carry-free split the kernel may use with these constants (BConv::split_kind)
    auto bits_of = [](u64 v) { int n = 0; while (v) { n++; v >>= 1; } return n; };
    int by = 0, bm = 0;       // residues of the input primes are below 2^by, matrix entries (residues of the output moduli) below 2^bm
    for (uint32_t row : ip) by = std::max(by, bits_of(c.primes[row] - 1));
    for (uint32_t row : op) bm = std::max(bm, bits_of(c.primes[row] - 1));
    uint32_t sm = 30;         // the matrix entries are cut at this bit
    b.split_kind = 0;
    b.row_pad = kBcRowPad;
    const uint32_t widest = std::max(b.isz, family_isz);   // converters launched together (mod-up digits) take one kind
    if (widest <= (uint32_t)kBcRowPad && by <= 60 && bm <= 60) b.split_kind = 1;
    else if (widest <= 32 && by <= 60 && bm <= 62) { b.split_kind = 2; sm = 31; }
    else if (widest <= 32 && by <= 62 && bm <= 60) { b.split_kind = 3; sm = 30; }
    if (b.split_kind >= 2 && b.isz > (uint32_t)kBcRowPad) b.row_pad = 32;
    // Montgomery form for the split-accumulator kernel: rows hold qhat_i * 2^64 mod p_j, so that REDC of the accumulated
    // sum gives sum_i y_i * qhat_i mod p_j directly (valid for odd p_j; the BEHZ converter with m_tilde = 2^32 keeps Barrett)
    // and for sum_i y_i * entry < 2^64 p_j: 16 inputs of 60 bits, or in general sum_i q_i < 2^64
    unsigned __int128 sum_q = 0;
    for (uint32_t row : ip) sum_q += c.primes[row];
    b.mont = b.split_kind == 1 || (b.split_kind != 0 && (sum_q >> 64) == 0);
    for (uint32_t j = 0; j < b.osz; j++) b.mont = b.mont && (c.primes[op[j]] & 1);
    // r06: with at most 15 inputs and the 30 / 30 cuts the kernels reduce word by word in base 2^30 straight from the four split
    // accumulators (mont_redc90_split, pha_arith.h): the rows then carry 2^90.  A sixteenth term would overflow its first step.
    b.r90 = b.mont && b.split_kind == 1 && widest <= 15;
    std::vector<u64> oninv(b.osz, 0);
    if (b.mont)
        for (uint32_t j = 0; j < b.osz; j++) {
            const u64 pj = c.primes[op[j]];
            u64 inv = pj;                                        // Newton: inv * pj = 1 mod 2^(3 * 2^k)
            for (int it = 0; it < 6; it++) inv *= 2 - pj * inv;
            oninv[j] = 0 - inv;
        }
    std::vector<uint32_t> mat30((size_t)b.osz * b.row_pad * 2, 0u);
    if (b.isz <= b.row_pad)
        for (uint32_t j = 0; j < b.osz; j++)
            for (uint32_t i = 0; i < b.isz; i++) {
                u64 m = mat[(size_t)j * b.isz + i];
                if (b.mont) {
                    const u64 pj = c.primes[op[j]];
                    m = (u64)((((unsigned __int128)m) << 64) % pj);
                    if (b.r90) m = (u64)((((unsigned __int128)m) << 26) % pj);   // 2^64 * 2^26 = 2^90
                }
                mat30[((size_t)j * b.row_pad + i) * 2] = (uint32_t)(m & ((1u << sm) - 1));
                mat30[((size_t)j * b.row_pad + i) * 2 + 1] = (uint32_t)(m >> sm);
            }
    b.hat_inv.upload(hat_inv);
    b.mat.upload(mat);
    b.mat30.upload(mat30);
    b.oninv.upload(oninv);
    b.d_iprime.upload(ip);
    b.d_oprime.upload(op);
}

// out_scale (optional, [osz]): row j of the matrix is multiplied by out_scale[j] mod p_j, i.e. the converter delivers
// out_scale[j] * (converted value) mod p_j at no extra cost (pha_keyswitch_rescale: P^-1 mod q_j)
void build_bconv(Context &c, BConv &b, const std::vector<uint32_t> &ip, const std::vector<uint32_t> &op,
                 const std::vector<u64> *out_scale, uint32_t family_isz) {
    const uint32_t isz = (uint32_t)ip.size(), osz = (uint32_t)op.size();
    std::vector<u64x2> hat_inv(isz);
    for (uint32_t i = 0; i < isz; i++) {
        const u64 qi = c.primes[ip[i]];
        u64 h = 1;
        for (uint32_t k = 0; k < isz; k++)
            if (k != i) h = h_mulmod(h, c.primes[ip[k]] % qi, qi);
        const u64 inv = h_invmod(h, qi);
        hat_inv[i] = u64x2{inv, h_shoup(inv, qi)};
    }
    std::vector<u64> mat((size_t)osz * isz);
    for (uint32_t j = 0; j < osz; j++) {
        const u64 pj = c.primes[op[j]];
        for (uint32_t i = 0; i < isz; i++) {
            u64 h = out_scale ? (*out_scale)[j] % pj : 1;
            for (uint32_t k = 0; k < isz; k++)
                if (k != i) h = h_mulmod(h, c.primes[ip[k]] % pj, pj);
            mat[(size_t)j * isz + i] = h;
        }
    }
    upload_bconv(c, b, ip, op, hat_inv, mat, family_isz);
}

// bConv_BEHZ_var1 (src/rns_bconv.cu:231-246, constants src/host/rns.cu:469-496): the quotient-style conversion
// x -> about P * x / Q in base P: phase 1 multiplies by -P * qhat_i^-1 mod q_i, the matrix is q_i^-1 mod p_j.
void build_bconv_var1(Context &c, BConv &b, const std::vector<uint32_t> &ip, const std::vector<uint32_t> &op) {
    const uint32_t isz = (uint32_t)ip.size(), osz = (uint32_t)op.size();
    std::vector<u64x2> hat_inv(isz);
    for (uint32_t i = 0; i < isz; i++) {
        const u64 qi = c.primes[ip[i]];
        u64 h = 1, pm = 1;
        for (uint32_t k = 0; k < isz; k++)
            if (k != i) h = h_mulmod(h, c.primes[ip[k]] % qi, qi);
        for (uint32_t j = 0; j < osz; j++) pm = h_mulmod(pm, c.primes[op[j]] % qi, qi);
        const u64 v = qi - h_mulmod(pm, h_invmod(h, qi), qi);
        hat_inv[i] = u64x2{v, h_shoup(v, qi)};
    }
    std::vector<u64> mat((size_t)osz * isz);
    for (uint32_t j = 0; j < osz; j++) {
        const u64 pj = c.primes[op[j]];
        for (uint32_t i = 0; i < isz; i++) mat[(size_t)j * isz + i] = h_invmod(c.primes[ip[i]] % pj, pj);
    }
    upload_bconv(c, b, ip, op, hat_inv, mat);
}

Tool &Context::tool(uint32_t size_ql) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = tools.find(size_ql);
    if (it != tools.end()) return *it->second;
    if (size_ql < 1 || size_ql > size_q) throw std::invalid_argument("RNSBase is invalid");
    PHA_HIP(hipSetDevice(device));
    auto t = std::make_unique<Tool>();
    t->size_ql = size_ql;
    t->alpha = size_p;
    t->size_qlp = size_ql + size_p;
    for (uint32_t i = 0; i < size_ql; i++) t->qlp_prime.push_back(i);
    for (uint32_t i = 0; i < size_p; i++) t->qlp_prime.push_back(size_q + i);
    t->d_qlp_prime.upload(t->qlp_prime);
    // rescale: q_last^-1 mod q_i (rns.cu:66-80)
    if (size_ql > 1) {
        std::vector<u64> v(size_ql - 1), vs(size_ql - 1);
        std::vector<u64x2> v2(size_ql - 1);
        for (uint32_t i = 0; i + 1 < size_ql; i++) {
            v[i] = h_invmod(primes[size_ql - 1] % primes[i], primes[i]);
            vs[i] = h_shoup(v[i], primes[i]);
            v2[i] = u64x2{v[i], vs[i]};
        }
        t->inv_q_last.upload(v);
        t->inv_q_last_shoup.upload(vs);
        t->inv_q_last2.upload(v2);
    }
    if (size_p) {
        // P^-1 mod q_i (rns.cu:110-123)
        std::vector<u64> v(size_ql), vs(size_ql);
        std::vector<u64x2> v2(size_ql);
        for (uint32_t i = 0; i < size_ql; i++) {
            u64 p = 1;
            for (uint32_t k = 0; k < size_p; k++) p = h_mulmod(p, primes[size_q + k] % primes[i], primes[i]);
            v[i] = h_invmod(p, primes[i]);
            vs[i] = h_shoup(v[i], primes[i]);
            v2[i] = u64x2{v[i], vs[i]};
        }
        t->pinv.upload(v);
        t->pinv_shoup.upload(vs);
        t->pinv2.upload(v2);
        {   // P mod q_i (bigP_mod_q, rns.cu:110-123): key generation and the bgv mod-down
            std::vector<u64x2> pm(size_ql);
            for (uint32_t i = 0; i < size_ql; i++) {
                u64 p = 1;
                for (uint32_t k = 0; k < size_p; k++) p = h_mulmod(p, primes[size_q + k] % primes[i], primes[i]);
                pm[i] = u64x2{p, h_shoup(p, primes[i])};
            }
            t->p_mod_q2.upload(pm);
        }
        // digits (rns.cu:152-190)
        t->beta = (size_ql + t->alpha - 1) / t->alpha;
        t->digit.resize(t->beta);
        std::vector<u64> phi(size_ql), phis(size_ql);
        for (uint32_t b = 0; b < t->beta; b++) {
            const uint32_t s = t->alpha * b;
            const uint32_t len = (b == t->beta - 1) ? size_ql - t->alpha * (t->beta - 1) : t->alpha;
            std::vector<uint32_t> ip, op;
            for (uint32_t j = 0; j < t->size_qlp; j++) {
                if (j >= s && j < s + len) ip.push_back(t->qlp_prime[j]);
                else op.push_back(t->qlp_prime[j]);
            }
            build_bconv(*this, t->digit[b], ip, op, nullptr, t->alpha);
            std::vector<u64x2> hi(len);
            PHA_HIP(hipMemcpy(hi.data(), t->digit[b].hat_inv.p, len * sizeof(u64x2), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < len; i++) { phi[s + i] = hi[i].x; phis[s + i] = hi[i].y; }
        }
        t->part_hat_inv.upload(phi);
        t->part_hat_inv_shoup.upload(phis);
        // P -> Ql (rns.cu:196-198)
        std::vector<uint32_t> ip, op;
        for (uint32_t i = 0; i < size_p; i++) ip.push_back(size_q + i);
        for (uint32_t i = 0; i < size_ql; i++) op.push_back(i);
        build_bconv(*this, t->p_to_ql, ip, op);
        build_bconv(*this, t->p_to_ql_pinv, ip, op, &v);   // the same converter delivering P^-1 * (.) mod q_j (key switch + rescale)
        {   // the P -> Ql converter's phase-1 factors, addressable by the limb index of a [Ql || P] buffer
            std::vector<u64x2> hi(size_p);
            PHA_HIP(hipMemcpy(hi.data(), t->p_to_ql.hat_inv.p, size_p * sizeof(u64x2), hipMemcpyDeviceToHost));
            std::vector<u64> v(t->size_qlp, 1), vs(t->size_qlp);
            for (uint32_t i = 0; i < t->size_qlp; i++) vs[i] = h_shoup(1, primes[t->qlp_prime[i]]);
            for (uint32_t i = 0; i < size_p; i++) { v[size_ql + i] = hi[i].x; vs[size_ql + i] = hi[i].y; }
            t->p_hat_inv_by_limb.upload(v);
            t->p_hat_inv_by_limb_shoup.upload(vs);
        }
        // device descriptors for the batched launches
        auto describe = [](const BConv &b, uint32_t pad_start, uint32_t pad_len, uint32_t src_limb, uint32_t copy_own) {
            return BConvDev{b.hat_inv.p, b.d_iprime.p, b.d_oprime.p, b.mat.p, b.mat30.p, b.mont ? b.oninv.p : nullptr, b.isz, b.osz,
                            pad_start, pad_len, src_limb, copy_own, b.row_pad, b.r90 ? 1u : 0u};
        };
        std::vector<BConvDev> dd;
        for (uint32_t b = 0; b < t->beta; b++) {
            const uint32_t s = t->alpha * b;
            dd.push_back(describe(t->digit[b], s, t->digit[b].isz, s, 1));
        }
        t->d_digit_convs.upload(dd);
        t->d_p_to_ql_conv.upload({describe(t->p_to_ql, 0xffffffffu, 0, size_ql, 0)});
        t->d_p_to_ql_pinv_conv.upload({describe(t->p_to_ql_pinv, 0xffffffffu, 0, size_ql, 0)});
        t->split_ok = true;
        for (uint32_t i = 0; i < t->size_qlp; i++)
            if (primes[t->qlp_prime[i]] >> 60) t->split_ok = false;
        // the carry-free form the conversion launches may use: the digits' common kind (1 = the 30 / 30 split of the headline
        // sets; 2 / 3 = special bases of 17..32 primes or 61-bit primes, BConv::split_kind), 0 if they differ
        t->modup_split = t->digit[0].split_kind;
        for (uint32_t b = 1; b < t->beta; b++)
            if (t->digit[b].split_kind != t->modup_split) t->modup_split = 0;
        t->moddown_split = t->p_to_ql.split_kind;
    }
    if (plain_t) {
        // rns.cu:200-212 (q_last, q_last^-1 mod t), :270-284 (P, P^-1 mod t; P -> {t} converter row)
        const u64 pt = plain_t;
        t->t_mod = h_modulus(pt);
        const u64 iq = h_invmod(primes[size_ql - 1] % pt, pt);
        t->inv_q_last_mod_t = u64x2{iq, h_shoup(iq, pt)};
        if (size_ql > 1) {
            std::vector<u64x2> v(size_ql - 1);
            for (uint32_t i = 0; i + 1 < size_ql; i++) {
                const u64 r = primes[size_ql - 1] % primes[i];
                v[i] = u64x2{r, h_shoup(r, primes[i])};
            }
            t->q_last_mod_q2.upload(v);
        }
        if (size_p) {
            u64 p_t = 1 % pt;
            std::vector<u64> hat(size_p);
            for (uint32_t k = 0; k < size_p; k++) {
                p_t = h_mulmod(p_t, primes[size_q + k] % pt, pt);
                u64 h = 1 % pt;
                for (uint32_t j = 0; j < size_p; j++)
                    if (j != k) h = h_mulmod(h, primes[size_q + j] % pt, pt);
                hat[k] = h;
            }
            const u64 ip = h_invmod(p_t, pt);
            t->pinv_mod_t = u64x2{ip, h_shoup(ip, pt)};
            t->p_hat_mod_t.upload(hat);
        }
        {   // BFV enc / add / sub (rns.cu:292-324): -Ql mod t and t^-1 mod q_i; q_i - t for the plain lift
            u64 ql_t = 1 % pt;
            std::vector<u64> ti(size_ql), tis(size_ql), inc(size_ql);
            for (uint32_t i = 0; i < size_ql; i++) {
                ql_t = h_mulmod(ql_t, primes[i] % pt, pt);
                ti[i] = h_invmod(pt % primes[i], primes[i]);
                tis[i] = h_shoup(ti[i], primes[i]);
                inc[i] = primes[i] - pt;
            }
            const u64 neg = pt - ql_t;
            t->neg_ql_mod_t = u64x2{neg, h_shoup(neg, pt)};
            t->t_inv_mod_q.upload(ti);
            t->t_inv_mod_q_shoup.upload(tis);
            t->plain_upper_half_increment.upload(inc);
        }
        t->bgv_ready = true;
    }
    Tool &ref = *t;
    tools[size_ql] = std::move(t);
    return ref;
}

// ---- per-thread arenas of the sentinel streams ---------------------------------------------------------------------------------
// hipStreamPerThread and the null stream name a different real stream in every host thread, so those two handles get one arena
// per calling thread, keyed by a process-wide thread number (never reused: no two live threads share an arena).  A thread that
// exits gives its arenas back: a thread_local reaper walks the live contexts at thread exit (a pool of short-lived threads no
// longer grows one scratch arena -- hundreds of MB at N = 2^16 -- per thread for the life of the context).
namespace {
std::mutex g_live_mu;
std::vector<Context *> g_live_contexts;
std::atomic<size_t> g_thread_counter{0};

struct ThreadReaper {
    size_t id = 0;
    ~ThreadReaper() {
        if (!id) return;
        // the blocks are unhooked under the locks and freed after them: hipFree waits for the device, and no other caller of any
        // context should stall behind that wait (r03 advisor)
        std::vector<std::unique_ptr<Arena>> dead;
        {
            std::lock_guard<std::mutex> lk(g_live_mu);
            for (Context *c : g_live_contexts) c->release_thread_arenas(id, dead);
        }
        int dev = 0;
        const bool have_dev = hipGetDevice(&dev) == hipSuccess;
        dead.clear();   // (~DevBuf: hipFree; device pointers are valid process-wide, the current device does not matter)
        if (have_dev) (void)hipSetDevice(dev);
    }
};
thread_local ThreadReaper t_reaper;

size_t this_thread_number() {
    if (!t_reaper.id) t_reaper.id = ++g_thread_counter;
    return t_reaper.id;
}
}  // namespace

void register_context(Context *c, bool alive) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (alive) g_live_contexts.push_back(c);
    else g_live_contexts.erase(std::remove(g_live_contexts.begin(), g_live_contexts.end(), c), g_live_contexts.end());
}

void Context::release_thread_arenas(size_t thread_number, std::vector<std::unique_ptr<Arena>> &dead) {
    std::lock_guard<std::mutex> lk(mu);
    // (the caller frees `dead` outside the locks; DevBuf's hipFree waits for the device, so nothing the exiting thread enqueued can
    //  still be using a block.  A hipGraph that the thread captured on hipStreamPerThread / the null stream has these pointers baked
    //  in and must not be replayed after the thread has exited: capture on an explicit stream instead -- include/phantom_amd.h)
    for (auto *m : {&arenas, &outer_arenas})
        for (auto it = m->begin(); it != m->end();) {
            if (it->first.second == thread_number) {
                dead.push_back(std::move(it->second));
                it = m->erase(it);
            } else {
                ++it;
            }
        }
}

Context::ArenaKey Context::arena_key(void *stream) {
    const bool sentinel = stream == nullptr || as_stream(stream) == hipStreamPerThread;
    return ArenaKey{stream, sentinel ? this_thread_number() : 0};
}

void Lanes::ensure() {
    if (s[0]) return;
    for (int l = 0; l < 2; l++) {
        PHA_HIP(hipStreamCreateWithFlags(&s[l], hipStreamNonBlocking));
        PHA_HIP(hipEventCreateWithFlags(&join[l], hipEventDisableTiming));
    }
    PHA_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
}
Lanes::~Lanes() {
    for (int l = 0; l < 2; l++) {
        if (join[l]) (void)hipEventDestroy(join[l]);
        if (s[l]) (void)hipStreamDestroy(s[l]);
    }
    if (fork) (void)hipEventDestroy(fork);
}

u64 *Context::scratch(void *stream, size_t words) {
    std::lock_guard<std::mutex> lk(mu);
    auto &a = arenas[arena_key(stream)];
    if (!a) a = std::make_unique<Arena>();
    if (a->buf.count < words) {
        // growing invalidates earlier pointers: make sure nothing in flight still uses the old block
        PHA_HIP(hipStreamSynchronize(as_stream(stream)));
        a->buf.alloc(words);
    }
    return a->buf.p;
}

uint32_t *Context::ntt_flags(void *stream, size_t units) {
    if (units > kFlagUnits || !flag_pool.p) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    const ArenaKey key = arena_key(stream);
    auto it = flag_arenas.find(key);
    if (it == flag_arenas.end()) {
        if (flag_arenas.size() >= kFlagArenas) return nullptr;
        it = flag_arenas.emplace(key, (uint32_t)flag_arenas.size()).first;
    }
    return flag_pool.p + (size_t)it->second * (2 * kFlagUnits + 16);
}

u64 *Context::scratch_outer(void *stream, size_t words) {
    std::lock_guard<std::mutex> lk(mu);
    auto &a = outer_arenas[arena_key(stream)];
    if (!a) a = std::make_unique<Arena>();
    if (a->buf.count < words) {
        PHA_HIP(hipStreamSynchronize(as_stream(stream)));
        a->buf.alloc(words);
    }
    return a->buf.p;
}

const uint32_t *Context::galois_table(uint32_t elt) {
    // NTT-domain permutation table, include/galois.cuh:98-113
    std::lock_guard<std::mutex> lk(mu);
    auto it = galois_tables.find(elt);
    if (it != galois_tables.end()) return it->second.p;
    if (!(elt & 1) || elt >= 2 * n) throw std::invalid_argument("Galois element is not valid");
    std::vector<uint32_t> tab(n);
    for (uint32_t i = (uint32_t)n; i < 2 * n; i++) {
        const uint32_t rev = h_brev(i, (int)log_n + 1);
        u64 raw = ((u64)elt * rev) >> 1;
        raw &= (u64)(n - 1);
        tab[i - n] = h_brev((uint32_t)raw, (int)log_n);
    }
    DevBuf<uint32_t> d;
    d.upload(tab);
    const uint32_t *p = d.p;
    galois_tables[elt] = std::move(d);
    return p;
}

}  // namespace pha

using namespace pha;

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *pha_last_error(void) { return g_last_error.c_str(); }

int pha_coeff_modulus_create(uint64_t n, const int *bit_sizes, size_t count, uint64_t *out) {
    PHA_API_BEGIN
    if (n < 2 || n > 131072 || (n & (n - 1))) throw std::invalid_argument("poly_modulus_degree is invalid");
    if (count > 64) throw std::invalid_argument("bit_sizes is invalid");
    // src/host/modulus.cu:98-109: per size, find `need` primes descending, hand out from the back
    std::map<int, std::vector<u64>> table;
    std::map<int, size_t> need;
    for (size_t i = 0; i < count; i++) need[bit_sizes[i]]++;
    for (auto &kv : need) get_primes(n, kv.first, kv.second, table[kv.first]);
    for (size_t i = 0; i < count; i++) {
        auto &v = table[bit_sizes[i]];
        out[i] = v.back();
        v.pop_back();
    }
    PHA_API_END
}

int pha_context_create(pha_context_t *out, uint32_t log_n, const uint64_t *primes_qp, uint32_t size_qp,
                       uint32_t size_p, int device_id) {
    PHA_API_BEGIN
    if (!out || !primes_qp) throw std::invalid_argument("null argument");
    auto h = std::make_unique<pha_context>();
    context_init(h->c, log_n, primes_qp, size_qp, size_p, device_id);
    register_context(&h->c, true);
    *out = h.release();
    PHA_API_END
}

void pha_context_destroy(pha_context_t ctx) {
    if (!ctx) return;
    register_context(&ctx->c, false);
    (void)hipSetDevice(ctx->c.device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

uint32_t pha_context_log_n(pha_context_t ctx) { return ctx->c.log_n; }
uint32_t pha_context_size_qp(pha_context_t ctx) { return ctx->c.size_qp; }
uint32_t pha_context_size_p(pha_context_t ctx) { return ctx->c.size_p; }

int pha_context_set_plain_modulus(pha_context_t ctx, uint64_t plain_modulus) {
    PHA_CTX_BEGIN(ctx)
    if (!ctx) throw std::invalid_argument("null context");
    Context &c = ctx->c;
    if (plain_modulus == 1 || (plain_modulus >> 60)) throw std::invalid_argument("plain_modulus is not valid");
    auto gcd = [](u64 a, u64 b) {
        while (b) { const u64 r = a % b; a = b; b = r; }
        return a;
    };
    for (u64 q : c.primes)  // every q_i and p_j must be invertible modulo t (rns.cu:207,274)
        if (plain_modulus && gcd(q, plain_modulus) != 1) throw std::logic_error("invalid rns bases");
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.plain_t != plain_modulus) {
        PHA_HIP(hipSetDevice(c.device));
        PHA_HIP(hipDeviceSynchronize());  // per-level tools are rebuilt lazily with the new constants
        c.tools.clear();
        c.behz_tool.reset();   // (their auxiliary table rows stay; a later tool appends fresh ones)
        c.hps_tool.reset();
        c.hpsq_tools.clear();
        c.plain_t = plain_modulus;
    }
    PHA_API_END
}

int pha_context_prime_info(pha_context_t ctx, uint32_t i, uint64_t *value, uint64_t ratio[2], uint64_t *root,
                           uint64_t *n_inv) {
    PHA_CTX_BEGIN(ctx)
    Context &c = ctx->c;
    std::lock_guard<std::mutex> lk(c.mu);   // the auxiliary rows (BFV bases) are appended on first use
    if (i >= c.rows) throw std::invalid_argument("prime index out of range");
    if (value) *value = c.primes[i];
    if (ratio) { ratio[0] = c.mods[i].ratio0; ratio[1] = c.mods[i].ratio1; }
    if (root) *root = c.roots[i];
    if (n_inv) *n_inv = c.n_inv[i];
    PHA_API_END
}

int pha_context_download_twiddle(pha_context_t ctx, uint32_t i, int which, uint64_t *host_out) {
    PHA_CTX_BEGIN(ctx)
    Context &c = ctx->c;
    if (i >= c.size_qp || which < 0 || which > 3) throw std::invalid_argument("bad twiddle selector");
    PHA_HIP(hipSetDevice(c.device));
    std::vector<u64x2> row(c.n);
    const u64x2 *src = (which < 2 ? c.d_tw.p : c.d_itw.p) + (size_t)i * c.n;
    PHA_HIP(hipMemcpy(row.data(), src, c.n * sizeof(u64x2), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < c.n; k++) host_out[k] = (which & 1) ? row[k].y : row[k].x;
    if (which >= 2) {  // present the reference's folded slot 1 (src/host/ntt.cu:53-55)
        const u64 q = c.primes[i];
        const u64 w1 = h_mulmod(row[1].x, c.n_inv[i], q);
        host_out[1] = (which == 2) ? w1 : h_shoup(w1, q);
    }
    PHA_API_END
}

int pha_tool_beta(pha_context_t ctx, uint32_t size_ql, uint32_t *beta) {
    PHA_CTX_BEGIN(ctx)
    *beta = ctx->c.tool(size_ql).beta;
    PHA_API_END
}

}  // extern "C"
